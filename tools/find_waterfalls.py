"""Scan hipcc assembly (-S --cuda-device-only) for readfirstlane "waterfall" loops around memory instructions: the
sign that a buffer descriptor or scalar offset the source meant to be wave-uniform ended up in VGPRs.
usage: find_waterfalls.py file.s ..."""
import re, sys
for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    kernel, found = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+|\w+_kernel\w*):", l)
        if m:
            kernel = m.group(1)
        if "s_cbranch_execnz" in l:
            tgt = l.split()[-1]
            j = i
            while j > 0 and i - j < 24 and not lines[j].startswith(tgt + ":"):
                j -= 1
            body = lines[j:i]
            if lines[j].startswith(tgt + ":") and any("v_readfirstlane" in b for b in body) and \
               any(re.search(r"\b(buffer|global|ds)_", b) for b in body):
                found[kernel] = found.get(kernel, 0) + 1
    print(path, "none" if not found else "")
    for k, v in found.items():
        print(f"   {v:3d} waterfall loop(s) in {k}")
