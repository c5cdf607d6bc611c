"""Compare two fixture files key by key: ``python tests/golden/compare_fixture.py got.npz want.npz [rtol]``.
Used to pin parity: ``got`` from tools/dump_tf_train_step.py (the real TF/Keras reference, run elsewhere),
``want`` a committed fixture of this repository (made by make_golden.py from the float64 oracle).  Every key
present in both is compared relative to the max-abs of the wanted array (default 1e-5; gsum.* relative to
the matching gabs.*)."""
import sys

import numpy as np


def compare(got_path, want_path, rtol=1e-5):
    got, want = np.load(got_path), np.load(want_path)
    bad, n = [], 0
    for k in want.files:
        if k not in got.files:
            continue
        n += 1
        a, b = np.asarray(got[k], dtype=np.float64), np.asarray(want[k], dtype=np.float64)
        if a.shape != b.shape:
            bad.append((k, "shape", a.shape, b.shape))
            continue
        scale = float(np.asarray(want["gabs." + k[5:]])) if k.startswith("gsum.") else float(np.abs(b).max())
        err = float(np.abs(a - b).max()) / (scale + 1e-30)
        if not err <= rtol:
            bad.append((k, err))
    return n, bad


def which_semantics(got_path, want_path, rtol=1e-5):
    """``got`` against the committed fixture AND its alternate-[KERAS-SEM] siblings (``<fixture>__bn-batch.npz``,
    ``__bce-probs.npz``, ``__bn-batch+bce-probs.npz``, made by make_golden.py): returns [(flags, keys, bad)], so
    that a real TF run that does not follow the default reading names the switches to flip
    (Net.set_semantics / veex_net_set_semantics) instead of just failing."""
    import os
    out = [("bn=inference bce=logits (default)",) + compare(got_path, want_path, rtol)]
    for tag, flags in (("bn-batch", "bn=batch bce=logits"), ("bce-probs", "bn=inference bce=probs"),
                       ("bn-batch+bce-probs", "bn=batch bce=probs")):
        alt = want_path[:-4] + "__" + tag + ".npz"
        if os.path.exists(alt):
            out.append((flags,) + compare(got_path, alt, rtol))
    return out


if __name__ == "__main__":
    rtol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-5
    res = which_semantics(sys.argv[1], sys.argv[2], rtol)
    for flags, n, bad in res:
        print(f"[{flags}] {n} keys compared, {len(bad)} outside tolerance")
    match = [flags for flags, n, bad in res if n > 0 and not bad]
    print("MATCHES: " + (", ".join(match) if match else "none of the Keras-semantics flag sets"))
    n, bad = res[0][1], res[0][2]
    for b in bad:
        print("  ", b)
    sys.exit(1 if bad or n == 0 else 0)
