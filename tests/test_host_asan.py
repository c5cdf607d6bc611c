"""CPU: libveex's host code under AddressSanitizer.  `make asan` builds the library with -Xarch_host
-fsanitize=address (device code is not instrumented; GPU ASan is not available here) and links
tests/host_asan/host_api_smoke.cpp against it: net creation, tensor tables (with a deliberately short name buffer),
workspace / row-split / tile planners, argument checking and the error buffer -- every path that runs on the host
before anything is enqueued.  No GPU is touched, so this runs in the CPU suite."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vae_eeg_xai_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_host_api_is_clean_under_address_sanitizer():
    b = subprocess.run(["make", "-C", CSRC, "-j8", "asan"], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-2000:]
    r = subprocess.run([os.path.join(CSRC, "_asan", "host_api_smoke")], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    assert "host ASan smoke: ok" in r.stdout
