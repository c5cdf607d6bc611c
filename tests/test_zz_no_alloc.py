"""The allocation-free entry point veex_minmax_static (include/veex.h) through the C-ABI: bit-identical to
veex_minmax (the existing entry point, whose partials come in a caller buffer), on two calls in a row, the second
with a smaller input (the array keeps the first call's partial rows), with no device allocation during the
calls, and a clear error beyond the stated limit.  Collected last (test_zz_*): the round's end runs pytest -x."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_minmax_static_matches_minmax_twice_and_allocates_nothing():
    import torch
    from vae_eeg_xai_amd._lib import lib, ptr, current_stream
    rng = np.random.default_rng(0)
    big = torch.from_numpy(rng.standard_normal(3_000_003).astype(np.float32) * 50.0).cuda()     # 733 blocks, ragged tail
    small = torch.from_numpy(rng.standard_normal(1_001).astype(np.float32)).cuda()             # 1 block
    small[17] = 4.5      # extrema well inside the first call's range: stale partial rows would show
    small[400] = -3.25
    out_s = torch.zeros(2, dtype=torch.float32, device="cuda")
    out_r = torch.zeros(2, dtype=torch.float32, device="cuda")
    part = torch.empty(2 * 2048, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for x in (big, small):
        n = x.numel()
        before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
        assert lib.veex_minmax_static(ptr(x), n, ptr(out_s), current_stream()) == 0, lib.veex_last_error()
        torch.cuda.synchronize()
        after = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
        if x is small:      # (the first call on a device may load the code object; from then on nothing may move)
            assert after == before
        assert lib.veex_minmax(ptr(x), n, ptr(part), ptr(out_r), current_stream()) == 0
        torch.cuda.synchronize()
        got, ref = out_s.cpu().numpy(), out_r.cpu().numpy()
        assert np.array_equal(got, ref), (n, got, ref)
        xh = x.cpu().numpy()
        assert got[0] == xh.min() and got[1] == xh.max()
    assert np.array_equal(out_s.cpu().numpy(), np.array([-3.25, 4.5], dtype=np.float32))
    # beyond the stated limit: refused on the host, nothing launched
    assert lib.veex_minmax_static(ptr(big), (1 << 32) + 1, ptr(out_s), current_stream()) == -1
    assert b"limit" in lib.veex_last_error()
    assert lib.veex_minmax_static(None, 5, ptr(out_s), current_stream()) == -1
