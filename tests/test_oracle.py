"""CPU tests of the oracle itself: two independent formulations, finite differences, known-answer
vectors for Philox and the TF padding arithmetic, and the committed golden fixtures."""
import numpy as np
import pytest

from oracle import vae_oracle as vo, torch_ref as tr
from util import golden_cases, rel_err


@pytest.mark.parametrize("variant", ["vae", "ae"])
def test_numpy_and_torch_formulations_agree(variant):
    L, B = 28, 2
    p = vo.init_params(L, variant, seed=42, jitter=0.1)
    x = vo.synth_topomaps(B, 1)
    eps = np.random.default_rng(7).standard_normal((B, L))
    c = vo.forward(p, x, eps, variant)
    g = vo.backward(p, c, variant)
    tot, rec, kl, gt = tr.grads(p, x, eps, variant)
    assert abs(tot - c["total"]) < 1e-9 * abs(tot)
    assert abs(rec - c["rec_loss"]) < 1e-9 * abs(rec)
    assert abs(kl - c["kl_loss"]) < 1e-9 * max(abs(kl), 1.0)
    for k in vo.trainable_names(L, variant):
        assert rel_err(g[k], gt[k]) < 1e-10, k
    s1 = vo.attribution(p, x, [0, 5], variant)
    s2 = tr.saliency(p, x, [0, 5], variant)
    assert rel_err(s1, s2) < 1e-10


def test_finite_differences():
    L, B = 8, 2
    p = vo.init_params(L, "vae", seed=3, jitter=0.1)
    x = vo.synth_topomaps(B, 2)
    eps = np.random.default_rng(1).standard_normal((B, L))
    g = vo.backward(p, vo.forward(p, x, eps))
    rng = np.random.default_rng(0)
    for k in ["enc.conv2.kernel", "dec.deconv1.kernel", "enc.conv1.bias", "dec.dense.kernel", "enc.z_log_var.kernel",
              "dec.deconv5.kernel", "dec.deconv6.kernel"]:
        idx = tuple(int(rng.integers(0, s)) for s in p[k].shape)
        h = 1e-6
        p[k][idx] += h
        fp = vo.forward(p, x, eps)["total"]
        p[k][idx] -= 2 * h
        fm = vo.forward(p, x, eps)["total"]
        p[k][idx] += h
        fd = (fp - fm) / (2 * h)
        assert abs(fd - g[k][idx]) < 1e-5 * max(1.0, abs(fd)), (k, fd, g[k][idx])


def test_param_count_and_order():
    spec = vo.param_spec(28, "vae")
    n_tr = sum(int(np.prod(s)) for _, s, t in spec if t)
    assert n_tr == 2_288_253
    assert sum(1 for _, _, t in spec if t) == 44   # 18 encoder + 26 decoder arrays
    assert sum(1 for _, _, t in spec if not t) == 18
    assert spec[0][0] == "enc.conv1.kernel" and spec[0][1] == (3, 3, 1, 64)
    for L in (2, 8, 16, 28, 64, 128):
        n = sum(int(np.prod(s)) for _, s, t in vo.param_spec(L, "vae") if t)
        assert n == 2_167_909 + 4_298 * L


def test_tf_padding_arithmetic():
    # k3 s2 SAME on even input -> pad 0 before / 1 after
    for n in (40, 20, 10):
        assert vo.same_pad(n, 3, 2) == (n // 2, 0, 1)
    # spatial chain 4 -> 8 -> 8 -> 17 -> 19 -> 39 -> 40 and the crop-at-start values
    h = 4
    chain, crops = [], []
    for name, cin, cout, k, s, pad in vo.DEC_DECONVS:
        h, pb, full = vo.deconv_geometry(h, k, s, pad)
        chain.append(h)
        crops.append(pb)
    assert chain == [8, 8, 17, 19, 39, 40]
    assert crops == [0, 1, 0, 0, 0, 0]


def test_conv_transpose_is_adjoint_of_conv():
    # <conv(x), y> == <x, conv_transpose(y)> for the SAME/stride-2 pair (defines the crop rule)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 8, 8, 3))
    W = rng.standard_normal((3, 3, 3, 5))
    y = rng.standard_normal((2, 4, 4, 5))
    lhs = (vo.conv2d_fwd(x, W, np.zeros(5), 2, "same") * y).sum()
    rhs = (x * vo.deconv2d_fwd(y, W, np.zeros(3), 2, "same")).sum()
    assert abs(lhs - rhs) < 1e-10 * abs(lhs)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = vo.philox4x32(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(v) for v in got) == want


def test_philox_normal_is_split_invariant():
    full = vo.philox_normal(42, 0, 64, 28)
    part = vo.philox_normal(42, 16, 8, 28)
    assert np.array_equal(full[16:24], part)
    assert abs(full.mean()) < 0.1 and abs(full.std() - 1) < 0.1


def test_adam_tf_form_single_step():
    p = {"w": np.array([1.0, -2.0])}
    g = {"w": np.array([0.5, -0.25])}
    m = {"w": np.zeros(2)}
    v = {"w": np.zeros(2)}
    vo.adam_step(p, g, m, v, 1, 1e-3)
    # t=1: m = 0.1 g, v = 0.001 g^2, alpha = lr*sqrt(0.001)/0.1 -> step = lr * g/(|g| + eps/sqrt(0.001)) ~ lr*sign(g)
    alpha = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    want = np.array([1.0, -2.0]) - alpha * (0.1 * g["w"]) / (np.sqrt(0.001 * g["w"] ** 2) + 1e-7)
    assert np.allclose(p["w"], want, rtol=0, atol=1e-15)


@pytest.mark.parametrize("path,variant,L,B,seed", golden_cases())
def test_golden_fixtures_reproduce(path, variant, L, B, seed):
    f = np.load(path, allow_pickle=False)
    p = vo.init_params(L, variant, seed=seed, jitter=0.1)
    eps = f["eps"] if variant == "vae" else None
    c = vo.forward(p, f["x"], eps, variant)
    assert rel_err(c["logits"], f["logits"]) < 1e-12
    assert abs(c["total"] - float(f["total"])) < 1e-12 * abs(float(f["total"]))
    g = vo.backward(p, c, variant)
    for k in vo.trainable_names(L, variant):
        assert abs(g[k].sum() - float(f["gsum." + k])) <= 1e-9 * (float(f["gabs." + k]) + 1e-30), k
    sal = vo.attribution(p, f["x"], [int(u) for u in f["units"]], variant)[..., 0]
    assert rel_err(sal, f["saliency"]) < 1e-12


# "probs64": the probability form in real arithmetic (the formula and its derivative are what these tests check);
# sem "probs" evaluates it in float32, as Keras does -- test_probability_bce_float32_evaluation below
SEMS = [{"bn": "batch"}, {"bce": "probs64"}, {"bn": "batch", "bce": "probs64"}]


@pytest.mark.parametrize("sem", SEMS, ids=lambda s: "+".join(f"{k}={v}" for k, v in s.items()))
@pytest.mark.parametrize("variant", ["vae", "ae"])
def test_alternate_keras_semantics_two_formulations_agree(variant, sem):
    """The two unpinnable [KERAS-SEM] readings behind switches: BatchNormalization on batch
    statistics with the backward through the statistics, and the clipped-probability BCE -- hand-derived NumPy
    against torch autograd."""
    L, B = 8, 3
    p = vo.init_params(L, variant, seed=5, jitter=0.1)
    x = vo.synth_topomaps(B, 4)
    eps = np.random.default_rng(2).standard_normal((B, L))
    c = vo.forward(p, x, eps, variant, sem)
    g = vo.backward(p, c, variant)
    tot, rec, kl, gt = tr.grads(p, x, eps, variant, sem=sem)
    assert abs(tot - c["total"]) < 1e-9 * abs(tot) and abs(rec - c["rec_loss"]) < 1e-9 * abs(rec)
    for k in vo.trainable_names(L, variant):
        assert rel_err(g[k], gt[k]) < 1e-9, k
    d = vo.forward(p, x, eps, variant)                        # and the switch changes the numbers
    assert abs(d["total"] - c["total"]) > 1e-9 * abs(d["total"])


@pytest.mark.parametrize("sem", SEMS[:2], ids=["bn=batch", "bce=probs"])
def test_alternate_keras_semantics_finite_differences(sem):
    L, B = 8, 3
    p = vo.init_params(L, "vae", seed=6, jitter=0.1)
    x = vo.synth_topomaps(B, 3)
    eps = np.random.default_rng(3).standard_normal((B, L))
    g = vo.backward(p, vo.forward(p, x, eps, "vae", sem))
    rng = np.random.default_rng(1)
    for k in ["enc.conv2.kernel", "enc.conv1.gamma", "dec.deconv1.kernel", "dec.dense.beta", "dec.deconv4.bias",
              "dec.deconv5.gamma", "dec.deconv6.kernel", "enc.z_mean.kernel"]:
        idx = tuple(int(rng.integers(0, s)) for s in p[k].shape)
        h = 1e-6
        p[k][idx] += h
        fp = vo.forward(p, x, eps, "vae", sem)["total"]
        p[k][idx] -= 2 * h
        fm = vo.forward(p, x, eps, "vae", sem)["total"]
        p[k][idx] += h
        fd = (fp - fm) / (2 * h)
        assert abs(fd - g[k][idx]) < 2e-5 * max(1.0, abs(fd)), (k, fd, g[k][idx])


def test_probability_bce_is_the_logit_form_away_from_saturation_and_clips_at_it():
    a = np.array([-30.0, -17.0, -3.0, 0.0, 2.5, 17.0, 30.0])
    t = np.array([0.0, 0.3, 1.0, 0.5, 0.2, 1.0, 0.4])
    lo, pr = vo.bce_logits(a, t), vo.bce_probs(a, t, f32=False)
    # the +1e-7 inside the logarithms shifts a term by ~1e-7 / p: 2e-6 at p = 0.05 -- the two forms differ at the
    # 1e-6 .. 1e-5 level in loss and gradients (SURVEY A.4 says "<~ 1e-7"), i.e. right at the parity tolerance
    assert np.allclose(lo[2:5], pr[2:5], rtol=0, atol=5e-6)
    assert abs(pr[0] - (-np.log(1 - vo.BCE_CLIP + vo.BCE_CLIP))) < 1e-12
    # float32's 1 - 1e-7 is 1 - 2**-23: a saturated pixel costs -log(2.19e-7)
    assert abs(pr[6] - (-(0.4 * np.log(1.0) + 0.6 * np.log(2.0 ** -23 + 1e-7)))) < 1e-6
    d = vo.bce_probs_dlogit(a, t, f32=False)
    assert d[0] == 0.0 and d[6] == 0.0 and d[4] != 0.0                # no gradient through an active clip


def test_probability_bce_float32_evaluation():
    """sem "probs" follows Keras's float32 arithmetic: equal to the real-arithmetic formula below the saturation band
    (to float32 rounding), a different -- piecewise constant -- function inside it."""
    rng = np.random.default_rng(0)
    a = rng.uniform(-vo.BCE_PROBS_BAND, vo.BCE_PROBS_BAND, 4000)
    t = rng.uniform(0, 1, 4000)
    assert np.abs(vo.bce_probs(a, t) - vo.bce_probs(a, t, f32=False)).max() < 2e-2      # 6e-8 / (1 - p) at |a| = 12
    m = np.abs(a) < 6
    assert np.abs(vo.bce_probs(a, t) - vo.bce_probs(a, t, f32=False))[m].max() < 5e-5
    dd = np.abs(vo.bce_probs_dlogit(a, t) - vo.bce_probs_dlogit(a, t, f32=False))
    assert dd.max() < 3e-4 and dd[np.abs(a) < 8].max() < 1e-6
    # inside the band: float32's sigmoid is exactly 1 from a = 16.7 on and the clip edge moves to where 1 - p rounds
    hi = np.array([15.0, 15.5, 15.9, 16.3, 17.0, 40.0])
    d32, d64 = vo.bce_probs_dlogit(hi, np.zeros(6)), vo.bce_probs_dlogit(hi, np.zeros(6), f32=False)
    assert d32[-1] == 0.0 and d64[-1] == 0.0 and d32[0] > 0.5
    assert np.abs(d32 - d64).max() > 0.1                              # the discontinuity the band exists for
    c32 = vo.bce_probs(hi, np.zeros(6))
    assert abs(c32[-1] - (-np.log(2.0 ** -23 + 1e-7))) < 1e-6


def test_batch_statistics_mode_updates_the_moving_statistics():
    """momentum 0.99; unbiased batch variance into the 4-D layers' moving variance, biased into dec.dense's."""
    L, B = 8, 3
    p = vo.init_params(L, "vae", seed=7, jitter=0.1)
    x = vo.synth_topomaps(B, 5)
    eps = np.random.default_rng(4).standard_normal((B, L))
    c = vo.forward(p, x, eps, "vae", {"bn": "batch"})
    a = c["enc.conv2.a"]
    n = a.size // a.shape[-1]
    assert np.allclose(c["bn_state"]["enc.conv2.mean"], 0.99 * p["enc.conv2.mean"] + 0.01 * a.mean(axis=(0, 1, 2)))
    assert np.allclose(c["bn_state"]["enc.conv2.var"], 0.99 * p["enc.conv2.var"] + 0.01 * a.var(axis=(0, 1, 2)) * n / (n - 1))
    ad = c["dec.dense.a"]
    assert np.allclose(c["bn_state"]["dec.dense.var"], 0.99 * p["dec.dense.var"] + 0.01 * ad.var(axis=0))
    assert "bn_state" not in vo.forward(p, x, eps, "vae")              # default: moving statistics stay frozen
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(v_) for k, v_ in p.items()}
    before = p["enc.conv2.mean"].copy()
    vo.train_step(p, m, v, 1, 1e-3, x, eps, "vae", {"bn": "batch"})
    assert not np.array_equal(before, p["enc.conv2.mean"])
