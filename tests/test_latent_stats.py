"""Latent statistics (reference analysis.py:214-376).  CPU: the restatement against
np.quantile on float64 data and sklearn's roc_auc_score.  GPU (through the C-ABI): exact order statistics,
bit-exact float64 quantiles and exact TP/FN/FP/TN counts against the restatement, AUC ranking identical."""
import numpy as np
import pytest

from oracle import stats_oracle as so


def _latents(n, L, seed):
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n, L)).astype(np.float32) * rng.uniform(0.2, 3.0, L).astype(np.float32)
    z[rng.integers(0, n, n // 10), rng.integers(0, L, n // 10)] = 0.0          # ties and signed zeros
    z[: n // 20] = np.round(z[: n // 20], 1)                                    # more ties
    y = rng.integers(0, 3, n)                                                   # labels {0,1,2} (label_enum.py:1-3)
    z[y == 1, 0] *= 2.5                                                         # a unit that separates blinks
    return z, y


def test_quantile_restatement_agrees_with_numpy_on_float64():
    z, _ = _latents(2701, 28, 0)
    for q in (0.05, 0.25, 0.5, 0.75, 0.95, 0.0, 1.0):
        ref = np.quantile(z.astype(np.float64), q, axis=0)
        got = so.quantile(z, q)
        assert got.dtype == np.float64
        assert np.allclose(got, ref, rtol=1e-6, atol=1e-7)


def test_quantile_matrix_has_18_columns_without_the_median():
    z, _ = _latents(500, 8, 1)
    M = so.quantile_matrix(z)
    assert M.shape == (8, 18)
    assert np.all(np.diff(M, axis=1) >= 0)


def test_host_auc_equals_sklearn():
    sk = pytest.importorskip("sklearn.metrics")
    import importlib, sys, types
    # the host AUC lives in the product module; import it without touching the GPU library
    src = open("vae_eeg_xai_amd/latent_stats.py").read()
    ns = {}
    start = src.index("def _roc_auc")
    end = src.index("def auc_roc")
    exec("import numpy as np\n" + src[start:end], ns)
    rng = np.random.default_rng(3)
    for _ in range(20):
        s = np.round(rng.uniform(0, 1, 18), 1)      # ties on purpose
        y = np.concatenate((np.zeros(9), np.ones(9)))
        assert abs(ns["_roc_auc"](y, s) - sk.roc_auc_score(y, s)) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("n,L", [(1, 3), (2, 28), (37, 28), (2701, 28), (100_000, 28), (5000, 128)])
def test_device_quantiles_are_bit_exact(n, L):
    from vae_eeg_xai_amd import latent_stats as ls
    z, _ = _latents(n, L, n)
    qs = [0.0, 0.05, 0.25, 0.5, 0.75, 0.95, 1.0, 1.0 / 3.0]
    got = ls.column_quantiles(z, qs).cpu().numpy()
    ref = so.quantile(z, qs)
    assert got.dtype == np.float64 and got.shape == ref.shape
    assert np.array_equal(got, ref)
    assert np.array_equal(ls.column_quantiles(z, 0.25).cpu().numpy(), so.quantile(z, 0.25))


@pytest.mark.gpu
def test_device_more_than_24_quantiles_and_quantile_matrix():
    from vae_eeg_xai_amd import latent_stats as ls
    z, _ = _latents(3333, 28, 5)
    qs = list(np.linspace(0, 1, 41))
    assert np.array_equal(ls.column_quantiles(z, qs).cpu().numpy(), so.quantile(z, qs))
    assert np.array_equal(ls.quantile_matrix(z), so.quantile_matrix(z))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [300, 2701, 50_000])
def test_device_counts_and_auc_ranking_match_the_reference_loops(n):
    from vae_eeg_xai_amd import latent_stats as ls
    z, y = _latents(n, 28, 11)
    zb, zn = z[y == 1], z[y == 0]
    q25, q75, tp, fn, fp, tn = ls.histogram_25_75_counts(z, zb, zn)
    r25, r75, rtp, rfn, rfp, rtn = so.histogram_25_75_counts(z, zb, zn) if n <= 3000 else (None,) * 6
    if r25 is not None:
        assert np.array_equal(q25, r25) and np.array_equal(q75, r75)
        assert np.array_equal(tp, rtp) and np.array_equal(fn, rfn) and np.array_equal(fp, rfp) and np.array_equal(tn, rtn)
    assert np.all(tp + fn == len(zb)) and np.all(fp + tn == len(zn))
    qm = ls.quantile_matrix(z)
    got = ls.auc_roc(qm, zb, zn)
    ref = so.auc_roc(so.quantile_matrix(z), zb, zn, ls._roc_auc)
    assert [j for j, _ in got] == [j for j, _ in ref]
    assert np.allclose([a for _, a in got], [a for _, a in ref], rtol=0, atol=1e-15)
    assert got[0][0] == 0      # the planted unit ranks first (top-k = 1 selection of analysis.py:725-726)
    try:
        from sklearn.metrics import roc_auc_score
        ref2 = so.auc_roc(so.quantile_matrix(z), zb, zn, roc_auc_score)
        assert np.allclose([a for _, a in got], [a for _, a in ref2], atol=1e-12)
    except ImportError:
        pass
