"""Host logic that needs no GPU: the Keras / TensorFlow names and module paths the reference's drivers
import (train.py:9,10,15,18; analysis.py:23), EarlyStopping, the custom-op registrations, the
get_weights() order of the oracle's parameter table, and the data-parallel bookkeeping."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_early_stopping_patience_zero_and_improvement():
    from vae_eeg_xai_amd.keras_compat import EarlyStopping
    es = EarlyStopping("val_loss", patience=0)
    es.on_train_begin()
    for epoch, v in enumerate([5.0, 4.0, 3.0]):        # improving every epoch: Keras returns before the stop check
        es.on_epoch_end(epoch, {"val_loss": v})
        assert not es.stop_training
    es.on_epoch_end(3, {"val_loss": 3.5})
    assert es.stop_training and es.stopped_epoch == 3
    es = EarlyStopping("val_loss", patience=2)
    es.on_train_begin()
    for epoch, v in enumerate([5.0, 6.0, 4.0, 4.5, 4.4]):
        es.on_epoch_end(epoch, {"val_loss": v})
    assert es.stop_training and es.stopped_epoch == 4 and es.best == 4.0


def test_reference_import_lines_resolve_through_compat():
    """The import block of the reference's train.py (lines 9, 10, 15, 18) and the names it then uses, in a
    fresh interpreter with compat/ first on sys.path (a fresh one: ``keras`` / ``tensorflow`` are global names)."""
    code = (
        "import sys; sys.path.insert(0, sys.argv[1])\n"
        "from keras.optimizers.legacy import Adam\n"
        "from keras.src.callbacks import EarlyStopping\n"
        "from tensorflow.python.ops.numpy_ops import np_config\n"
        "import tensorflow as tf, keras\n"
        "np_config.enable_numpy_behavior()\n"
        "assert Adam(1e-5).learning_rate == 1e-5 and EarlyStopping('val_loss', patience=30, verbose=1).patience == 30\n"
        "assert isinstance(tf.config.list_physical_devices('GPU'), list)\n"
        "tf.keras.backend.clear_session(); keras.backend.clear_session()\n"
        "from keras.callbacks import EarlyStopping as E2; assert E2 is EarlyStopping\n"
        "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "compat")], capture_output=True, text=True,
                       cwd="/tmp", timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-1500:]


def test_custom_ops_are_registered_for_the_gpu_only():
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    import vae_eeg_xai_amd.ops  # noqa: F401
    for name in ("train_fwd_bwd", "forward", "encode", "decode", "attribute", "adam", "sample", "sample_bwd"):
        assert hasattr(torch.ops.veex, name), name
    schema = str(torch.ops.veex.train_fwd_bwd.default._schema)
    assert "Tensor(a9!) grads" in schema and "-> ()" in schema          # declared in-place on the flat gradient buffer
    with pytest.raises(NotImplementedError):                             # no CPU kernel: there is no CPU path
        torch.ops.veex.sample(torch.zeros(2, 3), torch.zeros(2, 3), torch.zeros(2, 3))
    with FakeTensorMode():
        x = torch.empty(5, 40, 40, 1, device="cuda")
        p = torch.empty(16, device="cuda")
        zm, zlv, z = torch.ops.veex.encode(0, p, p, x, None, 1, 42, 0, p, 28, True)
        assert zm.shape == zlv.shape == z.shape == (5, 28)
        recon, losses = torch.ops.veex.forward(0, p, p, x, None, 1, 42, 0, p, None, True)
        assert recon.shape == (5, 40, 40, 1) and losses.shape == (3,)
        assert torch.ops.veex.attribute(0, p, p, x, [1, 2], True, p).shape == (2, 5, 40, 40)


def test_get_weights_order_is_per_sublayer():
    """SURVEY A.6 [KERAS-SEM], parity unpinned: Model.weights = encoder.variables + decoder.variables, a Layer's
    variables = trainable (creation order) + non-trainable."""
    from oracle import vae_oracle as vo
    names = [n for n, _, _ in vo.param_spec(28, "vae")]
    assert len(names) == 62
    assert names[:4] == ["enc.conv1.kernel", "enc.conv1.bias", "enc.conv1.gamma", "enc.conv1.beta"]
    assert names[12:18] == ["enc.dense.kernel", "enc.dense.bias", "enc.z_mean.kernel", "enc.z_mean.bias",
                            "enc.z_log_var.kernel", "enc.z_log_var.bias"]
    assert names[18:24] == ["enc.conv1.mean", "enc.conv1.var", "enc.conv2.mean", "enc.conv2.var", "enc.conv3.mean",
                            "enc.conv3.var"]
    assert names[24] == "dec.dense.kernel" and names[49] == "dec.deconv6.bias"
    assert names[50:52] == ["dec.dense.mean", "dec.dense.var"] and names[-1] == "dec.deconv5.var"
    ae = [n for n, _, _ in vo.param_spec(28, "ae")]
    assert len(ae) == 58 and ae[14] == "enc.conv1.mean" and ae[20] == "dec.dense.kernel"
    # the committed golden fixtures were drawn in layer-creation order: init_params must not depend on this table
    p = vo.init_params(4, "vae", seed=12, jitter=0.1)
    assert abs(float(p["enc.conv1.kernel"].ravel()[0]) - float(vo.init_params(4, "vae", seed=12)["enc.conv1.kernel"].ravel()[0])) == 0.0


def test_ragged_shards_with_the_global_divisor_sum_to_the_global_mean():
    """The arithmetic of VAE._shard_fwd_bwd on the oracle: every rank divides its shard's summed
    gradient by n_global / world; SUM over ranks, times 1 / world, is the mean over the global batch for unequal
    shards and for ranks without samples."""
    from oracle import vae_oracle as vo
    from vae_eeg_xai_amd import dist as vdist
    L, n_global = 4, 7
    p = vo.init_params(L, "vae", seed=2, jitter=0.1)
    x = vo.synth_topomaps(n_global, 9)
    eps = vo.philox_normal(7, 50, n_global, L)
    names = vo.trainable_names(L, "vae")
    cf = vo.forward(p, x, eps, "vae")
    gf = vo.backward(p, cf, "vae")
    for world in (2, 3, 8):
        acc = {k: 0.0 for k in names}
        loss = 0.0
        for r in range(world):
            lo, hi = vdist.shard_range(n_global, r, world)
            if hi == lo:
                continue
            assert np.array_equal(vo.philox_normal(7, 50 + lo, hi - lo, L), eps[lo:hi])
            c = vo.forward(p, x[lo:hi], eps[lo:hi], "vae")
            g = vo.backward(p, c, "vae")
            w = (hi - lo) / (n_global / world)        # local mean -> shard sum / (n_global / world)
            for k in names:
                acc[k] = acc[k] + w * g[k]
            loss += w * c["total"]
        for k in names:
            assert np.abs(acc[k] / world - gf[k]).max() <= 1e-12 * max(1.0, np.abs(gf[k]).max()), (world, k)
        assert abs(loss / world - cf["total"]) < 1e-10 * abs(cf["total"])


def test_layer_op_geometry_is_the_oracles_and_fake_shapes_trace():
    """torch.ops.veex.conv2d / conv_transpose2d: the Keras 'same' / 'valid' arithmetic of ops.py equals the oracle's
    for every layer of the model, and the ops trace under FakeTensorMode (register_fake) without a GPU."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from oracle import vae_oracle as vo
    from vae_eeg_xai_amd import ops
    for n_in, k, s, pad in ((40, 3, 2, "same"), (20, 3, 2, "same"), (10, 3, 2, "same"), (7, 3, 1, "valid"), (9, 5, 2, "valid")):
        if pad == "same":
            out, pb, _ = vo.same_pad(n_in, k, s)
        else:
            out, pb = (n_in - k) // s + 1, 0
        assert ops.conv_geometry(n_in, k, s, pad) == (out, pb)
    for n_in, k, s, pad in ((4, 3, 2, "same"), (8, 3, 1, "same"), (8, 3, 2, "valid"), (17, 3, 1, "valid"), (19, 3, 2, "valid"),
                            (39, 2, 1, "valid")):
        assert ops.deconv_geometry(n_in, k, s, pad) == vo.deconv_geometry(n_in, k, s, pad)[:2]
    with FakeTensorMode():
        x = torch.empty(4, 20, 20, 64, device="cuda")
        y = torch.ops.veex.conv2d(x, torch.empty(3, 3, 64, 128, device="cuda"), torch.empty(128, device="cuda"), 2, "same", True)
        assert tuple(y.shape) == (4, 10, 10, 128)
        z = torch.ops.veex.conv_transpose2d(y, torch.empty(3, 3, 64, 128, device="cuda"), torch.empty(64, device="cuda"), 2, "valid", False)
        assert tuple(z.shape) == (4, 21, 21, 64)
        dx, dw, db = torch.ops.veex.conv2d_bwd(y, x, torch.empty(3, 3, 64, 128, device="cuda"), y, 2, "same", True)
        assert tuple(dx.shape) == (4, 20, 20, 64) and tuple(dw.shape) == (3, 3, 64, 128) and tuple(db.shape) == (128,)


def test_device_array_behaves_like_the_ndarrays_analysis_py_expects():
    """analysis.py:250-264 / 323-333 index ``z_mean_blink[i, j]`` in O(N * L * quantiles) Python loops and do
    arithmetic on encoder outputs: one cached host copy, NumPy semantics (the wrapper is host logic; a CPU tensor
    stands in for the device tensor here)."""
    import time
    import torch
    from vae_eeg_xai_amd.models import DeviceArray
    a = np.random.default_rng(0).standard_normal((2700, 28)).astype(np.float32)
    d = DeviceArray(torch.from_numpy(a.copy()))
    q25, q75 = np.quantile(d, 0.25, axis=0), np.quantile(d, 0.75, axis=0)
    assert np.array_equal(q25, np.quantile(a, 0.25, axis=0))
    t0 = time.perf_counter()
    tp = 0
    for j in range(d.shape[1]):                  # the double loop of analysis.py:255-264
        for i in range(d.shape[0]):
            if d[i, j] < q25[j] or d[i, j] > q75[j]:
                tp += 1
    dt = time.perf_counter() - t0
    assert tp == int(((a < q25) | (a > q75)).sum())
    assert dt < 1.0, f"{dt:.2f} s for the 2 700 x 28 scalar-index loop"
    # arithmetic / comparisons / ufuncs
    for got, ref in ((d - 1, a - 1), (2 * d, 2 * a), (d * d, a * a), (-d, -a), (d > 0, a > 0), (np.exp(d), np.exp(a)),
                     (d / 3, a / 3), (1 - d, 1 - a), (abs(d), abs(a)), (d[:, 3] + d[:, 4], a[:, 3] + a[:, 4])):
        assert isinstance(got, np.ndarray) and np.array_equal(got, ref)
    assert d.mean() == a.mean() and d.T.shape == (28, 2700) and d.size == a.size
    # fancy indices are bounds-checked on the host (a device-side gather assert would poison the HIP context)
    e = DeviceArray(torch.from_numpy(a.copy()))
    for bad in ([0, 2700], np.array([-2701]), (np.array([5]), np.array([28]))):
        with pytest.raises(IndexError):
            e[bad]
    assert e._host is None                      # ... before anything was moved
    assert np.array_equal(e[[3, -1, 7]], a[[3, -1, 7]]) and np.array_equal(e[a[:, 0] > 0], a[a[:, 0] > 0])
    assert np.array_equal(e[np.array([1, 2]), 5], a[np.array([1, 2]), 5])
    with pytest.raises(IndexError):
        e[2700]
