"""Pin parity against the REAL reference (never run in this repository: TensorFlow is not installable here).
For someone with TF 2.13 + Keras 2.13 and a checkout of alessandroquirile/VAE-EEG-XAI:
    python tools/dump_tf_train_step.py /path/to/VAE-EEG-XAI tests/golden/vae_L28_B2_s11.npz out_tf.npz
runs the reference's own models.py (train_step, models.py:148-180) on the fixture's weights (oracle seed), inputs
and epsilon, and writes the fixture's keys; `python tests/golden/compare_fixture.py out_tf.npz <fixture>` then
compares at 1e-5.  See INTEGRATION.md "Pinning parity"."""
import os, sys
import numpy as np
ref_dir, fixture, out_path = sys.argv[1:4]
sys.path[:0] = [ref_dir, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import tensorflow as tf, models                      # the reference's models.py
from oracle import vae_oracle as vo
variant, L, B, seed = os.path.basename(fixture)[:-4].split("_"); L, seed = int(L[1:]), int(seed[1:])
f = np.load(fixture); x, eps = f["x"], f["eps"]
p = vo.init_params(L, "vae", seed=seed, jitter=0.1)
models.tf.random.normal = lambda shape, **kw: tf.constant(eps)          # inject the fixture's epsilon into sample()
vae = models.VAE(models.Encoder(L), models.Decoder()); vae.compile(tf.keras.optimizers.legacy.Adam(1e-3))
vae(x)                                               # create the variables
own = [p[n].astype("float32") for n, _, _ in vo.param_spec(L, "vae")]  # get_weights() order assumed by SURVEY A.6
cur = vae.get_weights(); assert [w.shape for w in cur[:len(own)]] == [w.shape for w in own], "get_weights() order differs"
vae.set_weights(own + cur[len(own):])
with tf.GradientTape() as tape:                      # the same expressions as models.py:152-163, to read the gradients
    zm, zlv, z = vae.encoder(x); rec = vae.decoder(z)
    rl = tf.reduce_mean(tf.reduce_sum(tf.keras.losses.binary_crossentropy(x, rec), axis=(1, 2)))
    kl = tf.reduce_mean(tf.reduce_sum(-0.5 * (1 + zlv - tf.square(zm) - tf.exp(zlv)), axis=1)); tot = rl + kl
g = tape.gradient(tot, vae.trainable_weights)
out = {"x": x, "eps": eps, "total": tot.numpy(), "rec_loss": rl.numpy(), "kl_loss": kl.numpy(), "recon": rec.numpy(),
       "z": z.numpy(), "z_mean": zm.numpy(), "z_log_var": zlv.numpy()}
for n, gi in zip(vo.trainable_names(L, "vae"), g):
    out["gsum." + n], out["gabs." + n], out["ghead." + n] = gi.numpy().sum(), np.abs(gi.numpy()).sum(), gi.numpy().ravel()[:16]
for t in (1, 2, 3):                                  # three real train_step calls (models.py:148-180), Adam lr 1e-3
    out[f"loss_step{t}"] = float(vae.train_on_batch(x, x, return_dict=True)["loss"])
np.savez(out_path, **out)
