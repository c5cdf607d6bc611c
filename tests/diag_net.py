"""GPU diagnostic: every intermediate and gradient of one train step vs the float64 oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import vae_oracle as vo
from vae_eeg_xai_amd.net import Net

def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)

def run(variant, L, B, jitter=0.1):
    print(f"==== variant={variant} L={L} B={B}")
    p = vo.init_params(L, variant, seed=3, jitter=jitter)
    x = vo.synth_topomaps(B, 5)
    eps = np.random.default_rng(7).standard_normal((B, L)).astype(np.float32) if variant == "vae" else None
    c = vo.forward(p, x, eps, variant)
    net = Net(L, variant)
    net.set_weights_dict(p)
    losses = net.train_fwd_bwd(x, eps)
    torch.cuda.synchronize()
    # ReLU decisions of the device; disagreements must sit at fp32-rounding kinks
    masks = {}
    layers = ["enc.conv1", "enc.conv2", "enc.conv3", "enc.dense", "dec.dense"] + [f"dec.deconv{i}" for i in range(1, 6)]
    for ln in layers:
        ref_pre = c[ln + ".pre"]
        got = net.ws_region(B, "act." + ln).cpu().numpy().reshape(ref_pre.shape)
        m = got > 0
        dis = m != (ref_pre > 0)
        if dis.any():
            print(f"  mask {ln}: {int(dis.sum())} disagreements, max |pre_ref| there = {np.abs(ref_pre[dis]).max():.3e} (scale {np.abs(ref_pre).max():.3e})")
        masks[ln] = m
    g = vo.backward(p, c, variant, masks)
    bad = 0
    def chk(name, got, ref, tol=2e-5):
        nonlocal bad
        r = rel(got, ref)
        flag = "" if r < tol else "   <<<<<< FAIL"
        if flag: bad += 1
        print(f"  {name:28s} rel={r:.3e} max|ref|={np.abs(ref).max():.3e}{flag}")
    def ws(name, shape):
        return net.ws_region(B, name).cpu().numpy().reshape(shape)
    for nm in ["conv1", "conv2", "conv3"]:
        ref = c[f"enc.{nm}.a"]; chk(f"act.enc.{nm}", ws(f"act.enc.{nm}", ref.shape), ref)
    chk("act.enc.dense", ws("act.enc.dense", c["enc.dense.a"].shape), c["enc.dense.a"])
    if variant == "vae":
        mulv = ws("mulv", (B, 2 * L))
        chk("z_mean", mulv[:, :L], c["z_mean"]); chk("z_log_var", mulv[:, L:], c["z_log_var"])
        chk("z", ws("z", (B, L)), c["z"])
    chk("act.dec.dense", ws("act.dec.dense", (B, 4096)), c["dec.dense.a"])
    for i in range(1, 6):
        ref = c[f"dec.deconv{i}.a"]; chk(f"act.dec.deconv{i}", ws(f"act.dec.deconv{i}", ref.shape), ref)
    chk("dlogit", ws("dlogit", (B, 40, 40, 1)), g["dlogits"])
    l = losses.cpu().numpy()
    chk("loss.total", l[0], c["total"]); chk("loss.rec", l[1], c["rec_loss"])
    if variant == "vae": chk("loss.kl", l[2], c["kl_loss"])
    for i in range(5, 0, -1):
        ref = g[f"dpre.dec.deconv{i}"]; chk(f"dpre.dec.deconv{i}", ws(f"dpre.dec.deconv{i}", ref.shape), ref)
    chk("dz", ws("dz", (B, L)), g["dz"])
    if variant == "vae":
        dm = ws("dmulv", (B, 2 * L)); chk("dz_mean", dm[:, :L], g["dz_mean"]); chk("dz_log_var", dm[:, L:], g["dz_log_var"])
    for nm in ["conv3", "conv2", "conv1"]:
        ref = g[f"dpre.enc.{nm}"]; chk(f"dpre.enc.{nm}", ws(f"dpre.enc.{nm}", ref.shape), ref)
    gd = net.get_weights_dict(net.grads)
    for k, v in gd.items():
        chk("grad " + k, v, g[k])
    print(f"  -> {bad} failures")
    return bad

if __name__ == "__main__":
    bad = 0
    bad += run("vae", 28, 4)
    bad += run("vae", 28, 70)
    bad += run("ae", 28, 6)
    bad += run("vae", 2, 5)
    print("TOTAL FAILURES", bad)
    sys.exit(1 if bad else 0)
