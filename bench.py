#!/usr/bin/env python3
"""bench.py -- VAE train samples/s on synthetic EEG topomaps (BASELINE.json metric, config[1]).

One step = one VAE.train_step (forward + ELBO + all gradients [+ RCCL all-reduce] + TF-form
Adam) over one batch of 512 (local) 40x40 topomaps that already sit in HBM.

  python bench.py --gpus 1 --steps 50 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the roofline arithmetic.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LATENT = 28
LOCAL_BATCH = 512
N_TRAIN = 100_000          # config[1]: 100k synthetic topomaps
LR = 1e-5

# algorithmic work (SURVEY 8d / BASELINE.md section 2)
FWD_MACS = 82_526_496
TRAIN_MACS = 247_356_704
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def synth(n, seed=1234):
    """SURVEY 8d generator: coarse 8x8 normal field -> cubic zoom -> disc mask -> global min-max."""
    from scipy.ndimage import zoom
    rng = np.random.default_rng(seed)
    out = np.empty((n, 40, 40, 1), dtype=np.float32)
    yy, xx = np.mgrid[0:40, 0:40]
    disc = (((yy - 19.5) ** 2 + (xx - 19.5) ** 2) <= 19.5 ** 2).astype(np.float32)
    chunk = 10000
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        coarse = rng.standard_normal((m, 8, 8))
        fine = zoom(coarse, (1, 5, 5), order=3).astype(np.float32) * disc[None]
        out[s:s + m, :, :, 0] = fine
    lo, hi = out.min(), out.max()
    out -= lo
    out /= (hi - lo)
    return out


def cpu_baseline(seconds_budget=15.0):
    """config[0] on the host cores: the torch-CPU restatement (oracle/torch_ref.py, fp32, oneDNN)
    of the reference train step, batch 64 -- kind 'port' (TensorFlow is not installable here)."""
    import torch
    from oracle import torch_ref, vae_oracle as vo
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    p = vo.init_params(LATENT, "vae", seed=42)
    x = synth(1000, seed=99)
    tr = torch_ref.TorchTrainer(p, LR, "vae", torch.float32)
    tr.train_step(x[:64])            # warm-up
    n_done, t0 = 0, time.perf_counter()
    i = 0
    while True:
        xb = x[(i * 64) % 960:(i * 64) % 960 + 64]
        tr.train_step(xb)
        n_done += xb.shape[0]
        i += 1
        if time.perf_counter() - t0 > seconds_budget or i >= 2000:
            break
    dt = time.perf_counter() - t0
    return {"value": n_done / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{i} train steps of batch 64 (config[0] shape, 1k synthetic maps) in {dt:.1f}s, torch-CPU fp32 restatement"}


ATTR_BATCH = 4096          # config[3]: attribution batches of 4096 maps
ATTR_MACS_FWD = 14_177_584   # encoder forward without the unused log-variance head
ATTR_MACS_BWD = 14_177_584   # encoder input-gradient chain incl. conv1


def attribution_bench(net, x_all, world, iters=8, warmup=2):
    """config[3]: saliency maps d z_mean[:, j] / d x back to the 40x40 electrode grid, inputs sharded
    over ranks (replicas only, no collective).  One map = one (sample, latent unit) pair.  Returns
    whole-job maps/s for k=1 (analysis.py:725 picks the top-1 unit) and for all 28 units."""
    import torch
    import torch.distributed as dist
    B = min(ATTR_BATCH, x_all.shape[0])
    nb = x_all.shape[0] // B
    out = {}
    for label, units in (("top1", [3]), ("all28", list(range(LATENT)))):
        def step(i):
            net.attribute(x_all[(i % nb) * B:(i % nb + 1) * B], units, grad_x_input=True)
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(iters):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            from vae_eeg_xai_amd import dist as vdist
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dt = float(vdist.allreduce_max_(t).item())
        maps = iters * B * len(units) * world
        flop = 2.0 * (ATTR_MACS_FWD * iters * B + ATTR_MACS_BWD * iters * B * len(units)) * world
        out[label] = {"maps_per_s": maps / dt, "ms_per_batch": 1e3 * dt / iters, "tflops": flop / dt / 1e12}
    return {"metric": "attribution maps/sec (gradient x input, latent unit -> 40x40 grid)", "unit": "maps/s",
            "value": out["top1"]["maps_per_s"], "batch_per_gpu": B, "units": 1, "top1": out["top1"], "all28": out["all28"],
            "sharding": "inputs split across ranks, replicas only"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=LOCAL_BATCH, help="local batch per GPU")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="arithmetic of the conv/deconv contractions: exact fp32 MFMA (parity path, default) or bf16 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-attribution", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
    # rehearsal on a one-GPU box: VEEX_DIST_BACKEND=gloo VEEX_DIST_ONE_GPU=1 puts every rank on cuda:0 and moves the
    # collectives through the host (checks the N > 1 control flow, not its speed)
    backend = os.environ.get("VEEX_DIST_BACKEND", "nccl")
    if os.environ.get("VEEX_DIST_ONE_GPU", "0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)

    from vae_eeg_xai_amd.net import Net
    from vae_eeg_xai_amd import roofline as rl
    from vae_eeg_xai_amd import dist as vdist

    def note(msg):     # progress on stderr (stdout carries only the JSON line)
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    B = args.batch
    note(f"world {world}, backend {backend if world > 1 else '-'}, batch {B}/GPU, {args.dtype}")
    net = Net(LATENT, "vae", compute=args.dtype)
    net.init_weights(seed=42)                      # same weights on every rank
    # weak scaling: every rank holds its own shard of the synthetic set (N_TRAIN / world maps)
    n_local = max(B, (N_TRAIN // world) // B * B)
    x_all = torch.from_numpy(synth(n_local, seed=1234 + rank)).cuda()
    nb = n_local // B
    note(f"{n_local} synthetic maps resident per rank")

    def step(i):
        xb = x_all[(i % nb) * B:(i % nb + 1) * B]
        # Philox epsilon keyed by the global sample index -> independent of the world size
        net.train_fwd_bwd(xb, None, first_sample=(i * world + rank) * B)
        if world > 1:
            vdist.allreduce_sum_(net.grads)        # one RCCL all-reduce: grads + 3 loss scalars
        net.adam_step(LR, 1.0 / world)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    note("warm-up done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dt = float(vdist.allreduce_max_(t).item())
    loss = float(net.grads[net.n_params].item()) / world
    note(f"{args.steps} steps timed: {1e3 * dt / args.steps:.3f} ms/step")
    attr = None if args.no_attribution else attribution_bench(net, x_all, world)

    if rank == 0:
        sps = args.steps * B * world / dt
        out = {
            "metric": "VAE train samples/sec", "value": sps, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "configs[1]: models.py VAE train_step, 100k synthetic 40x40 topomaps, "
                                   f"batch {B}/GPU, latent 28, Adam lr 1e-5, " +
                                   ("fp32 (exact-f32 MFMA)" if args.dtype == "f32" else
                                    "bf16 MFMA (fp32 accumulate / storage / master weights)"),
                       "global_batch": B * world, "parallelism": f"dp{world}", "final_loss": loss},
            "train_tflops": sps * TRAIN_MACS * 2 / 1e12,
            "frac_mfma_peak": sps * TRAIN_MACS * 2 / 1e12 / ((PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else 2500.0) * world),
        }
        if attr is not None:
            out["attribution"] = attr
        if not args.no_roofline:
            out["roofline"] = rl.dominant_kernel_roofline(net, B)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()          # rank 0 may still have been measuring its roofline: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
