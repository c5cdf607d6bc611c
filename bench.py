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

LATENT = 28                # train.py:276; --latent sweeps it (configs[2]: 2, 8, 16, 28, 64, 128)
LOCAL_BATCH = 512          # configs[1]: batch 512 on one GPU
DP_LOCAL_BATCH = 1024      # configs[2]: global batch 8192 on 8 GPUs = 1024 per GPU
N_TRAIN = 100_000          # config[1]: 100k synthetic topomaps
LR = 1e-5


# algorithmic work (SURVEY 8d / BASELINE.md section 2): exact in-bounds MACs per sample; the latent size enters
# through the two heads (200 L) and the decoder's dense layer (4096 L)
def fwd_macs(latent):
    return 82_526_496 + (latent - 28) * (200 + 4096)


def train_macs(latent):
    return 3 * fwd_macs(latent) - 222_784      # fwd + every input gradient except conv1's + every weight gradient


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


def host_threads():
    """Threads this process may really use: its CPU affinity, capped by OMP_NUM_THREADS when the box sets it
    (os.cpu_count() shows the whole machine, of which a one-GPU box owns a share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cap = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    return max(1, min(n, cap) if cap > 0 else n)


def cpu_baseline():
    """configs[0] exactly -- 1 000 synthetic maps, batch 64 (15 x 64 + 1 x 40 steps), one
    epoch, L=28, Adam lr 1e-5 -- on the host cores with the torch-CPU fp32 restatement of the reference train step
    (oracle/torch_ref.py, oneDNN; kind 'port': TensorFlow is not installable here).  One warm-up epoch, then the
    median of 5 timed epochs."""
    import torch
    from oracle import torch_ref, vae_oracle as vo
    cores = host_threads()
    torch.set_num_threads(cores)
    p = vo.init_params(28, "vae", seed=42)
    x = synth(1000, seed=1234)
    tr = torch_ref.TorchTrainer(p, LR, "vae", torch.float32)

    def epoch():
        t0 = time.perf_counter()
        for s in range(0, 1000, 64):
            tr.train_step(x[s:s + 64])
        return time.perf_counter() - t0

    epoch()                          # warm-up
    times = sorted(epoch() for _ in range(5))
    med = times[2]
    return {"value": 1000.0 / med, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"configs[0]: 1 epoch over 1000 synthetic maps, batch 64 (15x64 + 1x40 steps), L=28, Adam lr 1e-5; "
                      f"median of 5 epochs after 1 warm-up ({med:.2f} s, min {times[0]:.2f}, max {times[-1]:.2f}); "
                      f"torch-CPU fp32 restatement on {cores} threads (os.cpu_count() = {os.cpu_count()})"}


ATTR_BATCH = 4096          # config[3]: attribution batches of 4096 maps


def attr_macs(latent):
    """Encoder forward without the unused log-variance head = encoder input-gradient chain incl. conv1
    (14 177 584 MACs at L=28; the z_mean head is 100 L of them)."""
    return 14_174_784 + 100 * latent



def attribution_bench(net, x_all, world, latent, iters=8, warmup=2):
    """config[3]: saliency maps d z_mean[:, j] / d x back to the 40x40 electrode grid, inputs sharded
    over ranks (replicas only, no collective).  One map = one (sample, latent unit) pair.  Returns
    whole-job maps/s for k=1 (analysis.py:725 picks the top-1 unit) and for all 28 units."""
    import torch
    import torch.distributed as dist
    B = min(ATTR_BATCH, x_all.shape[0])
    nb = x_all.shape[0] // B
    out = {}
    for label, units in (("top1", [min(3, latent - 1)]), ("all", list(range(latent)))):
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
        flop = 2.0 * attr_macs(latent) * (iters * B + iters * B * len(units)) * world
        out[label] = {"maps_per_s": maps / dt, "ms_per_batch": 1e3 * dt / iters, "tflops": flop / dt / 1e12}
    return {"metric": "attribution maps/sec (gradient x input, latent unit -> 40x40 grid)", "unit": "maps/s",
            "value": out["top1"]["maps_per_s"], "batch_per_gpu": B, "units": 1, "top1": out["top1"],
            f"all{latent}": out["all"], "sharding": "inputs split across ranks, replicas only"}


def engine_leg(compute, x_all, B, L, world, rank, steps, warmup):
    """The same train step on another engine, timed like the headline loop on the same resident data; the object rides
    beside `attribution`.  "bf16": configs[1] names "fp32 and bf16 MFMA" -- bf16 MFMA, fp32 accumulation, fp32 master
    weights, its own stated tolerance.  "f32": the fp32-MFMA engine (v_mfma_f32_32x32x2_f32 everywhere), the engine of
    the round-1/2 headline, which meets the same 1e-5 parity bar as the split-fp32 engine behind `value`."""
    import torch
    import torch.distributed as dist
    from vae_eeg_xai_amd.net import Net
    from vae_eeg_xai_amd import roofline as rl
    from vae_eeg_xai_amd import dist as vdist
    net = Net(L, "vae", compute=compute)
    net.init_weights(seed=42)
    nb = x_all.shape[0] // B

    def step(i):
        net.train_fwd_bwd(x_all[(i % nb) * B:(i % nb + 1) * B], None, first_sample=(i * world + rank) * B)
        if world > 1:
            vdist.allreduce_sum_(net.grads)
        net.adam_step(LR, 1.0 / world)

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dt = float(vdist.allreduce_max_(t).item())
    if rank != 0:
        return None
    sps = steps * B * world / dt
    if compute == "f32":
        out = {"metric": "VAE train samples/sec, fp32-MFMA engine", "value": sps, "unit": "samples/s", "ms_per_step": 1e3 * dt / steps,
               "steps": steps, "warmup": warmup, "dtype": "f32", "train_tflops": sps * train_macs(L) * 2 / 1e12,
               "frac_f32_mfma_peak": sps * train_macs(L) * 2 / 1e12 / (PEAK_F32_MFMA_TFLOPS * world),
               "final_loss": float(net.grads[net.n_params].item()) / world,
               "tolerance": "1e-5 of the tensor's max-abs against the float64 oracle (tests/test_gpu_net.py), as the headline engine"}
        out["roofline"] = rl.dominant_kernel_roofline(net, B)
        return out
    out = {"metric": "VAE train samples/sec, bf16 MFMA engine", "value": sps, "unit": "samples/s", "ms_per_step": 1e3 * dt / steps,
           "steps": steps, "warmup": warmup, "dtype": "bf16",
           "train_tflops": sps * train_macs(L) * 2 / 1e12,
           "frac_bf16_mfma_peak": sps * train_macs(L) * 2 / 1e12 / (2500.0 * world),
           "final_loss": float(net.grads[net.n_params].item()) / world,
           "tolerance": "bf16 operands (8-bit significand), fp32 accumulation / master weights: losses within 5e-3 relative, "
                        "activations within 2e-2 and gradients within 5e-2 of the tensor's max-abs against the fp32 path "
                        "(tests/test_gpu_bf16.py, tests/test_gpu_fullsize.py)"}
    # every MFMA launch of the step stand-alone against its own roof = max(FLOP / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s)
    rows = rl.layer_table(B, iters=10, bf16=True)
    dom = max(rows, key=lambda r: r["us"])
    out["dominant_launch"] = {"launch": dom["kernel"], "us": dom["us"], "roof_us": dom["roof_us"], "bound": dom["roof"],
                              "frac_of_roof": dom["frac_roof"], "tflops": dom["tflops"], "algorithmic_bytes": dom["bytes"]}
    out["launches"] = {r["kernel"]: {"us": round(r["us"], 1), "frac_of_roof": round(r["frac_roof"], 3), "bound": r["roof"]} for r in rows}
    out["sum_us"] = sum(r["us"] for r in rows)
    out["sum_roof_us"] = sum(r["roof_us"] for r in rows)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0,
                    help="local batch per GPU (default: 512 on one GPU = configs[1]; 1024 with --gpus > 1 = configs[2], "
                         "global batch 8192 on 8 GPUs)")
    ap.add_argument("--latent", type=int, default=LATENT, help="latent dimension (configs[2] sweeps 2, 8, 16, 28, 64, 128)")
    ap.add_argument("--dtype", choices=["f32", "f32x3", "bf16"], default="f32x3",
                    help="engine of the conv/deconv contractions: f32x3 = fp32 results on the bf16 matrix cores (three-term "
                         "operand split, six partial products, fp32 accumulate; default, 1e-5 parity), f32 = fp32 MFMA "
                         "instructions everywhere (1e-5 parity), bf16 = bf16 MFMA on rounded operands (own tolerance)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-attribution", action="store_true")
    ap.add_argument("--buckets", type=int, choices=[1, 2], default=2,
                    help="N > 1: 2 = decoder gradients all-reduced on a communication stream while the encoder's backward "
                         "pass runs (dist.TwoBucketReducer), 1 = one all-reduce after the backward pass")
    ap.add_argument("--no-bf16", action="store_true", help="skip the other engines' legs (the `bf16` and `f32_mfma` objects of the JSON line)")
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

    B = args.batch if args.batch > 0 else (LOCAL_BATCH if world == 1 else DP_LOCAL_BATCH)
    L = args.latent
    TRAIN_MACS = train_macs(L)
    note(f"world {world}, backend {backend if world > 1 else '-'}, batch {B}/GPU, latent {L}, {args.dtype}")
    net = Net(L, "vae", compute=args.dtype)
    net.init_weights(seed=42)                      # same weights on every rank
    # weak scaling: every rank holds its own shard of the synthetic set (N_TRAIN / world maps)
    n_local = max(B, (N_TRAIN // world) // B * B)
    x_all = torch.from_numpy(synth(n_local, seed=1234 + rank)).cuda()
    nb = n_local // B
    note(f"{n_local} synthetic maps resident per rank")

    red = vdist.TwoBucketReducer(net) if (world > 1 and args.buckets == 2) else None

    def step(i, exchange=True):
        xb = x_all[(i % nb) * B:(i % nb + 1) * B]
        # Philox epsilon keyed by the global sample index -> independent of the world size
        net.train_fwd_bwd(xb, None, first_sample=(i * world + rank) * B)
        if world > 1 and exchange:
            if red is not None:
                red.reduce_()                      # decoder bucket behind its event on the comm stream, the rest here
            else:
                vdist.allreduce_sum_(net.grads)    # one RCCL all-reduce: grads + 3 loss scalars
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
    local_ms = None
    if world > 1:
        # the same local work without the exchange, on this rank alone: what one GPU does at this batch (the driver's
        # N=1 line is configs[1] at batch 512, so it is not the equal-work single-GPU number of this line)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(i, exchange=False)
        torch.cuda.synchronize()
        local_ms = 1e3 * (time.perf_counter() - t1) / args.steps
        dist.barrier()
    if red is not None:
        red.close()
    attr = None if args.no_attribution else attribution_bench(net, x_all, world, L)
    bf16 = f32_mfma = None
    if args.dtype != "bf16" and not args.no_bf16:
        note("bf16 leg")
        bf16 = engine_leg("bf16", x_all, B, L, world, rank, args.steps, args.warmup)
        if args.dtype == "f32x3":
            note("fp32-MFMA leg")
            f32_mfma = engine_leg("f32", x_all, B, L, world, rank, args.steps, args.warmup)

    if rank == 0:
        sps = args.steps * B * world / dt
        out = {
            "metric": "VAE train samples/sec", "value": sps, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "f32",
            "data": "synthetic",
            "config": {"workload": (f"configs[1]: models.py VAE train_step, 100k synthetic 40x40 topomaps, batch {B}/GPU, "
                                    if world == 1 else
                                    f"configs[2]: data-parallel models.py VAE train_step, global batch {B * world} "
                                    f"({B}/GPU x {world} GPUs; 8192 on 8), the flat gradient buffer all-reduced over RCCL once per "
                                    f"step ({'two buckets: the decoder share on a communication stream under the encoder backward' if args.buckets == 2 else 'one collective'}), "
                                    f"{n_local} synthetic 40x40 topomaps resident per GPU, ") +
                                   f"latent {L}, Adam lr 1e-5, " +
                                   {"f32": "fp32 (fp32 MFMA instructions)",
                                    "f32x3": "fp32 results, split-fp32 engine: the conv / deconv forward, input-gradient and "
                                             "weight-gradient GEMMs on the bf16 matrix cores from three-term operand splits (six "
                                             "partial products, fp32 accumulate), the small layers on the fp32 MFMA; 1e-5 parity bar",
                                    "bf16": "bf16 MFMA (fp32 accumulate / storage / master weights)"}[args.dtype],
                       "engine": args.dtype,
                       "global_batch": B * world, "local_batch": B, "latent": L, "parallelism": f"dp{world}",
                       "final_loss": loss},
            "train_tflops": sps * TRAIN_MACS * 2 / 1e12,
            "frac_mfma_peak": sps * TRAIN_MACS * 2 / 1e12 / ((2500.0 if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS) * world),
        }
        if local_ms is not None:
            out["local_step_ms_without_allreduce"] = local_ms
            out["allreduce_exposed_ms"] = out["ms_per_step"] - local_ms     # what the exchange adds to a step on this rank
            out["config"]["grad_buckets"] = args.buckets
            out["config"]["equal_work_single_gpu"] = (
                f"local_step_ms_without_allreduce: the same {B}-sample step on one GPU with no exchange; the driver's N=1 "
                f"line is configs[1] at batch {LOCAL_BATCH}/GPU and is NOT the equal-work denominator of this line")
        if attr is not None:
            out["attribution"] = attr
        if bf16 is not None:
            out["bf16"] = bf16
        if f32_mfma is not None:
            out["f32_mfma"] = f32_mfma
        if args.dtype == "f32x3":
            out["frac_mfma_peak_note"] = ("fp32 FLOP of the step / 157.3 TFLOP/s, the fp32 MFMA instruction's peak; the 20 large "
                                          "GEMM launches run on the bf16 pipe instead (roofline.peak_note)")
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
