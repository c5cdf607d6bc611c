"""Condense gpurun_out/<round>/ (tools/run_profiles.sh) into the tracked files under profiles/."""
import collections, csv, glob, json, os, shutil, sys


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


R = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = f"gpurun_out/{R}", "profiles"
os.makedirs(dst, exist_ok=True)
shutil.copy(f"{src}/bench.json", f"{dst}/{R}_bench.json")
shutil.copy(f"{src}/layer_table.txt", f"{dst}/{R}_layer_table.txt")
for extra in ("bench_bf16.json", "layer_table_bf16.txt", "hbm_table.txt", "dp2_rehearsal.json", "latent_sweep_b1024.jsonl",
              "bench_batches.jsonl", "bench_bf16_batches.jsonl", "layer_table_bf16_b4096.txt"):
    if os.path.exists(f"{src}/{extra}"):
        if extra.startswith(("dp2_", "latent_sweep", "bench_batches", "bench_bf16_batches")):   # one JSON line per run; drop what gloo / the launcher printed beside it
            open(f"{dst}/{R}_{extra}", "w").writelines(l for l in open(f"{src}/{extra}") if l.startswith("{"))
        else:
            shutil.copy(f"{src}/{extra}", f"{dst}/{R}_{extra}")
ks = newest(f"{src}/trace/*/*kernel_stats.csv")
shutil.copy(ks, f"{dst}/{R}_kernel_stats.csv")
with open(f"{dst}/{R}_kernel_stats.txt", "w") as f:
    f.write("rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attribution --no-bf16   (the default engine: split-fp32)\n")
    for r in csv.DictReader(open(ks)):
        f.write(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:9.1f} pct={float(r['Percentage']):6.2f}\n")


# the same trace split by launch shape: one kernel name serves several layers, the grid tells them apart
# (gather_gemm_v3_kernel<true, 1> with 11552 x 256 threads is dec.deconv4 forward at batch 512, bench.py's `roofline` kernel)
kt = newest(f"{src}/trace/*/*kernel_trace.csv")
by = collections.OrderedDict()
for r in csv.DictReader(open(kt)):
    if "gather_gemm" in r["Kernel_Name"] or "wgrad_" in r["Kernel_Name"]:
        by.setdefault((r["Kernel_Name"][:60], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])), []).append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(f"{dst}/{R}_kernel_stats_by_launch.txt", "w") as f:
    f.write("same rocprofv3 --kernel-trace run, MFMA launches grouped by (kernel, workgroups): with the side stream on, a launch "
            "that shares the GPU with a weight-gradient GEMM is stretched; see *_seq_f32.txt for undisturbed times\n")
    for (name, blocks), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        f.write(f"{name:60s} workgroups={blocks:6d} calls={len(v):4d} avg_us={sum(v) / len(v):8.1f} min_us={min(v):8.1f}\n")

for extra, hdr in (("seq_f32.txt", "fp32"), ("seq_bf16.txt", "bf16")):
    if os.path.exists(f"{src}/{extra}"):
        with open(f"{dst}/{R}_{extra}", "w") as f:
            f.write(f"VEEX_OVERLAP=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 ... ({hdr}; "
                    "side stream off, so no kernel's duration is stretched by a concurrent one; tools/prof_seq.sh)\n")
            f.write(open(f"{src}/{extra}").read())
if os.path.exists(f"{src}/micro_mfma_issue.txt"):
    shutil.copy(f"{src}/micro_mfma_issue.txt", f"{dst}/{R}_micro_mfma_issue.txt")
att = glob.glob(f"{src}/attr/*/*kernel_stats.csv")
if att:
    with open(f"{dst}/{R}_attr_kernel_stats.txt", "w") as f:
        f.write("rocprofv3 --kernel-trace --stats -- python tools/prof_attr.py   (attribution, batch 4096: 3 x [unit 3], 3 x [all 28 units])\n")
        for r in csv.DictReader(open(max(att, key=os.path.getmtime))):
            f.write(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:9.1f} pct={float(r['Percentage']):6.2f}\n")


def per_dispatch(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [(k, v) for k, v in d.items() if "gather_gemm" in k[1] or "wgrad_" in k[1]]


def pmc_summary(sfx, what):
    """gpurun_out/<R>/pmc_{fetch,write,sq}<sfx> -> profiles/<R>_pmc_gemms<sfx>.{txt,json}; returns the per-launch dict"""
    if not os.path.exists(f"{src}/pmc_fetch{sfx}.log"):
        return {}
    order = open(f"{src}/pmc_fetch{sfx}.log").read().split("launch order:")[1].split()[0].split(",")
    f_ = per_dispatch(newest(f"{src}/pmc_fetch{sfx}/*/*counter_collection.csv"))
    w_ = per_dispatch(newest(f"{src}/pmc_write{sfx}/*/*counter_collection.csv"))
    s_ = per_dispatch(newest(f"{src}/pmc_sq{sfx}/*/*counter_collection.csv"))
    reps = len(f_) // len(order)
    out = {}
    with open(f"{dst}/{R}_pmc_gemms{sfx}.txt", "w") as f:
        f.write(f"{what}: separate rocprofv3 --pmc passes over tools/prof_gemms.py 512 2 (last repetition of each launch)\n"
                "hbm_read = FETCH_SIZE KB * 1024 * 2 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM); "
                "hbm_write = WRITE_SIZE KB * 1024; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CU * 4 SIMD)\n")
        for i, name in enumerate(order):
            j = reps * i + reps - 1
            fe, wr, sq = f_[j][1]["FETCH_SIZE"] * 1024 * 2, w_[j][1]["WRITE_SIZE"] * 1024, s_[j][1]
            gui = sq["GRBM_GUI_ACTIVE"] / 8
            util = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 256 * 4)
            out[name] = {"kernel": f_[j][0][1], "hbm_read_bytes": fe, "hbm_write_bytes": wr, "mfma_util": util,
                         "gui_cycles": gui, "wait_any": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                         "wait_inst": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]}
            f.write(f"{name:20s} hbm_read={fe / 1e6:8.1f} MB hbm_write={wr / 1e6:8.1f} MB mfma_util={util:.3f} "
                    f"gui_cycles={gui:.4g} wait_any={out[name]['wait_any']:.2f} wait_inst={out[name]['wait_inst']:.2f}  {f_[j][0][1][:70]}\n")
    json.dump(out, open(f"{dst}/{R}_pmc_gemms{sfx}.json", "w"), indent=1)
    print(open(f"{dst}/{R}_pmc_gemms{sfx}.txt").read())
    return out


pmc = {"f32": pmc_summary("", "fp32-MFMA engine"), "f32x3": pmc_summary("_x3", "split-fp32 engine")}
for extra in ("layer_table_x3.txt", "bench_f32_mfma.json"):
    if os.path.exists(f"{src}/{extra}"):
        shutil.copy(f"{src}/{extra}", f"{dst}/{R}_{extra}")
if os.path.exists(f"{src}/seq_x3.txt"):
    with open(f"{dst}/{R}_seq_x3.txt", "w") as f:
        f.write("VEEX_OVERLAP=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 ... (split-fp32 engine, the default; "
                "side stream off, so no kernel's duration is stretched by a concurrent one; tools/prof_seq.sh)\n")
        f.write(open(f"{src}/seq_x3.txt").read())
# bench.py ran before this run's counter passes were condensed, so its `traffic` came from the PREVIOUS summary in
# profiles/: put this run's own figure in its place (same box, same build) and say so.
bj = f"{dst}/{R}_bench.json"
if os.path.exists(bj):
    lines = open(bj).read().strip().split("\n")
    d = json.loads(lines[-1])
    eng = d.get("config", {}).get("engine", "f32")
    for roof, out in ((d.get("roofline"), pmc.get(eng, {})), ((d.get("f32_mfma") or {}).get("roofline"), pmc["f32"])):
        if isinstance(roof, dict) and "dec.deconv4.fwd" in out:
            e = out["dec.deconv4.fwd"]
            roof["traffic"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
            roof["traffic_source"] = (roof.get("traffic_source") or "") + \
                " [filled in by tools/summarize_profiles.py from the counter passes of the same run_profiles.sh call]"
    lines[-1] = json.dumps(d)
    open(bj, "w").write("\n".join(lines) + "\n")
