"""Condense gpurun_out/<round>/ (tools/run_profiles.sh) into the tracked files under profiles/."""
import collections, csv, glob, json, os, shutil, sys


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


R = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = f"gpurun_out/{R}", "profiles"
os.makedirs(dst, exist_ok=True)
shutil.copy(f"{src}/bench.json", f"{dst}/{R}_bench.json")
shutil.copy(f"{src}/layer_table.txt", f"{dst}/{R}_layer_table.txt")
for extra in ("bench_bf16.json", "layer_table_bf16.txt"):
    if os.path.exists(f"{src}/{extra}"):
        shutil.copy(f"{src}/{extra}", f"{dst}/{R}_{extra}")
ks = newest(f"{src}/trace/*/*kernel_stats.csv")
shutil.copy(ks, f"{dst}/{R}_kernel_stats.csv")
with open(f"{dst}/{R}_kernel_stats.txt", "w") as f:
    f.write("rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attribution\n")
    for r in csv.DictReader(open(ks)):
        f.write(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:9.1f} pct={float(r['Percentage']):6.2f}\n")


def per_dispatch(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [(k, v) for k, v in d.items() if "gather_gemm" in k[1] or "wgrad_" in k[1]]


order = open(f"{src}/pmc_fetch.log").read().split("launch order:")[1].split()[0].split(",")
f_ = per_dispatch(newest(f"{src}/pmc_fetch/*/*counter_collection.csv"))
w_ = per_dispatch(newest(f"{src}/pmc_write/*/*counter_collection.csv"))
s_ = per_dispatch(newest(f"{src}/pmc_sq/*/*counter_collection.csv"))
reps = len(f_) // len(order)
out = {}
with open(f"{dst}/{R}_pmc_gemms.txt", "w") as f:
    f.write("separate rocprofv3 --pmc passes over tools/prof_gemms.py 512 (last repetition of each launch)\n"
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
json.dump(out, open(f"{dst}/{R}_pmc_gemms.json", "w"), indent=1)
print(open(f"{dst}/{R}_pmc_gemms.txt").read())
