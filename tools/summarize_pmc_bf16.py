"""Condense the bf16 rocprofv3 --pmc passes of tools/pmc_bf16.sh (gpurun_out/<round>/) into profiles/<round>_pmc_gemms_bf16.{txt,json}."""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = f"gpurun_out/{R}", "profiles"


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def per_dispatch(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [(k, v) for k, v in d.items() if "gather_gemm" in k[1] or "wgrad_" in k[1]]


from vae_eeg_xai_amd import roofline as rl   # noqa: E402  (algorithmic bytes only; no GPU needed)
macs = {n: m for n, *_, m in rl.CONVS + rl.DECONVS}
order = open(f"{src}/pmc_fetch_bf16.log").read().split("launch order:")[1].split()[0].split(",")
f_ = per_dispatch(newest(f"{src}/pmc_fetch_bf16/*/*counter_collection.csv"))
w_ = per_dispatch(newest(f"{src}/pmc_write_bf16/*/*counter_collection.csv"))
s_ = per_dispatch(newest(f"{src}/pmc_sq_bf16/*/*counter_collection.csv"))
i_ = per_dispatch(newest(f"{src}/pmc_inst_bf16/*/*counter_collection.csv"))
times = {}
for line in open(f"{src}/layer_table_bf16.txt"):
    p = line.split()
    if len(p) > 2 and p[2] == "us":
        times[p[0]] = float(p[1])
reps = len(f_) // len(order)
out = {}
with open(f"{dst}/{R}_pmc_gemms_bf16.txt", "w") as f:
    f.write("bf16 engine, batch 512: separate rocprofv3 --pmc passes over tools/prof_gemms.py 512 2 bf16 (last repetition of each launch), tools/pmc_bf16.sh\n"
            "hbm_read = FETCH_SIZE KB * 1024 * 2 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B); hbm_write = WRITE_SIZE KB * 1024; "
            "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CU * 4 SIMD); */mfma = instructions per MFMA instruction;\n"
            "alg = algorithmic bytes (roofline.algorithmic_bytes); roof = max(FLOP / 2.5 PFLOP/s, alg / 8 TB/s); us = HIP-event time (roofline.py 512 bf16, un-profiled)\n")
    for i, name in enumerate(order):
        j = reps * i + reps - 1
        fe, wr, sq, ins = f_[j][1]["FETCH_SIZE"] * 1024 * 2, w_[j][1]["WRITE_SIZE"] * 1024, s_[j][1], i_[j][1]
        gui = sq["GRBM_GUI_ACTIVE"] / 8
        nm = max(ins["SQ_INSTS_MFMA"], 1.0)
        alg = rl.algorithmic_bytes(name, 512, True)
        layer = name.rsplit(".", 1)[0]
        roof_us = max(2.0 * macs[layer] * 512 / 2.5e15, alg / 8e12) * 1e6
        us = times.get(name, float("nan"))
        out[name] = {"kernel": f_[j][0][1], "us": us, "hbm_read_bytes": fe, "hbm_write_bytes": wr, "algorithmic_bytes": alg,
                     "roof_us": roof_us, "frac_roof": roof_us / us,
                     "mfma_util": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 256 * 4),
                     "wait_any": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], "wait_inst": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                     "valu_per_mfma": ins["SQ_INSTS_VALU"] / nm, "salu_per_mfma": ins["SQ_INSTS_SALU"] / nm,
                     "lds_per_mfma": ins["SQ_INSTS_LDS"] / nm, "vmem_rd_per_mfma": ins["SQ_INSTS_VMEM_RD"] / nm,
                     "vmem_wr_per_mfma": ins["SQ_INSTS_VMEM_WR"] / nm}
        o = out[name]
        f.write(f"{name:18s} us={us:6.1f} roof={roof_us:5.1f} ({100 * o['frac_roof']:4.1f}%) alg={alg / 1e6:6.1f} MB hbm_read={fe / 1e6:6.1f} hbm_write={wr / 1e6:6.1f} "
                f"mfma_util={o['mfma_util']:.3f} wait_any={o['wait_any']:.2f} wait_inst={o['wait_inst']:.2f} valu/mfma={o['valu_per_mfma']:5.1f} "
                f"salu/mfma={o['salu_per_mfma']:5.1f} lds/mfma={o['lds_per_mfma']:4.1f} vmem_rd/mfma={o['vmem_rd_per_mfma']:4.2f} vmem_wr/mfma={o['vmem_wr_per_mfma']:4.2f}  "
                f"{f_[j][0][1][:48]}\n")
json.dump(out, open(f"{dst}/{R}_pmc_gemms_bf16.json", "w"), indent=1)
print(open(f"{dst}/{R}_pmc_gemms_bf16.txt").read())
