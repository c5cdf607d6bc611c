#!/bin/bash
# GPU box: where the fp32 MFMA launches wait -- LDS conflict / FIFO counters and VMEM issue counters, one rocprofv3 --pmc
# run per pass (at most 8 SQ counters each) over tools/prof_gemms.py 512 2.   usage: bash tools/pmc_f32_stalls.sh TAG
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/lds -- python tools/prof_gemms.py 512 2 > $O/lds.log 2>&1 || { tail $O/lds.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/vmem -- python tools/prof_gemms.py 512 2 > $O/vmem.log 2>&1 || { tail $O/vmem.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/mix -- python tools/prof_gemms.py 512 2 > $O/mix.log 2>&1 || { tail $O/mix.log; exit 1; }
python - "$O" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
order = open(o + "/lds.log").read().split("launch order:")[1].split()[0].split(",")
def per_dispatch(pat):
    f = max(glob.glob(pat))
    d = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "gather_gemm" in r["Kernel_Name"] or "wgrad_" in r["Kernel_Name"]:
            d.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(d.values())
L, V, M = (per_dispatch(f"{o}/{k}/*/*counter_collection.csv") for k in ("lds", "vmem", "mix"))
reps = len(L) // len(order)
print("per launch (last repetition): counters summed over the chip; ratios against SQ_BUSY_CYCLES-independent bases")
for i, n in enumerate(order):
    l, v, m = L[reps * i + reps - 1], V[reps * i + reps - 1], M[reps * i + reps - 1]
    wc = v["SQ_WAVE_CYCLES"]
    print(f"{n:18s} lds_bank_conflict/lds_active={l['SQ_LDS_BANK_CONFLICT'] / max(l['SQ_LDS_IDX_ACTIVE'], 1):.3f} addr_conflict={l['SQ_LDS_ADDR_CONFLICT']:.3g} "
          f"data_fifo_full={l['SQ_LDS_DATA_FIFO_FULL']:.3g} cmd_fifo_full={l['SQ_LDS_CMD_FIFO_FULL']:.3g} unaligned={l['SQ_LDS_UNALIGNED_STALL']:.3g} "
          f"wait_lds/wave_cyc={l['SQ_WAIT_INST_LDS'] / wc:.3f} act_lds/wave_cyc={l['SQ_ACTIVE_INST_LDS'] / wc:.3f} | "
          f"ta_addr_full={v['SQ_VMEM_TA_ADDR_FIFO_FULL']:.3g} ta_cmd_full={v['SQ_VMEM_TA_CMD_FIFO_FULL']:.3g} vmem_rd_cyc/wave_cyc={v['SQ_INST_CYCLES_VMEM_RD'] / wc:.3f} "
          f"act_vmem/wave_cyc={v['SQ_ACTIVE_INST_VMEM'] / wc:.3f} act_valu/wave_cyc={v['SQ_ACTIVE_INST_VALU'] / wc:.3f} mfma_coexec/busy={v['SQ_VALU_MFMA_COEXEC_CYCLES'] / max(v['SQ_BUSY_CYCLES'],1):.3f} | "
          f"mfma_busy={m['SQ_VALU_MFMA_BUSY_CYCLES']:.4g} insts mfma={m['SQ_INSTS_MFMA']:.4g} lds={m['SQ_INSTS_LDS']:.4g} vmem_rd={m['SQ_INSTS_VMEM_RD']:.4g} valu={m['SQ_INSTS_VALU']:.4g} salu={m['SQ_INSTS_SALU']:.4g} gui={m['GRBM_GUI_ACTIVE']:.4g}")
PY
