#!/bin/bash
# GPU box: separate rocprofv3 --pmc passes over the 21 bf16 MFMA launches of one train step (batch 512).
# Each pass is a run of its own (FETCH_SIZE and WRITE_SIZE do not fit one pass); stop at the first pass that fails.
set -e
export TMPDIR=/tmp
R=${1:-r02}
O=gpurun_out/$R
mkdir -p $O
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_bf16 -- python tools/prof_gemms.py 512 2 bf16 > $O/pmc_fetch_bf16.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_bf16 -- python tools/prof_gemms.py 512 2 bf16 > $O/pmc_write_bf16.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_bf16 -- python tools/prof_gemms.py 512 2 bf16 > $O/pmc_sq_bf16.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_inst_bf16 -- python tools/prof_gemms.py 512 2 bf16 > $O/pmc_inst_bf16.log 2>&1
timeout -k 10 120 python -m vae_eeg_xai_amd.roofline 512 bf16 > $O/layer_table_bf16.txt 2>/dev/null
echo PMC_BF16_DONE
