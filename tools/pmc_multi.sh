#!/bin/bash
# tools/pmc_multi.sh <tag> <bench args...>: several rocprofv3 PMC passes (one counter group each) of bench.py
TAG="$1"; shift
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE TCP_TCC_READ_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  tools/pmc.sh ${TAG}_$i "$grp" "$@"
  echo "== $grp"; cat gpurun_out/pmc_${TAG}_$i.md
done
