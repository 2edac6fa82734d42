#!/bin/bash
# SQ / TCC counter passes of the psi/phi builder (tools/exp_build.py T N): gpurun_out/<tag>_sq.md
TAG="$1"; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/${TAG}_sq.md
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/sqb_$TAG
  rocprofv3 --pmc $grp -d /tmp/sqb_$TAG -o r -- python tools/exp_build.py "$@" > /tmp/sqb_$TAG.log 2>&1
  python tools/rocprof_summary.py /tmp/sqb_$TAG/r_results.db /tmp/sqb_$TAG/r_results.db | grep -E "kb_psi_phi|kb_conv" >> gpurun_out/${TAG}_sq.md
done
cat gpurun_out/${TAG}_sq.md
