#!/bin/bash
# tools/sq_profile.sh <tag> <bench args...>: SQ / LDS / TCC counter passes of the search kernel (one
# rocprofv3 --pmc pass per group; no tracing in the same run), summarised into gpurun_out/<tag>_sq.md
TAG="$1"; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/${TAG}_sq.md
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/sq_$TAG
  rocprofv3 --pmc $grp -d /tmp/sq_$TAG -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-masked "$@" > /tmp/sq_$TAG.log 2>&1
  python tools/rocprof_summary.py /tmp/sq_$TAG/r_results.db /tmp/sq_$TAG/r_results.db | grep -E "kb_search|kb_sigmag" >> gpurun_out/${TAG}_sq.md
done
