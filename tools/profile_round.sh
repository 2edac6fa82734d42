#!/bin/bash
# [STEPS=n WARMUP=m] tools/profile_round.sh <tag> [bench args]: rocprofv3 kernel-trace stats + FETCH_SIZE / WRITE_SIZE /
# TCC passes of bench.py; writes gpurun_out/<tag>_*.md (small) for profiles/.
TAG="$1"; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
B="python bench.py --steps ${STEPS:-5} --warmup ${WARMUP:-2} --no-cpu-baseline --no-live-traffic --no-masked $@"
rm -rf /tmp/kt_$TAG; rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o r -- $B > gpurun_out/${TAG}_bench.json 2>/tmp/kt_$TAG.log
python tools/rocprof_summary.py /tmp/kt_$TAG/r_results.db > gpurun_out/${TAG}_kernel_stats.md
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/pmc_$TAG; rocprofv3 --pmc $C -d /tmp/pmc_$TAG -o r -- $B > /dev/null 2>&1
  python tools/rocprof_summary.py /tmp/pmc_$TAG/r_results.db /tmp/pmc_$TAG/r_results.db | grep -E "kb::|counter \||---\|---\|---\|---\|---\|---" > gpurun_out/${TAG}_pmc_$N.md
done
grep -h '"metric"' gpurun_out/${TAG}_bench.json | tail -1 > gpurun_out/${TAG}_bench_line.json
