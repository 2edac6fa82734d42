#!/bin/bash
# tools/pmc.sh <tag> "<counters>" <bench args...> : one rocprofv3 PMC pass of bench.py, summarised into
# gpurun_out/pmc_<tag>.md (the raw rocpd database stays in /tmp on the GPU box).
TAG="$1"; CTRS="$2"; shift; shift
export TMPDIR=/tmp
OUT=/tmp/prof_$TAG
rm -rf $OUT; mkdir -p gpurun_out
rocprofv3 --pmc $CTRS -d $OUT -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /tmp/prof_$TAG.log 2>&1
python tools/rocprof_summary.py $OUT/r_results.db $OUT/r_results.db | grep -E "kb::|counter|---" > gpurun_out/pmc_$TAG.md
