#!/bin/bash
# tools/profile_post.sh <tag>: the post-search kernels of SURVEY 8(f) (tools/bench_post_search.py: sigma-G matrix, coadds,
# grid filter, filter + sort at 2 M rows) under rocprofv3 -- kernel stats and separate FETCH_SIZE / WRITE_SIZE passes -- and
# the end-to-end time of StackSearch.search_all; writes gpurun_out/<tag>_post_*.{md,log} for profiles/.
TAG="$1"
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/bench_post_search.py > gpurun_out/${TAG}_post_bench.log 2>/dev/null
rm -rf /tmp/kt_post; rocprofv3 --kernel-trace --stats -d /tmp/kt_post -o r -- python tools/bench_post_search.py > /dev/null 2>/tmp/kt_post.log
python tools/rocprof_summary.py $(find /tmp/kt_post -name "*_results.db" | head -1) > gpurun_out/${TAG}_post_kernel_stats.md
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_post; rocprofv3 --pmc $C -d /tmp/pmc_post -o r -- python tools/bench_post_search.py > /dev/null 2>&1
  DB=$(find /tmp/pmc_post -name "*_results.db" | head -1)
  python tools/rocprof_summary.py $DB $DB | grep -E "kb::|rocprim|counter \||---\|---\|---\|---\|---\|---" > gpurun_out/${TAG}_post_pmc_$C.md
done
{ python tools/search_all_timing.py; echo "# 64 x 2048 x 2048, 64 candidates"; python tools/search_all_timing.py 64 2048 2048 32 2;
  echo "# 128 x 4096 x 4096, 64 candidates"; python tools/search_all_timing.py 128 4096 4096 32 2; } 2>/dev/null | grep -E "^#|min_lh" > gpurun_out/${TAG}_search_all_timing.log
