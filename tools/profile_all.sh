#!/bin/bash
# The round's profile set (run on the GPU box): tools/profile_all.sh r03
R=${1:-r03}
tools/profile_round.sh ${R}_f32
tools/profile_round.sh ${R}_cfg3 --num-bytes 1 --sigmag
STEPS=3 WARMUP=1 tools/profile_round.sh ${R}_cfg4 --frames 128 --size 4096 --vel-steps 32 --ang-steps 2
STEPS=2 WARMUP=1 tools/profile_round.sh ${R}_cfg5 --frames 512 --size 2048 --vel-steps 64 --ang-steps 64 --num-bytes 2
tools/sq_profile.sh ${R}_f32
tools/sq_build.sh ${R}_builder 32 4096 > /dev/null
bash tools/p2.sh > gpurun_out/${R}_builder_variants.md 2>&1
python bench.py > gpurun_out/${R}_bench_default.json 2>/dev/null
ls gpurun_out | grep ${R}_ | head -80
# the FITS ingest kernels (tools/fits_ingest_timing.py): kernel-trace stats + the HBM-side byte counters
export TMPDIR=/tmp
rm -rf /tmp/kt_fits; rocprofv3 --kernel-trace --stats -d /tmp/kt_fits -o r -- python tools/fits_ingest_timing.py > gpurun_out/${R}_fits_timing.log 2>/tmp/kt_fits.log
python tools/rocprof_summary.py /tmp/kt_fits/r_results.db > gpurun_out/${R}_fits_kernel_stats.md
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_fits; rocprofv3 --pmc $C -d /tmp/pmc_fits -o r -- python tools/fits_ingest_timing.py > /dev/null 2>&1
  python tools/rocprof_summary.py /tmp/pmc_fits/r_results.db /tmp/pmc_fits/r_results.db | grep -E "kb_fits|counter \||---\|---\|---\|---\|---\|---" > gpurun_out/${R}_fits_pmc_$C.md
done
