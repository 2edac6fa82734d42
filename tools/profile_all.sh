#!/bin/bash
# The round's profile set (run on the GPU box): tools/profile_all.sh r05
R=${1:-r06}
tools/profile_round.sh ${R}_f32
tools/profile_round.sh ${R}_cfg3 --num-bytes 1 --sigmag
STEPS=3 WARMUP=1 tools/profile_round.sh ${R}_cfg4 --frames 128 --size 4096 --vel-steps 32 --ang-steps 2
STEPS=2 WARMUP=1 tools/profile_round.sh ${R}_cfg5 --frames 512 --size 2048 --vel-steps 64 --ang-steps 64 --num-bytes 2
tools/sq_profile.sh ${R}_f32
tools/sq_build.sh ${R}_builder 32 4096 > /dev/null
{ # the psi/phi builder variants (tools/exp_build.py T N [build_flags] [num_bytes]: 0 2-D strip, 1 separable strip, 4 general tiles)
  export TMPDIR=/tmp
  for args in "32 4096 0" "32 4096 1" "32 4096 4" "32 4096 0 2" "64 512 0" "64 512 1"; do
    rm -rf /tmp/kt3; rocprofv3 --kernel-trace --stats -d /tmp/kt3 -o r -- python tools/exp_build.py $args > /tmp/kt3.log 2>&1
    echo "== $args"; python tools/rocprof_summary.py /tmp/kt3/r_results.db | grep "kb::"
  done; } > gpurun_out/${R}_builder_variants.md 2>&1
python bench.py > gpurun_out/${R}_bench_default.json 2>/dev/null
python bench.py --separable-psf --no-cpu-baseline --no-masked --no-live-traffic 2>/dev/null | grep '^{' > gpurun_out/${R}_bench_separable_builder.json
ls gpurun_out | grep ${R}_ | head -80
# the FITS ingest kernels (tools/fits_ingest_timing.py): kernel-trace stats + the HBM-side byte counters
export TMPDIR=/tmp
rm -rf /tmp/kt_fits; rocprofv3 --kernel-trace --stats -d /tmp/kt_fits -o r -- python tools/fits_ingest_timing.py > gpurun_out/${R}_fits_timing.log 2>/tmp/kt_fits.log
python tools/rocprof_summary.py /tmp/kt_fits/r_results.db > gpurun_out/${R}_fits_kernel_stats.md
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_fits; rocprofv3 --pmc $C -d /tmp/pmc_fits -o r -- python tools/fits_ingest_timing.py > /dev/null 2>&1
  python tools/rocprof_summary.py /tmp/pmc_fits/r_results.db /tmp/pmc_fits/r_results.db | grep -E "kb_fits|counter \||---\|---\|---\|---\|---\|---" > gpurun_out/${R}_fits_pmc_$C.md
done
# round 4 additions: the multi-GPU exchange budget measured on this one GPU (cfg4 and cfg2 with a likelihood threshold), the
# RCCL path at world size 1, the issue model, the post-search kernels, StackSearch.search_all end to end, other configurations
python tools/exchange_budget.py --dense > gpurun_out/${R}_exchange_budget_cfg4.json 2>/dev/null
python tools/exchange_budget.py --counted > gpurun_out/${R}_exchange_budget_cfg4_counted.json 2>/dev/null
python tools/exchange_budget.py --frames 64 --size 512 --vel-steps 32 --ang-steps 32 --counted > gpurun_out/${R}_exchange_budget_cfg2_lh10_counted.json 2>/dev/null
python tools/exchange_budget.py --rank-flags 0 > gpurun_out/${R}_exchange_budget_cfg4_nofloor.json 2>/dev/null
python tools/exchange_budget.py --frames 64 --size 512 --vel-steps 32 --ang-steps 32 --dense > gpurun_out/${R}_exchange_budget_cfg2_lh10.json 2>/dev/null
# round 6: the dense exchange with K records per rank + repair, against the 2 K stable lists (8 ranks played on this GPU)
python tools/exchange_repair_budget.py --ang-steps 32 > gpurun_out/${R}_exchange_repair_cfg2_1024_per_rank.json 2>/dev/null
python tools/exchange_repair_budget.py --ang-steps 4 > gpurun_out/${R}_exchange_repair_cfg2_128_per_rank.json 2>/dev/null
python tools/exchange_repair_budget.py --frames 128 --size 4096 --vel-steps 32 --ang-steps 2 --reps 3 > gpurun_out/${R}_exchange_repair_cfg4.json 2>/dev/null
python tools/cold_start_timing.py > gpurun_out/${R}_cold_start_timing.log 2>&1
KBMOD_FORCE_DIST=1 python bench.py --gpus 1 --frames 128 --size 4096 --vel-steps 32 --ang-steps 2 --min-lh 10 --steps 5 --verify --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${R}_cfg4_rccl_world1_bench_line.json
KBMOD_FORCE_DIST=1 python bench.py --gpus 1 --steps 10 --verify --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${R}_cfg2_rccl_world1_bench_line.json
(cd tools/ubench && ./ubench) > gpurun_out/${R}_ubench_issue_model.log 2>&1
bash tools/profile_post.sh ${R}
{
  echo "# ms per step, kernel ms, evals/s (python bench.py --steps 20 --warmup 3 ...)"
  for ARGS in "--mask-fraction 0.01" "--inset 64" "--size 2048" "--min-lh 10" "--flags 128" "--remake-padded-copy" \
              "--frames 128 --size 4096 --vel-steps 32 --ang-steps 2 --remake-padded-copy --steps 5" \
              "--frames 512 --size 2048 --vel-steps 64 --ang-steps 64 --num-bytes 2 --flags 16 --steps 2 --warmup 1"; do
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --no-masked $ARGS 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$ARGS |', round(d['ms_per_step'], 3), '|', round(d['roofline']['kernel_ms'], 3), '|', '%.3e' % d['value'], '|', d['roofline']['kernel'])"
  done
} > gpurun_out/${R}_other_configs.log
