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
