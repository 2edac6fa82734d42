#!/bin/bash
# builds (here or on the GPU box) and runs the micro-benchmarks
cd "$(dirname "$0")"
[ -x ./ubench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 ubench.hip -o ubench
./ubench
