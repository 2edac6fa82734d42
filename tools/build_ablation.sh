#!/bin/bash
# tools/build_ablation.sh: variant device libraries for timing ablations of kb_search_lds (not shipped)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/probe_bin
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude -Ikbmod_amd/csrc"
for v in COOP64:-DKB_COOP_MAX=64 COOP0:-DKB_COOP_MAX=0 COOP2:-DKB_COOP_MAX=2; do
  name=${v%%:*}; defs=${v#*:}
  ( /opt/rocm/bin/hipcc $FL $defs -c kbmod_amd/csrc/search_kernels.hip -o /tmp/abl_$name.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/abl_$name.o kbmod_amd/_obj/device_memory.o kbmod_amd/_obj/image_kernels.o kbmod_amd/_obj/result_kernels.o kbmod_amd/_obj/stamp_kernels.o -o tools/probe_bin/libkbmod_$name.so ) &
done
wait
ls -la tools/probe_bin/*.so
