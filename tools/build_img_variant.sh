#!/bin/bash
# tools/build_img_variant.sh NAME "-DFLAG ...": a device library whose image_kernels.hip is compiled with extra flags (timing variants)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/probe_bin
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude -Ikbmod_amd/csrc"
/opt/rocm/bin/hipcc $FL $2 -c kbmod_amd/csrc/image_kernels.hip -o /tmp/img_$1.o
objs=""; for o in kbmod_amd/_obj/*.o; do [ "$(basename $o)" = image_kernels.o ] || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/img_$1.o -o tools/probe_bin/libkbmod_$1.so
