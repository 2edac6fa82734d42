#!/bin/bash
# tools/build_variants.sh NAME:"-DKB_CHUNK=8 -DKB_LDS_SLOTS=2 -DKB_RING_WAVES=4" ...: device libraries with other
# tile geometries / tuning constants of the search kernels (search_common.h) for timing comparisons; not shipped.
# Each lands in tools/probe_bin/libkbmod_NAME.so and is picked up with KBMOD_HIP_LIB=... python bench.py
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/probe_bin
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude -Ikbmod_amd/csrc"
SRCS="search_lds search_lds_encoded search_direct search_kernels sigmag_kernels"
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  (
    for s in $SRCS; do
      /opt/rocm/bin/hipcc $FL $defs -c kbmod_amd/csrc/$s.hip -o /tmp/var_${name}_$s.o &
    done
    wait
    objs=""; for s in $SRCS; do objs="$objs /tmp/var_${name}_$s.o"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs \
      kbmod_amd/_obj/device_memory.o kbmod_amd/_obj/image_kernels.o kbmod_amd/_obj/result_kernels.o \
      kbmod_amd/_obj/stamp_kernels.o kbmod_amd/_obj/fits_kernels.o kbmod_amd/_obj/exchange_kernels.o -o tools/probe_bin/libkbmod_$name.so
    mkdir -p /tmp/objs_$name; for s in $SRCS; do cp /tmp/var_${name}_$s.o /tmp/objs_$name/$s.o; done
  ) &
done
wait
ls -la tools/probe_bin/*.so
