#!/bin/bash
# tools/build_asm_variant.sh NAME "KB_GEN_NO_GLOAD=1 ..." ["-DKB_EXP_..."] (e.g. nodma "KB_GEN_NO_DMA=1": the register-staged STREAM
# statements): a device library whose hand-scheduled statements
# come from a variant of tools/gen_lds_loop.py (timing experiments); only search_lds.hip is recompiled.
# -> tools/probe_bin/libkbmod_NAME.so (KBMOD_HIP_LIB=... python bench.py, tools/ab.sh)
set -e
cd "$(dirname "$0")/.."
NAME=$1; GENV=$2; DEFS=$3
mkdir -p tools/probe_bin /tmp/asm_$NAME
env $GENV KB_GEN_OUT=/tmp/asm_$NAME/search_lds_asm_variant.h python tools/gen_lds_loop.py
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude -Ikbmod_amd/csrc"
/opt/rocm/bin/hipcc $FL $DEFS -DKB_ASM_HEADER="\"/tmp/asm_$NAME/search_lds_asm_variant.h\"" -c kbmod_amd/csrc/search_lds.hip -o /tmp/asm_$NAME/search_lds.o 2>/tmp/asm_$NAME/err.log || { grep -m5 error /tmp/asm_$NAME/err.log; exit 1; }
objs=""; for s in search_lds_encoded search_direct search_kernels sigmag_kernels device_memory image_kernels result_kernels exchange_kernels stamp_kernels fits_kernels; do objs="$objs kbmod_amd/_obj/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/asm_$NAME/search_lds.o $objs -o tools/probe_bin/libkbmod_$NAME.so
echo built tools/probe_bin/libkbmod_$NAME.so
