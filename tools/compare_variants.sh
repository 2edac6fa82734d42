#!/bin/bash
# tools/compare_variants.sh lib1 lib2 ...: bench.py kernel time for each variant library of tools/build_variants.sh
# ("main" = the shipped library); extra bench arguments in $EXTRA, steps in $STEPS.
for l in "$@"; do for i in 1 2; do
  if [ "$l" = main ]; then unset KBMOD_HIP_LIB; else export KBMOD_HIP_LIB=tools/probe_bin/libkbmod_$l.so; fi
  timeout 900 python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-3} --no-cpu-baseline --verify $EXTRA 2>&1 | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); v=d.get('verify') or {}; print('$l', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), all(x for k,x in v.items() if k.endswith('_ok')))
"
done; done
