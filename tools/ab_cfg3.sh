export TMPDIR=/tmp
for rep in 1 2; do for l in main base; do
  if [ "$l" = main ]; then unset KBMOD_HIP_LIB; else export KBMOD_HIP_LIB=tools/probe_bin/libkbmod_$l.so; fi
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --no-masked --num-bytes 1 --sigmag "$@" 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$l', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['roofline']['kernel'][-30:], d.get('verify'))
"
done; done
