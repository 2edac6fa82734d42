#!/bin/bash
# One line per configuration: tools/bench_table.sh "<bench args>" "<bench args>" ...   (env: BT_STEPS, BT_WARMUP, BT_EXTRA, BT_ENV="NAME=value ...")
# -> args | ms per step | kernel ms | evals/s | traffic GB | verify | kernel
for ARGS in "$@"; do
  env $BT_ENV python bench.py --steps ${BT_STEPS:-10} --warmup ${BT_WARMUP:-2} --no-cpu-baseline --no-masked $BT_EXTRA $ARGS 2>/tmp/bt.err | python -c "
import sys, json
seen = False
for line in sys.stdin:
    if line.startswith('{'):
        seen = True
        d = json.loads(line); r = d['roofline']
        v = d.get('verify'); ok = None if v is None else all(x for k, x in v.items() if k.endswith('_ok'))
        t = r.get('traffic')
        print('$BT_ENV $ARGS |', round(d['ms_per_step'], 3), '|', round(r['kernel_ms'], 3), '|', '%.3e' % d['value'], '|',
              None if t is None else round(t / 1e9, 2), '|', ok, '|', r['kernel'], '|', d.get('first_search', {}).get('first_search_ms'))
if not seen:
    print('$BT_ENV $ARGS | FAILED |', open('/tmp/bt.err').read()[-400:].replace(chr(10), ' / '))"
done
