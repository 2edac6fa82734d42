#!/bin/bash
# tools/explore.sh: one bench.py line per argument set (stdin: "ENV=.. ENV=.. | bench args" per line), printed as
# ms/step, kernel ms, kernel instance
while IFS='|' read -r envs args; do
  [ -z "$envs$args" ] && continue
  out=$(env $envs timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $args 2>/dev/null | grep '"metric"' | tail -1)
  echo "$out" | python -c "
import sys, json
l = sys.stdin.read().strip()
tag = '''$envs | $args'''
if not l:
    print(tag, '-> no line'); sys.exit()
d = json.loads(l); r = d['roofline']
print(tag, '->', round(d['ms_per_step'], 3), 'ms/step, kernel', round(r['kernel_ms'], 3), 'ms', r['kernel'], '%.3g evals/s' % d['value'])
"
done
