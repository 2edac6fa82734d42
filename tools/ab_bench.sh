#!/bin/bash
# A/B of kernel variants on one box: tools/ab_bench.sh "<flags list>" [extra bench args]
FLAGS="$1"; shift
for f in $FLAGS; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --flags $f "$@" 2>/dev/null | tail -1 > /tmp/ab.json
  python - "$f" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print("flags", sys.argv[1], "evals/s %.4g" % d["value"], "kernel_ms %.3f" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"])
PY
done
