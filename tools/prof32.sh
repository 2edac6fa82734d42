export TMPDIR=/tmp KBMOD_EXP_PROFILE=1
run() { lbl=$1; shift; envs=$1; shift
  env $envs timeout 600 python bench.py --no-cpu-baseline --no-live-traffic --no-masked "$@" 2>&1 | grep -E "phase ticks|^\{" | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$lbl', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['roofline']['kernel'][-30:])
    else: print('$lbl', line.strip())
"
}
A="--steps 5 --warmup 1 --frames 128 --size 2048 --vel-steps 32 --ang-steps 8"
run "mid tree  " "KBMOD_CHUNK=32 KBMOD_HIP_LIB=tools/probe_bin/libkbmod_proftree.so" $A
run "mid halves" "KBMOD_CHUNK=32 KBMOD_HIP_LIB=tools/probe_bin/libkbmod_profhalves.so" $A
run "mid c16   " "KBMOD_HIP_LIB=tools/probe_bin/libkbmod_proftree.so" $A
B="--steps 10 --warmup 2"
run "cfg2 tree  " "KBMOD_CHUNK=32 KBMOD_HIP_LIB=tools/probe_bin/libkbmod_proftree.so" $B
run "cfg2 halves" "KBMOD_CHUNK=32 KBMOD_HIP_LIB=tools/probe_bin/libkbmod_profhalves.so" $B
run "cfg2 c16   " "KBMOD_HIP_LIB=tools/probe_bin/libkbmod_proftree.so" $B
