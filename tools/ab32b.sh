export TMPDIR=/tmp
run() { lbl=$1; shift; envs=$1; shift
  env $envs timeout 900 python bench.py --no-cpu-baseline --no-live-traffic --no-masked "$@" 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$lbl', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['roofline']['kernel'][-30:], d.get('verify'))
"
}
A="--steps 5 --warmup 1 --frames 128 --size 2048 --vel-steps 32 --ang-steps 8"
C5="--steps 3 --warmup 1 --frames 512 --size 2048 --vel-steps 64 --ang-steps 64 --num-bytes 2"
for rep in 1 2; do
run "mid seq    " "KBMOD_CHUNK=32" $A
run "mid halves " "KBMOD_CHUNK=32 KBMOD_HIP_LIB=tools/probe_bin/libkbmod_halves.so" $A
run "cfg5 seq   " "A=1" $C5
run "cfg5 halves" "KBMOD_HIP_LIB=tools/probe_bin/libkbmod_halves.so" $C5
done
run "cfg2 seq   " "KBMOD_CHUNK=32" --steps 20 --warmup 3 --verify
