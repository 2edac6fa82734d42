export TMPDIR=/tmp
run() { # label, env, args
  lbl=$1; shift; envs=$1; shift
  env $envs timeout 600 python bench.py --no-cpu-baseline --no-live-traffic --no-masked "$@" 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$lbl', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['roofline']['kernel'][-30:], d.get('verify'))
"
}
for rep in 1 2; do
 run "cfg2 tree   " "KBMOD_CHUNK=32" --steps 20 --warmup 3 --verify
 run "cfg2 halves " "KBMOD_CHUNK=32 KBMOD_HIP_LIB=tools/probe_bin/libkbmod_halves.so" --steps 20 --warmup 3
 run "cfg2 c16    " "A=1" --steps 20 --warmup 3
 run "cfg4 tree   " "KBMOD_CHUNK=32" --steps 10 --warmup 2 --frames 128 --size 4096 --vel-steps 32 --ang-steps 2
 run "cfg4 halves " "KBMOD_CHUNK=32 KBMOD_HIP_LIB=tools/probe_bin/libkbmod_halves.so" --steps 10 --warmup 2 --frames 128 --size 4096 --vel-steps 32 --ang-steps 2
 run "cfg4 c16    " "A=1" --steps 10 --warmup 2 --frames 128 --size 4096 --vel-steps 32 --ang-steps 2
done
run "cfg5 tree   " "A=1" --steps 3 --warmup 1 --frames 512 --size 2048 --vel-steps 64 --ang-steps 64 --num-bytes 2 --verify
run "cfg5 halves " "KBMOD_HIP_LIB=tools/probe_bin/libkbmod_halves.so" --steps 3 --warmup 1 --frames 512 --size 2048 --vel-steps 64 --ang-steps 64 --num-bytes 2
