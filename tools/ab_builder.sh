export TMPDIR=/tmp
for rep in 1 2; do for m in mirror plain; do
  if [ $m = plain ]; then export KBMOD_BUILD_NO_MIRROR=1; else unset KBMOD_BUILD_NO_MIRROR; fi
  python tools/exp_build.py 64 512 | sed "s/^/$m /"
  python tools/exp_build.py 128 4096 | sed "s/^/$m /"
done; done
unset KBMOD_BUILD_NO_MIRROR
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['psi_phi_build']); print(d['masked']['ms_per_step'], d['first_search'])
"
