export TMPDIR=/tmp
cd /root/repo
for args in "32 4096 0" "32 4096 1" "32 4096 4" "32 4096 0 2" "64 512 0" "64 512 1"; do
rm -rf /tmp/kt3; rocprofv3 --kernel-trace --stats -d /tmp/kt3 -o r -- python tools/exp_build.py $args > /tmp/kt3.log 2>&1
echo "== $args"; python tools/rocprof_summary.py /tmp/kt3/r_results.db | grep "kb::"
done
