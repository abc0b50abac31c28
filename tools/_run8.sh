set -o pipefail
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tests/stream_worker.py random pepper bands > gpurun_out/sw.log 2>&1; echo "worker rc=$?"; tail -3 gpurun_out/sw.log
grep -q "GPU core dump\|Memory access fault" gpurun_out/sw.log && exit 9
for rep in 1 2 3; do for v in base exp; do
echo "$v survey: $(python tmp_ab/$v/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | cut -c75-95) ocean: $(python tmp_ab/$v/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --terrain ocean 2>/dev/null | cut -c75-95) 1024: $(python tmp_ab/$v/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --height 1024 2>/dev/null | cut -c75-95)"
done; done
