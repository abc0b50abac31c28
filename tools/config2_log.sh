# Config 2 (4096x2048 stream-power erosion, 1000 iterations) on the GPU box: the phases of the flow solve, the wall time of the three forms
# of the upstream sums, the kernels of an iteration and the flow solve's own lines for the last iterations -> profiles/rNN_config2_phases.log
# usage (GPU box): bash tools/config2_log.sh > gpurun_out/r06_config2_phases.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export DMG_TEST_HOOKS=1
echo "== phases (DMG_FLOW_TIMING=1: every phase ends in a synchronisation, so their sum is more than the untimed run below)"
python3 tools/config2_phases.py
echo "== wall time, default (two phases, phase B along the reaches)"
python3 tools/config2_time.py
echo "== phase B as a dataflow in depth order (DMG_FLOW_WALK=0)"
DMG_FLOW_WALK=0 python3 tools/config2_time.py | tail -1
echo "== one pass over all cells in depth order (DMG_FLOW_TWO_PHASE=0: the form through round 5)"
DMG_FLOW_TWO_PHASE=0 python3 tools/config2_time.py | tail -1
echo "== kernels of an iteration (rocprofv3 --kernel-trace --stats over 300 iterations)"
rm -rf gpurun_out/c2k
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c2k -o x -- python3 tools/config2_kernels.py 300 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/c2k/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernels busy %.2f ms per iteration" % (tot / 300e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:26]:
    print("  %-60s calls/it %6.1f  avg %8.1f us  per iteration %6.3f ms" % (r["Name"][:60], int(r["Calls"]) / 300, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 300e6))
PY
rm -rf gpurun_out/c2k
echo "== the flow solve's lines, iterations 298-300"
DMG_FLOW_TIMING=1 python3 tools/config2_kernels.py 300 2>&1 | grep "two phases\|lakes,\|growth:" | tail -9
