# config 5's coupled iteration at 16384x8192 under rocprofv3 --kernel-trace --stats: the kernels of an iteration, by time
# usage (GPU box): bash tools/world_kernel_table.sh [library] > gpurun_out/rNN_world_kernels.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
[ -n "$1" ] && export DMG_LIB=$1
rm -rf gpurun_out/wk
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wk -o x -- python3 tools/world_kernels.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/wk/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernels busy %.1f ms per iteration (4 iterations profiled)" % (tot / 4e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print("  %-46s calls %5d  avg %8.1f us  per iteration %6.2f ms" % (r["Name"][:46], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 4e6))
PY
rm -rf gpurun_out/wk
