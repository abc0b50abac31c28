cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=r06
python bench.py > gpurun_out/${R}_bench_default.json.log 2> gpurun_out/${R}_bench_default.err; tail -c 400 gpurun_out/${R}_bench_default.json.log
rm -rf gpurun_out/${R}_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_stats -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > gpurun_out/${R}_stream_bench_profiled.json.log 2>&1
cp gpurun_out/${R}_stats/x_kernel_stats.csv gpurun_out/${R}_stream_kernel_stats.csv 2>/dev/null; rm -rf gpurun_out/${R}_stats
head -3 gpurun_out/${R}_stream_kernel_stats.csv
