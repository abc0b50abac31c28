set -o pipefail
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -q > gpurun_out/r02_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_gpu.log; tail -3 gpurun_out/r02_pytest_gpu.log
python bench.py > gpurun_out/r02_bench_default.json.log 2> gpurun_out/r02_bench_default.err; tail -c 600 gpurun_out/r02_bench_default.json.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_stats -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > gpurun_out/r02_stream_bench_profiled.json.log 2>&1
ls gpurun_out/r02_stats
DMG_ST_PROF=gpurun_out/r02_stream_unit_times.csv python bench.py --steps 1 --warmup 1 --sweeps 80 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1
DMG_TB_SHAPE=stream-p DMG_ST_PROF=gpurun_out/r02_stream_p_wave_times.csv python bench.py --steps 1 --warmup 1 --sweeps 80 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1
for t in survey ocean; do DMG_TB_SHAPE=stream-p python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --terrain $t 2>/dev/null | cut -c1-400 >> gpurun_out/r02_stream_p_bench.jsonl; done
for h in 8192 4096 2048 1024; do python bench.py --height $h --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | cut -c1-400 >> gpurun_out/r02_band_sizes.jsonl; done
python tools/config_lines.py > gpurun_out/r02_config_lines.jsonl 2> gpurun_out/r02_config_lines.err; tail -3 gpurun_out/r02_config_lines.jsonl | cut -c1-200
DMG_FLOW_TIMING=1 python tools/flow_timing.py > gpurun_out/r02_flow_timing.log 2>&1; tail -13 gpurun_out/r02_flow_timing.log
