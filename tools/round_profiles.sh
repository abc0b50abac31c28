# the evidence set of a round: run on the GPU box (gpurun -- 'bash tools/round_profiles.sh'), then copy what is to be judged
# from gpurun_out/ into profiles/.  R = round tag.
set -o pipefail
R=${R:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# PART=tests: the GPU test suite alone (a gpurun call is at most 1200 s); PART=twin: the twin units' A/B; PART=profiles (default): everything else
if [ "${PART:-profiles}" = tests ]; then
timeout -k 10 1100 python -u -m pytest tests -m gpu -q > gpurun_out/${R}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${R}_pytest_gpu.log; tail -3 gpurun_out/${R}_pytest_gpu.log
grep -q "GPU core dump\|Memory access fault" gpurun_out/${R}_pytest_gpu.log && exit 9
exit 0
fi
if [ "${PART:-profiles}" = twin ]; then
# the twin units' A/B (DMG_ST_TWIN=0/1 and the cut's parts), and the band kernel under the profiler
bash tools/twin_ab.sh > /dev/null 2>&1; cp gpurun_out/r04_twin_units.txt gpurun_out/${R}_twin_units.txt 2>/dev/null; head -30 gpurun_out/${R}_twin_units.txt
rm -rf gpurun_out/${R}_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_stats -o x -- python3 bench.py --steps 2 --warmup 1 --height 1024 --no-cpu-baseline --no-extra-legs > gpurun_out/${R}_twin_bench_profiled.json.log 2>&1
cp gpurun_out/${R}_stats/x_kernel_stats.csv gpurun_out/${R}_twin_kernel_stats.csv 2>/dev/null; rm -rf gpurun_out/${R}_stats
exit 0
fi
python bench.py > gpurun_out/${R}_bench_default.json.log 2> gpurun_out/${R}_bench_default.err; tail -c 600 gpurun_out/${R}_bench_default.json.log
rm -rf gpurun_out/${R}_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_stats -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > gpurun_out/${R}_stream_bench_profiled.json.log 2>&1
cp gpurun_out/${R}_stats/x_kernel_stats.csv gpurun_out/${R}_stream_kernel_stats.csv 2>/dev/null; rm -rf gpurun_out/${R}_stats
# SQ counters of the default kernel, one --pmc pass per group (no --stats / trace domains next to --pmc)
rm -rf gpurun_out/${R}_sq
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/${R}_sq/$(echo $grp | md5sum | cut -c1-6) -o x -- python3 bench.py --steps 1 --warmup 1 --sweeps 160 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1
done
python tools/pmc_sq.py gpurun_out/${R}_sq k_jacobi_stream > gpurun_out/${R}_stream_sq_counters.txt 2>&1; rm -rf gpurun_out/${R}_sq; cat gpurun_out/${R}_stream_sq_counters.txt
rm -f gpurun_out/${R}_band_sizes.jsonl
for h in 8192 4096 2048 1024; do python bench.py --height $h --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | cut -c1-420 >> gpurun_out/${R}_band_sizes.jsonl; done
python tools/config_lines.py > gpurun_out/${R}_config_lines.jsonl 2> gpurun_out/${R}_config_lines.err; tail -4 gpurun_out/${R}_config_lines.jsonl | cut -c1-200
DMG_FLOW_TIMING=1 python tools/flow_timing.py > gpurun_out/${R}_flow_timing.log 2>&1; tail -13 gpurun_out/${R}_flow_timing.log
python tools/config2_phases.py 1000 > gpurun_out/${R}_config2_phases.log 2>&1; cat gpurun_out/${R}_config2_phases.log
python tools/band_probe.py 8 4 2 --split 2>&1 | grep bands > gpurun_out/${R}_band_probe.log; cat gpurun_out/${R}_band_probe.log
timeout -k 10 300 python tools/bands_one_gpu.py 1600 loopback > gpurun_out/${R}_bands_one_gpu_loopback.log 2>&1; grep bands gpurun_out/${R}_bands_one_gpu_loopback.log
timeout -k 10 300 python tools/bands_one_gpu.py 1600 store > gpurun_out/${R}_bands_one_gpu_store.log 2>&1; grep bands gpurun_out/${R}_bands_one_gpu_store.log
timeout -k 10 300 python tools/bands_one_gpu.py 1600 peer > gpurun_out/${R}_bands_one_gpu_peer.log 2>&1; grep bands gpurun_out/${R}_bands_one_gpu_peer.log
rm -f gpurun_out/${R}_overlap_probe.log
for o in plain overlap; do
  python tools/overlap_probe.py $o 1024 2 640 32 2>&1 | grep order >> gpurun_out/${R}_overlap_probe.log
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_ovl_$o -o x -- python3 tools/overlap_probe.py $o 1024 2 640 32 > /dev/null 2>&1
  python tools/overlap_trace.py gpurun_out/${R}_ovl_$o $o >> gpurun_out/${R}_overlap_probe.log; rm -rf gpurun_out/${R}_ovl_$o
done
cat gpurun_out/${R}_overlap_probe.log
