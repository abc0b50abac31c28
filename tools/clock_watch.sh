#!/bin/bash
# What does the GPU clock at while the headline loop runs?  Starts bench.py (60 steps: ~10 s of back-to-back launches) and samples
# rocm-smi beside it.  usage (GPU box): bash tools/clock_watch.sh > gpurun_out/clock_watch.log
cd ${GRAFT_REPO_ROOT:-/root/repo}
rocm-smi --showclocks 2>&1 | head -20
echo "--- idle above, loaded below ---"
python bench.py --steps 60 --warmup 2 > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
BP=$!
sleep 25
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|fclk\|power" | tr '\n' ';'; echo
  cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | tr '\n' ' '; echo
  sleep 1
done
wait $BP
tail -c 600 gpurun_out/clock_bench.json
