#!/bin/bash
# the climate step under the profiler: one --stats pass, two SQ --pmc passes, FETCH / WRITE passes (separate passes: --pmc never
# next to --stats; the program directly after --), summarised by tools/kernel_counters.py.  usage: bash tools/climate_counters.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
TAG=${1:-r05}
D=$R/gpurun_out/${TAG}_cc
rm -rf $D; mkdir -p $D
cd $R
P=tools/climate_time_probe.py
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o x -- python3 $P > $D/run.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $D/sq1 -o x -- python3 $P >> $D/run.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $D/sq2 -o x -- python3 $P >> $D/run.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -o x -- python3 $P >> $D/run.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -o x -- python3 $P >> $D/run.log 2>&1
python tools/kernel_counters.py $D > $R/gpurun_out/${TAG}_climate_counters.txt 2>&1
grep -h "k_temperature\|k_land" $D/stats/*kernel_stats.csv >> $R/gpurun_out/${TAG}_climate_counters.txt
cat $R/gpurun_out/${TAG}_climate_counters.txt
rm -rf $D/sq1 $D/sq2 $D/fetch $D/write
