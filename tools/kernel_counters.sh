#!/bin/bash
# one --stats pass, two SQ --pmc passes and the FETCH / WRITE passes over every tool's kernel at 16384x8192
# (separate passes: --pmc never next to --stats; the program directly after --)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
TAG=${TAG:-r06}
D=$R/gpurun_out/${TAG}_kc
rm -rf $D; mkdir -p $D
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o x -- python3 tools/kernel_table_run.py > $D/run.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $D/sq1 -o x -- python3 tools/kernel_table_run.py >> $D/run.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $D/sq2 -o x -- python3 tools/kernel_table_run.py >> $D/run.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -o x -- python3 tools/kernel_table_run.py >> $D/run.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -o x -- python3 tools/kernel_table_run.py >> $D/run.log 2>&1
python tools/kernel_counters.py $D > $R/gpurun_out/${TAG}_kernel_counters.txt 2>&1
python tools/kernel_table.py $D/stats > $R/gpurun_out/${TAG}_kernel_table.txt 2>&1
cat $R/gpurun_out/${TAG}_kernel_counters.txt
# the traces are large: keep the tables only
# (the traces are ~60 MB: only the tables are merged back)
rm -rf $D/sq1 $D/sq2 $D/fetch $D/write
