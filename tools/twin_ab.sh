#!/bin/bash
# Same-box A/B of the twin units (round 4): writes gpurun_out/r04_twin_units.txt.  usage: bash tools/twin_ab.sh
cd "$(dirname "$0")/.."
L=gpurun_out/r04_twin_units.txt
mkdir -p gpurun_out
: > $L
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %.0f Mcell-steps/s, %.2f us per launch, %s' % (d['value'], d['roofline']['launch_us'], d['roofline'].get('kernel')))"; }
echo "# bench.py --height H (16384 wide, one GPU): plain units (DMG_ST_TWIN=0) against the default, two repetitions" >> $L
for rep in 1 2; do
for h in 512 1024 2048; do
  for terr in survey ocean; do
    for tw in 0 1; do
      echo "height $h $terr DMG_ST_TWIN=$tw" >> $L
      DMG_ST_TWIN=$tw timeout -k 10 300 python bench.py --steps 2 --warmup 1 --height $h --terrain $terr --no-extra-legs --no-cpu-baseline 2>/dev/null | line >> $L 2>&1
    done
  done
done
done
echo "# the middle band of an N-way split with its ghost rows (tools/band_probe.py N 32), two repetitions" >> $L
for rep in 1 2; do
for n in 8 4; do
  for tw in 0 1; do
    echo "bands $n DMG_ST_TWIN=$tw" >> $L; DMG_ST_TWIN=$tw python tools/band_probe.py $n 32 2>&1 | grep "whole\|split" >> $L
  done
done
done
for rep in 1 2; do
  for tw in 0 1; do
    echo "bands 2 DMG_ST_TWIN=$tw" >> $L; DMG_ST_TWIN=$tw python tools/band_probe.py 2 32 2>&1 | grep "whole\|split" >> $L
  done
done
echo "# taller launches: 16384 x 4096 (twin units by default), 16384 x 8192 (plain units by default; forced: one round of pairs of up to 232 rows, and the best of 1 / 1.5 / 2 / 3 rounds' worth)" >> $L
for rep in 1 2; do
  for terr in survey ocean; do
    for tw in 0 1; do
      echo "height 4096 $terr DMG_ST_TWIN=$tw" >> $L
      DMG_ST_TWIN=$tw timeout -k 10 300 python bench.py --steps 2 --warmup 1 --height 4096 --terrain $terr --no-extra-legs --no-cpu-baseline 2>/dev/null | line >> $L 2>&1
    done
    for cfg in "DMG_ST_TWIN=1" "DMG_ST_TWIN=2 DMG_ST_TWIN_ROWS=9000" "DMG_ST_TWIN=2 DMG_ST_TWIN_ROWS=9000 DMG_ST_TWIN_ROUNDS=3"; do
      echo "height 8192 $terr $cfg" >> $L
      env $cfg timeout -k 10 400 python bench.py --steps 2 --warmup 1 --terrain $terr --no-extra-legs --no-cpu-baseline 2>/dev/null | line >> $L 2>&1
    done
  done
done
echo "# what the cut's parts are worth on the 8-way band: without the packing under a common limit (DMG_ST_PACK=0: units dealt out by row weights), without the placement (DMG_ST_PLACE=0: longest first)" >> $L
for rep in 1 2; do
for env in "DMG_ST_PACK=1 DMG_ST_PLACE=1" "DMG_ST_PACK=0 DMG_ST_PLACE=1" "DMG_ST_PACK=1 DMG_ST_PLACE=0" "DMG_ST_PACK=0 DMG_ST_PLACE=0"; do
  echo "bands 8 $env" >> $L; env $env python tools/band_probe.py 8 32 2>&1 | grep "whole" >> $L
done
done
echo "# the cut of a 16384 x 1024 launch (DMG_ST_DEBUG)" >> $L
DMG_ST_DEBUG=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --height 1024 --no-extra-legs --no-cpu-baseline 2>&1 >/dev/null | grep "pairs of twin\|pairs placed\|rows \[0,1024) part 0" | cut -c1-260 >> $L
echo "# one warm launch of it, unit by unit and SIMD by SIMD (DMG_ST_PROF, the 31st launch; tools/unit_simds.py): twin units, then plain units" >> $L
for tw in 1 0; do
  DMG_ST_TWIN=$tw DMG_ST_PROF_SKIP=30 DMG_ST_PROF=$PWD/gpurun_out/r04_twin_prof_$tw.csv timeout -k 10 300 python bench.py --steps 1 --warmup 1 --height 1024 --no-extra-legs --no-cpu-baseline > /dev/null 2>&1
  echo "DMG_ST_TWIN=$tw" >> $L
  python tools/unit_simds.py gpurun_out/r04_twin_prof_$tw.csv >> $L 2>&1
done
