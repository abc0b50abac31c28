#!/bin/bash
# same-box A/B of two builds on the bands of the split grid: base (DMG_LIB=demiurge_amd/libdemiurge_hip_base.so) against the tree's
for rep in 1 2; do
  echo "== base (rep $rep)"; DMG_LIB=demiurge_amd/libdemiurge_hip_base.so python tools/band_probe.py 8 4 --sweeps 800 2>&1 | grep "band [0-9] \|slowest" | cut -c1-110
  echo "== new (rep $rep)";  python tools/band_probe.py 8 4 --sweeps 800 2>&1 | grep "band [0-9] \|slowest" | cut -c1-110
done
