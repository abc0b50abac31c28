python - <<'PY'
import os, subprocess, sys
code = """
import sys, time
sys.path.insert(0, '.')
import numpy as np
from demiurge_amd import capi, synth
ctx = capi.Context()
W, H = 4096, 2048
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
ctx.erosion_run(g, h, 50, 30, 0.5, 1.0, 1.0)
"""
env = dict(os.environ, DMG_FLOW_TIMING="1", DMG_TEST_HOOKS="1")
r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
out = (r.stdout + r.stderr).splitlines()
print("\n".join([l for l in out if "two phases" in l][-3:]))
PY
for nb in 128 256 512 2048 4096; do echo "NBLOCKS_A=$nb"; DMG_FLOW_NBLOCKS_A=$nb python tools/config2_time.py 2>&1 | tail -1; done
