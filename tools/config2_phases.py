"""Config 2 (4096x2048 stream-power erosion, 1000 iterations): where the time goes, by phase of the flow solve
(DMG_FLOW_TIMING lines summed over the run; GPU box).  usage: python tools/config2_phases.py [iterations]"""
import os, re, subprocess, sys, collections
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
code = """
import sys, time
sys.path.insert(0, '.')
import numpy as np
from demiurge_amd import capi, synth
ctx = capi.Context()
W, H = 4096, 2048
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
ctx.erosion_run(g, h, 50, 1, 0.5, 1.0, 1.0)
t0 = time.perf_counter(); ctx.erosion_run(g, h, 50, %d, 0.5, 1.0, 1.0); print('TOTAL %%.3f' %% (time.perf_counter() - t0))
""" % n
env = dict(os.environ, DMG_FLOW_TIMING="1", DMG_TEST_HOOKS="1")
r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
out = r.stdout + r.stderr
tot = collections.OrderedDict()
lakes = []
for line in out.splitlines():
    m = re.match(r"\[flow\] (.*?)\s+([0-9.]+) ms$", line)
    if m:
        tot[m.group(1)] = tot.get(m.group(1), 0.0) + float(m.group(2))
    m = re.match(r"\[flow\] (\d+) lakes", line)
    if m:
        lakes.append(int(m.group(1)))
    if line.startswith("TOTAL"):
        print(line)
for k, v in tot.items():
    print("%-40s %9.1f ms" % (k, v))
print("sum of phases %.1f ms; lakes per iteration: first %s ... last %s; iterations with >= 100000 lakes: %d" %
      (sum(tot.values()), lakes[1:4], lakes[-3:], sum(1 for x in lakes[1:] if x >= 100000)))
