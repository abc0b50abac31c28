"""PCIe-inclusive rate of one reference solve at 16384x8192: terrain and right-hand side uploaded from host arrays,
5000 sweeps, pressure downloaded (bench.py's `value` starts with the inputs resident; this is the other number)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from demiurge_amd import capi, synth
W,H=16384,8192
ctx=capi.Context(0); o=capi.Ocean(ctx,capi.make_grid(W,H))
img=synth.terrain(W,H); b=synth.divw(W,H)
o.set_terrain(img); o.set_divw(b); o.jacobi(5000); ctx.sync()
for rep in range(2):
    t0=time.perf_counter(); o.set_terrain(img); t1=time.perf_counter(); o.set_divw(b); t2=time.perf_counter()
    o.jacobi(5000); ctx.sync(); t3=time.perf_counter(); p=o.get(want=("pressure",))["pressure"]; t4=time.perf_counter()
    print("set_terrain %.1f ms (incl. codes/classes), set_divw %.1f ms, solve %.1f ms, get pressure %.1f ms -> host-to-host %.1f ms = %.0f Mcell-steps/s" % ((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t4-t3)*1e3,(t4-t0)*1e3, W*H*5000/(t4-t0)/1e6))
