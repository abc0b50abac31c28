"""The tools' kernels once each at 16384x8192, for a profiler (tools/kernel_table.py reads its kernel stats)."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from demiurge_amd import capi, synth
ctx = capi.Context(0)
W, H = 16384, 8192
g = capi.make_grid(W, H)
terrain = synth.terrain(W, H)
o = capi.Ocean(ctx, g)
o.set_terrain(terrain); o.set_divw(synth.divw(W, H))
prm_single = capi.default_ocean_params(); prm_single.jacobi_kernel = capi.JACOBI_SINGLE
o.step(); o.step(); ctx.sync()
o.close()
o1 = capi.Ocean(ctx, g, params=prm_single)
o1.set_terrain(terrain); o1.set_divw(synth.divw(W, H)); o1.zero_pressure(); o1.jacobi_sweeps(40); ctx.sync(); o1.close()
h = (terrain * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
# (a smooth field and a dozen steps: the reference's explicit scheme is unstable at this resolution, and a field of infinities takes
# the IEEE-division fallback in every cell -- tools/climate_time_probe.py)
lat = np.linspace(-1, 1, H, dtype=np.float32)[:, None]
T0 = (15 - 40 * np.abs(lat) + 3 * np.sin(np.linspace(0, 12 * np.pi, W, dtype=np.float32))[None, :]).astype(np.float32)
k = capi.Climate(ctx, g, T0, h); k.steps(0, 12); ctx.sync(); k.close()
ctx.erosion_run(g, h, 50, 1, 0.5, 1.0, 1.0)
ctx.blur(g, h, 3.0)
t = capi.Tectonics(ctx, g, h); t.step(2); ctx.sync(); t.close()
print("done")
