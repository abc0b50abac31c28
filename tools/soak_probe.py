import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from demiurge_amd import capi, synth
ctx = capi.Context(0)
W, H = 2048, 1024
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter(); out = ctx.erosion_run(g, h, 50, 300, 0.5, 1.0, 1.0); dt = time.perf_counter() - t0
free1 = torch.cuda.mem_get_info()[0]
print("erosion 300 rounds %dx%d: %.2f s (%.1f ms/round), finite %s, pool holds %.0f MB" % (W, H, dt, dt / 300 * 1e3, np.isfinite(out).all(), (free0 - free1) / 1e6))
ctx.trim(); print("after trim: %.0f MB" % ((free0 - torch.cuda.mem_get_info()[0]) / 1e6))
t = capi.Tectonics(ctx, capi.make_grid(4096, 2048), (synth.terrain(4096, 2048) * 3).astype(np.float32))
t0 = time.perf_counter(); t.step(70); ter = t.terrain(); dt = time.perf_counter() - t0
print("tectonics 70 steps 4096x2048: %.1f ms, plates %s, terrain values %s" % (dt * 1e3, [p.shape for p in t.plates()], np.unique(ter)))
