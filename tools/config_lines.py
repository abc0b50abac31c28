#!/usr/bin/env python3
"""One line per BASELINE.json config on one MI355X (what a single GPU can show of each): JSON objects on stdout."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: torch first)

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from demiurge_amd import capi, synth  # noqa: E402


def emit(**kw):
    print(json.dumps(kw), flush=True)


ctx = capi.Context(0)
emit(device=ctx.name())

# config 1: 1024x512 Gaussian blur (whole filter, host buffers in and out: PCIe-inclusive)
W, H = 1024, 512
g = capi.make_grid(W, H)
f = synth.terrain(W, H)
for r in (1.0, 10.0):
    ctx.blur(g, f, r)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.blur(g, f, r)
    emit(config=1, line="blur 1024x512 radius %g, host to host" % r, ms_per_filter=(time.perf_counter() - t0) / 5 * 1e3)

# config 2: 4096x2048 stream-power erosion (1000 iterations in the config; per-round time from a short run)
W, H = 4096, 2048
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
ctx.erosion_run(g, h, 50, 1, 0.5, 1.0, 1.0)
t0 = time.perf_counter(); ctx.erosion_run(g, h, 50, 2, 0.5, 1.0, 1.0); ta = time.perf_counter() - t0
t0 = time.perf_counter(); ctx.erosion_run(g, h, 50, 12, 0.5, 1.0, 1.0); tb = time.perf_counter() - t0
per = (tb - ta) / 10
emit(config=2, line="stream-power erosion round 4096x2048 (flow accumulation + erosion/uplift pass)", ms_per_round=per * 1e3,
     projected_s_for_1000_iters=per * 1000)
t0 = time.perf_counter(); ctx.erosion_run(g, h, 50, 1000, 0.5, 1.0, 1.0); tc = time.perf_counter() - t0
emit(config=2, line="stream-power erosion 4096x2048, the config's 1000 iterations, host to host", seconds=tc, ms_per_round=tc)

# configs 3 and 4: ocean outer step (advect + 50 diffusion + divergence + 5000 Jacobi + subDiv)
for cfg, (W, H) in ((3, (8192, 4096)), (4, (16384, 8192))):
    o = capi.Ocean(ctx, capi.make_grid(W, H))
    img = synth.terrain(W, H)
    o.set_terrain(img)
    v = synth.velocity(W, H); v[img > 0] = 0
    o.set_velocity(v); del v, img
    o.step(1); ctx.sync()
    t0 = time.perf_counter(); o.step(2); ctx.sync(); dt = (time.perf_counter() - t0) / 2
    emit(config=cfg, line="ocean outer step %dx%d on ONE GPU (the config's 8-GPU band split runs at round end)" % (W, H) if cfg == 4
         else "ocean outer step %dx%d" % (W, H), ms_per_outer_step=dt * 1e3, jacobi_mcell_steps_per_s_equiv=W * H * 5000 / dt / 1e6)
    o.close()

# config 5 pieces at 16384x8192: tectonics step, erosion round, climate step
W, H = 16384, 8192
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
t = capi.Tectonics(ctx, g, h)
t.step(3); ctx.sync()
t0 = time.perf_counter(); t.step(10); ctx.sync()
emit(config=5, line="tectonics step 16384x8192", ms_per_step=(time.perf_counter() - t0) * 100)
t.close()
ctx.erosion_run(g, h, 50, 1, 0.5, 1.0, 1.0)                      # (the context's scratch pool fills on the first call: 20 grid-sized buffers)
t0 = time.perf_counter(); ctx.erosion_run(g, h, 50, 2, 0.5, 1.0, 1.0); dt = time.perf_counter() - t0
emit(config=5, line="erosion round 16384x8192 (two rounds per call, incl. PCIe of the heights: 0.5 GB each way per call)", ms_per_round=dt / 2 * 1e3)
t0 = time.perf_counter(); ctx.erosion_run(g, h, 50, 6, 0.5, 1.0, 1.0); dt6 = time.perf_counter() - t0
emit(config=5, line="erosion round 16384x8192, the rounds alone (six rounds per call against two)", ms_per_round=(dt6 - dt) / 4 * 1e3)
T0 = np.full((H, W), 50, np.float32)
k = capi.Climate(ctx, g, T0, h)                                  # field resident on the device: a step costs its kernels
k.steps(0, 10); ctx.sync()
t0 = time.perf_counter(); k.steps(10, 100); ctx.sync(); dt = time.perf_counter() - t0
k.close()
emit(config=5, line="climate step 16384x8192 (field resident on the device; the reference's values: the explicit scheme leaves fp32 after ~40 steps at this resolution and the cells go through the IEEE fall-back)", ms_per_step=dt / 100 * 1e3)
lat = np.linspace(-1, 1, H, dtype=np.float32)[:, None]
Ts = (15 - 40 * np.abs(lat) + 3 * np.sin(np.linspace(0, 12 * np.pi, W, dtype=np.float32))[None, :]).astype(np.float32)
k = capi.Climate(ctx, g, Ts, h, params=capi.climate_params(diffusivity=0.55e4))     # values the scheme is stable with at this resolution
k.steps(0, 10); ctx.sync()
t0 = time.perf_counter(); k.steps(10, 500); ctx.sync(); dt = time.perf_counter() - t0
k.close()
emit(config=5, line="climate step 16384x8192, stable values (diffusivity 0.55e4): a year of 15000 steps = %.2f s" % (dt / 500 * 15000), ms_per_step=dt / 500 * 1e3)

# config 5 through its entry point: one coupled iteration (tectonics step -> erosion round -> 10 climate steps)
terr = (synth.terrain(W, H) * 100).astype(np.float32)
w = capi.World(ctx, g, terr, T0)
w.step()
t0 = time.perf_counter(); w.step(); dt = time.perf_counter() - t0
emit(config=5, line="dmg_world_step 16384x8192: 1 tectonics step + 1 erosion round + 10 climate steps, one GPU", ms_per_iteration=dt * 1e3)
w.close()
