"""Parity of every pass against the committed llvmpipe goldens (the reference's own GLSL) and against the oracle:
max|a-ref|/max|ref| and the fraction of bit-equal values.  GPU box; TEST tooling (uses oracle/)."""
import sys
from pathlib import Path

import numpy as np
import torch  # noqa: F401

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
import oracle as orc  # noqa: E402
from conftest import load_golden, relerr  # noqa: E402
from demiurge_amd import capi  # noqa: E402

ctx = capi.Context(0)


def rep(tag, a, gold, o=None):
    s = "%-34s vs golden %.2e (%.4f equal)" % (tag, relerr(a, gold), (a == gold).mean())
    if o is not None:
        s += "   vs oracle %.2e (%.4f equal)" % (relerr(a, o), (a == o).mean())
    print(s)


G = load_golden("jacobi_256x128_full")
grid, g = capi.make_grid(256, 128), orc.grid(256, 128)
oc = capi.Ocean(ctx, grid); oc.set_terrain(G["img"]); oc.set_divw(G["divw"])
for it, key in ((1, "p1"), (100, "p100"), (5000, "p5000")):
    oc.jacobi(it)
    rep("jacobi 256x128 x%d" % it, oc.get(("pressure",))["pressure"], G[key], orc.ocean_jacobi(g, G["img"], G["divw"], it, rows=True))
oc.close()
for n in ("full", "region"):
    G = load_golden("ocean_passes_128x64_" + n)
    c = tuple(G["coords"])
    grid, g = capi.make_grid(128, 64, c, float(G["circ"])), orc.grid(128, 64, c, float(G["circ"]))
    def fresh():
        o = capi.Ocean(ctx, grid); o.set_terrain(G["img"]); o.set_velocity(G["v"]); return o
    o = fresh(); o.divergence(); rep(n + " divergence", o.get(("divw",))["divw"], G["divergence"], orc.ocean_divergence(g, G["img"], G["v"], 100.0)); o.close()
    o = fresh(); o.diffusion(50); rep(n + " diffusion x50", o.get(("v",))["v"], G["diffusion50"], orc.ocean_diffusion(g, G["img"], G["v"], 50)); o.close()
    o = fresh(); o.advect(); rep(n + " advect", o.get(("v",))["v"], G["advect"], orc.ocean_advect(g, G["img"], G["v"])); o.close()
    o = fresh(); o.set_pressure(G["pressure200"]); o.subdiv(); rep(n + " subDiv", o.get(("v",))["v"], G["subdiv"], orc.ocean_subdiv(g, G["img"], G["v"], G["pressure200"], 100.0)); o.close()
G = load_golden("ocean_step_128x64_full")
grid = capi.make_grid(128, 64)
prm = capi.default_ocean_params(); prm.jacobi_iters, prm.diffusion_iters = int(G["jacobi_iters"]), int(G["diffusion_iters"])
o = capi.Ocean(ctx, grid, params=prm); o.set_terrain(G["img"]); o.step(int(G["n_outer"]))
r = o.get(("v", "pressure")); rep("outer step x2: v", r["v"], G["v"]); rep("outer step x2: pressure", r["pressure"], G["pressure"]); o.close()
for nm in ("blur_128x64_r1", "blur_128x64_r10", "blur_128x64_region_r3"):
    G = load_golden(nm); c = tuple(G["coords"])
    rep(nm, ctx.blur(capi.make_grid(128, 64, c, float(G["circ"])), G["field"], float(G["radius"])), G["out"],
        orc.blur(orc.grid(128, 64, c, float(G["circ"])), G["field"], float(G["radius"])))
G = load_golden("temperature_128x64_full"); grid, g = capi.make_grid(128, 64), orc.grid(128, 64)
rep("temperature x20", ctx.temperature_steps(grid, G["T0"], G["terrain"], 0, 20), G["T20"], orc.temperature(g, G["T0"], G["terrain"], 0, 20))
G = load_golden("temperature_128x64_rough")
rep("temperature, rough field x5", ctx.temperature_steps(grid, G["T0"], G["terrain"], 7000, 5), G["T5_from7000"], orc.temperature(g, G["T0"], G["terrain"], 7000, 5))
G = load_golden("erosion_128x64_full")
rep("erosion pass", ctx.erosion_pass(grid, G["h"], G["flow"], G["updrift"], 1.0, 1.0), G["out"], orc.erosion_pass(g, G["h"], G["flow"], G["updrift"], 1.0, 1.0))
G = load_golden("thermal_128x64_full"); rep("thermal erosion", ctx.thermal_erosion(grid, G["h"]), G["out"], orc.thermal_erosion(g, G["h"]))
G = load_golden("morph_128x64_full"); rep("morphological erode r5.5", ctx.morphological(grid, G["field"], 5.5, 0), G["erode_r5"])
