"""Times the climate step (Temperature.cpp:56-134) per step at WxH with the field resident on the device (dmg_climate).

    python tools/climate_time_probe.py [W]                verified per-row tap table (default)
    DMG_TEMPERATURE_LITERAL=1 python tools/...            literal offset() per tap and cell
"""
import os
import sys
import time
import zlib
from pathlib import Path

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from demiurge_amd import capi, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = W // 2
ctx = capi.Context(0)
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100).astype(np.float32)
# A smooth field and few steps: with the reference's constants the explicit scheme is unstable at this resolution (its
# diffusion number is ~10: rounding noise grows tenfold per step, a field of O(1) noise overflows fp32 in ~38 steps), and a
# field of infinities and NaNs takes the IEEE-division fallback in every cell -- not what the step costs on a field that
# means something.  DMG_CLIMATE_PROBE_STEPS=120 times that regime.
lat = np.linspace(-1, 1, H, dtype=np.float32)[:, None]
T0 = (15 - 40 * np.abs(lat) + 3 * np.sin(np.linspace(0, 12 * np.pi, W, dtype=np.float32))[None, :]).astype(np.float32)
if os.environ.get("DMG_CLIMATE_PROBE_T0") == "const":      # Temperature::Temperature's start: 50 everywhere
    T0 = np.full((H, W), 50, np.float32)
nsteps = int(os.environ.get("DMG_CLIMATE_PROBE_STEPS", "12"))
# ... and with values the scheme is stable with at this resolution (dmg_climate_params: the diffusivity a hundredth of the
# reference's: diffusion number 0.12 over land), where any number of steps stays a climate: 60 steps
stable = capi.climate_params(diffusivity=0.55e4)
for name, prm, n in (("reference values, smooth field, %d steps" % nsteps, None, nsteps), ("stable values (diffusivity 0.55e4), 60 steps", stable, 60)):
    best = 1e9
    for _ in range(3):
        k = capi.Climate(ctx, g, T0, h, params=prm)     # field resident on the device: the timed region holds kernels only
        k.steps(0, 2); ctx.sync()
        ctx.timer_start()
        k.steps(2, n)
        best = min(best, ctx.timer_stop() / n)
        out = k.get()
        k.close()
    print("%dx%d climate step, %s: %.3f ms (%.0f GB/s at 8.25 B per cell)  literal=%s  rows per wave=%s  field finite: %s, |T| <= %.3g, crc %08x"
          % (W, H, name, best, W * H * 8.25 / best / 1e6, os.environ.get("DMG_TEMPERATURE_LITERAL", "0"), os.environ.get("DMG_CLIMATE_ROWS", "default"),
             bool(np.isfinite(out).all()), float(np.nanmax(np.abs(out))), zlib.crc32(out.tobytes())))
