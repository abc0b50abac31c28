"""Worker for tests/test_gpu_filters.py::test_temperature_tap_table: runs the climate step on a few grids and saves the
fields (DMG_TEMPERATURE_LITERAL is read once per process, so the literal and the tabled path need a process each)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def cases():
    from demiurge_amd import synth
    f32 = np.float32
    FULL = (f32(-np.pi / 2), f32(np.pi / 2), f32(-np.pi), f32(np.pi))
    REGION = (f32(-0.6), f32(0.9), f32(-1.0), f32(2.0))
    BELT = (f32(-1.2), f32(1.0), f32(-np.pi), f32(np.pi))
    for name, W, H, coords in (("full_1000x500", 1000, 500, FULL), ("full_512x256", 512, 256, FULL), ("region_300x200", 300, 200, REGION),
                               ("belt_256x96", 256, 96, BELT)):
        T = ((synth.lcg_uniform(W * H, 41).reshape(H, W)) * 60).astype(f32)
        yield name, W, H, coords, T, synth.terrain(W, H)
    # values the reciprocal-based divisions of the kernel are not exact for: they must take the IEEE path
    W, H = 256, 128
    T = ((synth.lcg_uniform(W * H, 7).reshape(H, W)) * 40).astype(f32)
    T[10:20, 30:60] = 0.0
    T[30:40, :] *= f32(1e-33)
    T[50:52, 100:140] = f32(3e-39)                     # denormals
    T[60, 10] = np.inf; T[62, 50] = -np.inf; T[64, 90] = np.nan
    T[70:74, 200:220] = f32(2e38); T[80:84, 20:40] = f32(-1e30)
    T[90:100, :] = f32(-273.15)                        # t = 0: the OLR numerator is exactly 0
    T[100:104, 64:128] = f32(-273.15) + f32(1e-5)
    yield "weird_256x128", W, H, FULL, T, synth.terrain(W, H)


def main(out):
    from demiurge_amd import capi
    ctx = capi.Context(0)
    res = {}
    for name, W, H, coords, T, terr in cases():
        res[name] = ctx.temperature_steps(capi.make_grid(W, H, coords, 42000.0), T, terr, 11, 3)
    np.savez(out, **res)
    ctx.close()


if __name__ == "__main__":
    main(sys.argv[1])
