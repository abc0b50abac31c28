"""Worker for tests/test_gpu_filters.py::test_climate_kernel_forms_agree: the climate step's kernel form is chosen once per process
(DMG_CLIMATE_ROWS rows per wave, DMG_CLIMATE_NO_SEAMS), hence a child process per form.  Prints one CRC-32 per case:
full-longitude and regional canvases, widths that are and are not a multiple of 1024, fields that stay finite (the packed form) and
fields that blow up (the generic form), one band and three bands with shrinking rows between exchanges.

usage: python tests/climate_worker.py          (the form comes from the environment)
"""
import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

f32 = np.float32


def field(W, H, seed):
    rng = np.random.default_rng(seed)
    lat = np.linspace(-1, 1, H, dtype=f32)[:, None]
    T = (15 - 40 * np.abs(lat) + 3 * np.sin(np.linspace(0, 12 * np.pi, W, dtype=f32))[None, :]).astype(f32)
    T += rng.standard_normal((H, W)).astype(f32) * f32(0.01)
    h = rng.standard_normal((H, W)).astype(f32)
    return T, h


def main():
    from demiurge_amd import capi
    ctx = capi.Context(0)
    out = []
    cases = [("globe 2048x1024", 2048, 1024, capi.FULL_GLOBE), ("globe 1536x640", 1536, 640, capi.FULL_GLOBE),
             ("belt 1280x512", 1280, 512, (-1.0, 1.2, -2.0, 2.5)), ("globe 520x260", 520, 260, capi.FULL_GLOBE)]
    stable = capi.climate_params(diffusivity=0.55e2)
    for name, W, H, coords in cases:
        g = capi.make_grid(W, H, coords)
        T, h = field(W, H, W)
        for pname, prm, n in (("stable", stable, 7), ("reference", None, 40)):
            k = capi.Climate(ctx, g, T, h, params=prm)
            k.steps(0, n)
            o = k.get()
            k.close()
            out.append("%s %s %08x" % (name, pname, zlib.crc32(o.tobytes())))
    # three bands on this GPU, 4 steps between halo refreshes
    W, H = 2048, 1024
    g = capi.make_grid(W, H)
    T, h = field(W, H, 7)
    bands = [capi.Climate(ctx, g, T, h, params=stable, band_rank=r, band_count=3, ghost=4) for r in range(3)]
    for b in bands:
        b.steps(0, 4)
    full = np.zeros((H, W), f32)
    for b in bands:
        b.get(out=full)                                # a band's owned rows into the full-grid array
        b.close()
    out.append("three bands %08x" % zlib.crc32(full.tobytes()))
    print("\n".join(out))


if __name__ == "__main__":
    main()
