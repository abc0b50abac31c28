"""Worker for tests/test_gpu_diffusion_tb.py: runs the blocked diffusion kernel with a forced tile height
(DMG_DTB_ROWS is read once per process) and checks it against the oracle.  Exit code 0 = bit-identical."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


def cases():
    from demiurge_amd import synth
    f32 = np.float32
    FULL = (f32(-np.pi / 2), f32(np.pi / 2), f32(-np.pi), f32(np.pi))
    BELT = (f32(-1.0), f32(1.2), f32(-np.pi), f32(np.pi))          # wraps in longitude, no poles: CLAMP_TO_EDGE rows
    W, H = 512, 256
    img = synth.terrain(W, H)
    v = (synth.velocity(W, H) * f32(3)).astype(f32)
    yield "full globe", FULL, img, v, (1, 7, 8, 9, 50)
    yield "belt", BELT, img, v, (8, 21)
    # tiny velocities: ocean patches in a land world holding values down to the denormal range, where the
    # reciprocal-based division is not the IEEE quotient (the kernel must notice and divide exactly); zero elsewhere.
    # Every 256-column span of every patch row contains land: wave-rows without any land are not checked
    land = np.ones((H, W), f32)
    vs = np.zeros((H, W, 2), f32)
    for (y0, y1, x0, x1), scale in (((200, 216, 300, 340), 1.0), ((30, 40, 10, 60), 1e-29), ((100, 130, 100, 400), 1e-33),
                                    ((2, 9, 200, 420), 1e-37), ((240, 256, 20, 200), 3e-39)):
        land[y0:y1, x0:x1] = -1
        vs[y0:y1, x0:x1] = (v[y0:y1, x0:x1].astype(np.float64) * scale).astype(f32)
    yield "tiny velocities", FULL, land, vs, (8, 40)
    # a pole row that is ocean next to land across the pole, rough field
    rough = (synth.lcg_uniform(W * H * 2, 5).reshape(H, W, 2) * 10).astype(f32)
    yield "rough field", FULL, img, rough, (16,)


def main():
    import oracle
    from demiurge_amd import capi
    ctx = capi.Context(0)
    bad = 0
    for name, coords, img, v, sweeps in cases():
        H, W = img.shape
        grid = capi.make_grid(W, H, coords, 42000.0)
        og = oracle.grid(W, H, coords, 42000.0)
        for quirks in (None, 0):
            prm = capi.default_ocean_params()
            if quirks is not None:
                prm.quirks = quirks
            for n in sweeps if quirks is None else sweeps[:1]:
                o = capi.Ocean(ctx, grid, params=prm)
                o.set_terrain(img)
                o.set_velocity(v)
                o.diffusion(n)
                got = o.get(want=("v",))["v"]
                o.close()
                want = oracle.ocean_diffusion(og, img, v, n, scalar_rhs=quirks is None)
                ok = np.array_equal(got, want)
                tiny = int(((np.abs(want) > 0) & (np.abs(want) < 2.0 ** -100)).sum())
                print("%-20s quirks=%s sweeps=%2d  %s  (results in (0, 2^-100): %d)" % (name, quirks, n, "ok" if ok else "MISMATCH %d, first at %s" % ((got != want).sum(), np.argwhere(got != want)[:3].tolist()), tiny))
                bad += not ok
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
