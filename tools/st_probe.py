"""How llvmpipe interpolates the `st` varying of the reference's full-screen pass -- the whole reason results on
non-power-of-two grids are not bit-equal to (x+0.5)/W arithmetic.  Model (verified here against a shader that writes
`st`, 100 % of pixels at every size tried):

    the quad is two triangles split along the anti-diagonal; pixel (x, y) belongs to the lower-left one when
    (x+0.5)/W + (y+0.5)/H < 1
    ooa  = fl(1 / fl(W*H))                       one over the triangle's doubled area, in pixels
    dadx = fl(H * ooa)    (st.x per pixel in x)  -- equals fl(1/W) at most sizes, one ulp less at e.g. W = 333
    dady = fl(W * ooa)    (st.y per pixel in y)
    a0   = fl(a(v0) - fl(dadx * fl(x0 - 0.5)))   v0 = (0,0), a = 0 for the lower-left triangle; (W,H), a = 1 for the other
    st.x = fma(dadx, x, a0)                      x the integer pixel coordinate; likewise st.y

On power-of-two grids this is exactly (x+0.5)/W everywhere; elsewhere it is up to one ulp OF 1.0 off (the upper-right
triangle is anchored at a = 1), and differs between the two triangles of the same row.  Needs oracle/_ref/librefgl.so.

    python tools/st_probe.py [W H ...]
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "oracle" / "refgl"))
import refgl  # noqa: E402

f = np.float32
SRC = "#version 430\nin vec2 st; out vec2 fc;\nvoid main(){ fc = st; }\n"


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f)


def model(W, H):
    ooa = f(1) / f(f(W) * f(H))
    d, e = f(f(H) * ooa), f(f(W) * ooa)
    xi = np.arange(W, dtype=f)[None, :].repeat(H, 0)
    yi = np.arange(H, dtype=f)[:, None].repeat(W, 1)
    lower = ((xi.astype(np.float64) + 0.5) / W + (yi.astype(np.float64) + 0.5) / H) < 1
    a0 = lambda a, k, x0: f(f(a) - f(k * f(f(x0) - f(0.5))))
    sx = np.where(lower, fma(d, xi, a0(0, d, 0)), fma(d, xi, a0(1, d, W)))
    sy = np.where(lower, fma(e, yi, a0(0, e, 0)), fma(e, yi, a0(1, e, H)))
    return sx, sy


sizes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or \
        [(1000, 500), (100, 50), (96, 64), (300, 200), (127, 63), (768, 384), (1920, 1080), (333, 111), (640, 360), (512, 256), (1023, 511)]
g = refgl.RefGL.get()
for W, H in sizes:
    g.set_canvas(refgl.FULL_GLOBE, 42000.0)
    t = g.tex(W, H, 2, False)
    g.apply(g.program_source(SRC), t, {})
    r = g.download(t)
    g.delete(t)
    sx, sy = model(W, H)
    ex = ((np.arange(W, dtype=f) + f(.5)) / f(W))[None, :]
    ey = ((np.arange(H, dtype=f) + f(.5)) / f(H))[:, None]
    print("%5dx%-5d model == llvmpipe: st.x %.5f st.y %.5f of the pixels;  (x+0.5)/W == llvmpipe: %.4f, (y+0.5)/H: %.4f" %
          (W, H, (sx == r[..., 0]).mean(), (sy == r[..., 1]).mean(), (r[..., 0] == ex).mean(), (r[..., 1] == ey).mean()))
