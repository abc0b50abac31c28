"""Where does the climate step leave the reference's pixels?  Cuts the reference's Temperature shader (lifted by
oracle/refgl at build time) after each intermediate, runs the cut shader on llvmpipe over a rough temperature field and
several days of the year, and compares with the same quantity restated in numpy / the oracle (orc_qday_parts).

This is how these forms were found (each 0 mismatches; the literal forms 0.1 - 50 %):
    ASR   = ((S/pi) * 0.7) * X                  for (1-alpha) * (S/M_PI * X)
    OLR   = t2 * (t2 * 210) / powf(273.4,4) * 0.93   for 210*pow(T+273.15,4)/pow(273.4,4)*0.93  (t2 = t*t)
    fc    = T + change * (3.154*1e7) / 15000 / C     for T + change*3.154*1e7/15000/C
    acos  = Mesa's expansion with its own literals (pi/4 - 1 taken in float)

Needs /root/reference (oracle/_ref/librefgl.so); CPU only.      python tools/climate_probe.py
"""
import ctypes
import ctypes.util
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
for p in ("oracle", "oracle/refgl", "tests"):
    sys.path.insert(0, str(ROOT / p))
import oracle  # noqa: E402
import refgl  # noqa: E402
from conftest import load_golden  # noqa: E402

f = np.float32
G = load_golden("temperature_128x64_rough")
T, terr = G["T0"], G["terrain"]
H, W = T.shape
coords, circ = tuple(G["coords"]), float(G["circ"])
g = refgl.RefGL.get()
src = g.shader_code(refgl.K_TEMPERATURE)
L = oracle.lib()
L.orc_qday_parts.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
libm = ctypes.CDLL(ctypes.util.find_library("m"))
libm.powf.restype = ctypes.c_float
libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]


def upto(marker, tail):
    return src[:src.index(marker)] + tail + " }\n"


def run(fs, M):
    g.set_canvas(coords, circ)
    t_T, t_s, t_h = g.tex(W, H, 1, False, T), g.tex(W, H, 1, False), g.tex(W, H, 1, False, terr)
    p = g.program_source(fs)
    g.uniform1f(p, "M", M)
    g.apply(p, t_s, {"img": t_T, "scratch1": t_h})
    r = g.download(t_s)
    g.delete(t_T, t_s, t_h)
    return r


GRAD = "\tvec2 gradient = get_texture_laplacian"
CHANGE = "\tfloat change = ASR"
tot = {}
for step in (0, 2, 5, 7, 333, 3000, 7000, 14000):
    M = f(np.pi * 2 / 15000 * step)
    asr = run(upto("\tfloat OLR = 210 + 2 * T;", "fc = ASR;"), M)
    olr = run(upto(GRAD, "fc = OLR;"), M)
    gs = run(upto(CHANGE, "fc = 0.55*1e6*(gradient.x + gradient.y);"), M)
    ch = run(upto("    float atmosphere = 1e7;", "fc = change;"), M)
    full = run(src, M)
    parts = np.zeros(7, f)
    o_asr = np.zeros((H, 1), f)
    for y in range(H):
        lat = f(f((y + 0.5) / H) * f(coords[1] - coords[0]) + coords[0])
        L.orc_qday_parts(ctypes.c_float(lat), ctypes.c_float(M), ctypes.c_void_p(parts.ctypes.data))
        o_asr[y] = parts[6]
    t = T + f(273.15)
    t2 = t * t
    c4 = f(libm.powf(273.4, 4.0))
    atm = f(1e7)
    C = atm + np.where(terr > 0, atm * f(0.5), f(4 * 1.5) * atm).astype(f)
    checks = {
        "ASR = ((S/pi)*0.7)*X (oracle)": np.broadcast_to(o_asr, asr.shape) != asr,
        "OLR = t2*(t2*210)/powf(273.4,4)*0.93": ((t2 * (t2 * f(210))) / c4 * f(0.93)) != olr,
        "OLR literal 210*(t2*t2)/... (for contrast)": ((f(210) * (t2 * t2)) / c4 * f(0.93)) != olr,
        "change = (ASR-OLR) + 0.55e6*(gx+gy)": ((asr - olr) + gs) != ch,
        "fc = T + change*(3.154*1e7)/15000/C": (T + (ch * f(f(3.154) * f(1e7))) / f(15000) / C) != full,
        "fc literal (change*3.154)*1e7 (for contrast)": (T + ((ch * f(3.154)) * f(1e7)) / f(15000) / C) != full,
        "whole step, oracle": oracle.temperature(oracle.grid(W, H, coords, circ), T, terr, step, 1) != full,
    }
    for k, bad in checks.items():
        tot[k] = tot.get(k, 0) + int(bad.sum())
for k, n in tot.items():
    print("%-48s differing cells %6d of %d" % (k, n, 8 * W * H))
