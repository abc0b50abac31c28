"""does a "per-row weights" rendering of the blur (every tap where the arithmetic means it to be, not where fp32
texture coordinates put it) stay within north_star's 1e-5 of the reference?  CPU only (the oracle, orc_set_blur_clean).
    python tools/blur_tolerance_eval.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import oracle
from demiurge_amd import synth

L = oracle.lib()
L.orc_set_blur_clean.argtypes = [ctypes.c_int]


def rel(a, ref):
    return float(np.max(np.abs(a.astype(np.float64) - ref)) / np.max(np.abs(ref)))


def both(g, field, radius):
    L.orc_set_blur_clean(0); lit = oracle.blur(g, field, radius)
    L.orc_set_blur_clean(1); cln = oracle.blur(g, field, radius)
    L.orc_set_blur_clean(0)
    return lit, cln


G = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "blur_128x64_r10.npz"))
print("golden keys:", [k for k in G.files])
g = oracle.grid(128, 64, [float(x) for x in G["coords"]], float(G["circ"]))
key_in = "field"
lit, cln = both(g, G[key_in], 10.0)
out = ["out"]
print("128x64 radius 10 (llvmpipe golden's input): literal vs golden %.3g; clean weights vs golden %.3g" % (rel(lit, G[out[0]]), rel(cln, G[out[0]])))
rng = np.random.default_rng(3)
for W, H, what in ((1024, 512, "terrain"), (1024, 512, "rough"), (4096, 2048, "terrain"), (4096, 2048, "rough")):
    g = oracle.grid(W, H)
    f = (synth.terrain(W, H) * 1000).astype(np.float32) if what == "terrain" else rng.standard_normal((H, W)).astype(np.float32)
    for radius in (1.0, 10.0):
        lit, cln = both(g, f, radius)
        print("%dx%d %s field, radius %g: clean weights vs literal %.3g of the output's maximum" % (W, H, what, radius, rel(cln, lit)), flush=True)
