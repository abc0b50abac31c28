"""how far is a fused-coefficient rendering of the Jacobi sweep from the reference after its 5000 sweeps?
CPU only (the oracle).  max|d| / max|ref| against the llvmpipe golden (256x128) and against the exact form at larger sizes.
    python tools/jacobi_tolerance_eval.py [--big 2048]
"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import oracle
from demiurge_amd import synth

FORMS = {0: "exact (two-operation quotients = IEEE divisions)", 1: "three rounded coefficients, 2 fma", 2: "split coefficients, 4 fma",
         3: "split, residues folded on N/S, 3 fma", 4: "plain reciprocal multiplications"}

def rel(a, ref):
    return float(np.max(np.abs(a.astype(np.float64) - ref.astype(np.float64))) / np.max(np.abs(ref.astype(np.float64))))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=5000)
    ap.add_argument("--forms", default="1,2,3,4")
    a = ap.parse_args()
    forms = [int(x) for x in a.forms.split(",")]
    G = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "jacobi_256x128_full.npz"))
    g = oracle.grid(256, 128, [float(x) for x in G["coords"]], float(G["circ"]))
    for it, key in ((100, "p100"), (5000, "p5000")):
        ex = oracle.ocean_jacobi_form(g, G["img"], G["divw"], it, 0)
        print("256x128 x%d exact vs llvmpipe golden: %.3g" % (it, rel(ex, G[key])), flush=True)
        for f in forms:
            r = oracle.ocean_jacobi_form(g, G["img"], G["divw"], it, f)
            print("256x128 x%d form %d (%s): vs golden %.3g  vs exact %.3g" % (it, f, FORMS[f], rel(r, G[key]), rel(r, ex)), flush=True)
    if a.big:
        W, H = a.big, a.big // 2
        g = oracle.grid(W, H)
        img, divw = synth.terrain(W, H), synth.divw(W, H)
        t0 = time.time()
        ex = oracle.ocean_jacobi_form(g, img, divw, a.iters, 0)
        print("%dx%d x%d exact: %.1f s" % (W, H, a.iters, time.time() - t0), flush=True)
        for f in forms:
            r = oracle.ocean_jacobi_form(g, img, divw, a.iters, f)
            print("%dx%d x%d form %d (%s): vs exact %.3g" % (W, H, a.iters, f, FORMS[f], rel(r, ex)), flush=True)

if __name__ == "__main__":
    main()
