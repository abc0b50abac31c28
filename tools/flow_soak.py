"""The flow solve's sums run as concurrent pieces since round 6 (catchments on two streams, a persistent grid walking the reaches, marking walks
that race for cells): the same fields, bit for bit, run after run?  SHA-256 of the flow field of repeated solves of the same terrains, and of
the heights after repeated erosion runs.  usage (GPU box): python tools/flow_soak.py [repeats]"""
import sys, hashlib, time
import numpy as np
sys.path.insert(0, ".")
from demiurge_amd import capi, synth
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = capi.Context()
ok = True
for W, H, seed, amp in ((1024, 512, 3, 8.0), (4096, 2048, 5, 3.0), (16384, 8192, 11, 6.0)):
    g = capi.make_grid(W, H)
    h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, seed).reshape(H, W) * amp).astype(np.float32)
    hs = set()
    t0 = time.perf_counter()
    for r in range(rep if W < 16384 else max(2, rep // 2)):
        f = ctx.flow_accumulate(g, h, 0.5, 0.5, lakes=r & 1)
        hs.add((r & 1, hashlib.sha256(f.tobytes()).hexdigest()))
    same = len(hs) == 2
    ok = ok and same
    print("flow %dx%d: %d solves (lake drawing off / on alternating), %d distinct fields: %s  (%.1f s)" %
          (W, H, rep if W < 16384 else max(2, rep // 2), len(hs), "identical run to run" if same else "DIFFER", time.perf_counter() - t0), flush=True)
W, H = 4096, 2048
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
hs = set()
for r in range(rep):
    out = ctx.erosion_run(g, h, 50, 60, 0.5, 1.0, 1.0)
    hs.add(hashlib.sha256(np.asarray(out).tobytes()).hexdigest())
ok = ok and len(hs) == 1
print("erosion %dx%d, 60 iterations, %d runs: %d distinct results: %s" % (W, H, rep, len(hs), "identical run to run" if len(hs) == 1 else "DIFFER"))
print("SOAK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
