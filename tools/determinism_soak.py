"""Run the hot paths repeatedly on the same inputs and compare the bits of every repetition with the first (GPU box).
usage: python tools/determinism_soak.py [repetitions]"""
import hashlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # (the overlapped order: the library no longer sets it itself)
import numpy as np
sys.path.insert(0, ".")
from demiurge_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
ctx = capi.Context(0)
ok = True

W, H = 16384, 8192
grid = capi.make_grid(W, H)
o = capi.Ocean(ctx, grid)
o.set_terrain(synth.terrain(W, H)); o.set_divw(synth.divw(W, H))
ref = None
t0 = time.perf_counter()
for i in range(n):
    o.zero_pressure(); o.jacobi_sweeps(1000)
    h = sha(o.get(want=("pressure",))["pressure"])
    ref = ref or h
    ok &= h == ref
print("jacobi 16384x8192 x1000 sweeps, %d solves: %s (%s), %.1f s" % (n, "identical" if ok else "DIFFER", ref, time.perf_counter() - t0), flush=True)
o.close()

# round 6: the tolerance rendering through the deep form (two waves per unit that meet through LDS counters: the timing of the pair must
# not show in the bits)
prm = capi.default_ocean_params()
prm.quirks |= capi.QUIRK_JACOBI_TOLERANCE
o = capi.Ocean(ctx, grid, params=prm)
o.set_terrain(synth.terrain(W, H)); o.set_divw(synth.divw(W, H))
ref2, ok2 = None, True
t0 = time.perf_counter()
for i in range(n):
    o.zero_pressure(); o.jacobi_sweeps(1000)
    h = sha(o.get(want=("pressure",))["pressure"])
    ref2 = ref2 or h
    ok2 &= h == ref2
print("jacobi 16384x8192 x1000 sweeps, tolerance rendering (deep form), %d solves: %s (%s), %.1f s" % (n, "identical" if ok2 else "DIFFER", ref2, time.perf_counter() - t0), flush=True)
ok &= ok2
o.close()

for order, name in ((capi.MULTI_EXCHANGE_THEN_SWEEP, "exchange-then-sweep"), (capi.MULTI_OVERLAP_LAST_LAUNCH, "overlap")):
    for transport, tname in ((capi.MULTI_RCCL_LOOPBACK, "loopback"), (capi.MULTI_PEER, "peer"), (capi.MULTI_STORE, "store")):
        Wb, Hb = 16384, 4096
        m = capi.Multi(capi.make_grid(Wb, Hb), devices=[0] * 4, ghost=32, flags=transport | order)
        m.set_terrain(synth.terrain(Wb, Hb)); m.set_divw(synth.divw(Wb, Hb))
        r2, good = None, True
        t0 = time.perf_counter()
        for i in range(max(n // 2, 3)):
            m.jacobi(640, zero_first=True)
            h = sha(m.get(want=("pressure",))["pressure"])
            r2 = r2 or h
            good &= h == r2
        ok &= good
        print("4 bands x 1024 rows, %s, %s, 640 sweeps x %d: %s (%s), %.1f s" % (tname, name, max(n // 2, 3), "identical" if good else "DIFFER", r2, time.perf_counter() - t0), flush=True)
        m.close()

# the band of an 8-way split as it would run on eight GPUs (twin units, whole launches), eight of them on this one
Wb, Hb = 16384, 8192
m = capi.Multi(capi.make_grid(Wb, Hb), devices=[0] * 8, ghost=32, flags=capi.MULTI_PEER | capi.MULTI_EXCHANGE_THEN_SWEEP)
m.set_terrain(synth.terrain(Wb, Hb)); m.set_divw(synth.divw(Wb, Hb))
r2, good = None, True
t0 = time.perf_counter()
for i in range(max(n // 2, 3)):
    m.jacobi(640, zero_first=True)
    h = sha(m.get(want=("pressure",))["pressure"])
    r2 = r2 or h
    good &= h == r2
ok &= good
print("8 bands x 1024 rows, peer, exchange-then-sweep, 640 sweeps x %d: %s (%s), %.1f s" % (max(n // 2, 3), "identical" if good else "DIFFER", r2, time.perf_counter() - t0), flush=True)
m.close()

W, H = 4096, 2048
g = capi.make_grid(W, H)
hts = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
r3, good = None, True
t0 = time.perf_counter()
for i in range(max(n // 4, 3)):
    out = ctx.erosion_run(g, hts, 50, 20, 0.5, 1.0, 1.0)
    h = sha(out)
    r3 = r3 or h
    good &= h == r3
ok &= good
print("erosion 4096x2048 x20 rounds x %d: %s (%s), %.1f s" % (max(n // 4, 3), "identical" if good else "DIFFER", r3, time.perf_counter() - t0), flush=True)
Wc, Hc = 8192, 4096
T0 = (synth.lcg_uniform(Wc * Hc, 41).reshape(Hc, Wc) * 60).astype(np.float32)
terr = (synth.terrain(Wc, Hc) * 100).astype(np.float32)
r4, good = None, True
for i in range(3):
    k = capi.Climate(ctx, capi.make_grid(Wc, Hc), T0, terr)
    k.steps(0, 40)
    h = sha(np.nan_to_num(k.get(), nan=-12345.0, posinf=1e38, neginf=-1e38))
    k.close()
    r4 = r4 or h
    good &= h == r4
ok &= good
print("climate 8192x4096 x40 steps x 3: %s (%s)" % ("identical" if good else "DIFFER", r4), flush=True)
print("ALL IDENTICAL" if ok else "NOT DETERMINISTIC")
sys.exit(0 if ok else 1)
