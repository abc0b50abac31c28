"""Twin units against plain units, bit for bit, at sizes the oracle does not reach in seconds: the same solves run in two
child processes (DMG_ST_TWIN=2 forces twin units on every launch of up to DMG_ST_TWIN_ROWS rows, DMG_ST_TWIN=0 keeps plain
units) and the SHA-256 of the resulting pressure fields must agree.  Whole grids of 16384 columns at several heights, and the
16384x8192 grid in 2, 4 and 8 latitude bands on this one GPU (dmg_multi, peer copies and direct stores, both exchange orders:
the overlapped order runs the split launches whose send zones are short twin pairs).
usage: python tools/twin_vs_plain.py            (parent)        python tools/twin_vs_plain.py child      (one of the two runs)"""
import hashlib
import os
import subprocess
import sys

sys.path.insert(0, ".")


def child():
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import numpy as np
    from demiurge_amd import capi, synth
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:20]
    ctx = capi.Context(0)
    rng = np.random.default_rng(5)
    W = 16384
    for H, terrain, sweeps in ((1024, "survey", 64), (3000, "survey", 40), (4096, "ocean", 24), (4400, "survey", 24), (777, "pepper", 48), (2048, "stripes", 32)):
        grid = capi.make_grid(W, H)
        if terrain == "survey":
            img = synth.terrain(W, H)
        elif terrain == "ocean":
            img = -np.ones((H, W), np.float32)
        elif terrain == "pepper":
            img = np.where(rng.random((H, W)) < 0.1, 1.0, -1.0).astype(np.float32)
        else:
            img = -np.ones((H, W), np.float32)
            for x0 in range(0, W, 240):
                y = int(rng.integers(0, 12))
                while y < H:
                    y += int(rng.integers(1, 31))
                    h = int(rng.integers(1, 41))
                    img[y:y + h, x0:x0 + 240] = 1.0
                    y += h
        b = synth.divw(W, H)
        p0 = (synth.lcg_uniform(W * H, 17).reshape(H, W) - 0.5).astype(np.float32)
        o = capi.Ocean(ctx, grid)
        o.set_terrain(img); o.set_divw(b); o.set_pressure(p0)
        o.jacobi_sweeps(sweeps)
        print("grid %dx%d %s x%d [%s]: %s" % (W, H, terrain, sweeps, o.kernel_name(), sha(o.get(want=("pressure",))["pressure"])), flush=True)
        o.close()
    W, H = 16384, 8192
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    o = capi.Ocean(ctx, grid)
    o.set_terrain(img); o.set_divw(b); o.zero_pressure()
    o.jacobi_sweeps(256)
    print("16384x8192 in one band, 256 sweeps [%s]: %s" % (o.kernel_name(), sha(o.get(want=("pressure",))["pressure"])), flush=True)
    o.close()
    for n, ghost, transport, tname, order, oname in ((8, 32, capi.MULTI_PEER, "peer", capi.MULTI_EXCHANGE_THEN_SWEEP, "exchange-then-sweep"),
                                                    (8, 128, capi.MULTI_RCCL_LOOPBACK, "loopback", capi.MULTI_OVERLAP_LAST_LAUNCH, "overlap"),
                                                    (4, 32, capi.MULTI_STORE, "store", capi.MULTI_OVERLAP_LAST_LAUNCH, "overlap"),
                                                    (4, 64, capi.MULTI_PEER, "peer", capi.MULTI_EXCHANGE_THEN_SWEEP, "exchange-then-sweep"),
                                                    (2, 32, capi.MULTI_STORE, "store", capi.MULTI_OVERLAP_LAST_LAUNCH, "overlap"),
                                                    (2, 16, capi.MULTI_PEER, "peer", capi.MULTI_OVERLAP_LAST_LAUNCH, "overlap")):
        m = capi.Multi(grid, devices=[0] * n, ghost=ghost, params=capi.default_ocean_params(), flags=transport | order)
        m.set_terrain(img); m.set_divw(b)
        m.jacobi(256, zero_first=True); m.sync()
        print("16384x8192 in %d bands, ghost %d, %s, %s, 256 sweeps [%s]: %s" % (n, ghost, tname, oname, m.band(0).kernel_name(), sha(m.get(want=("pressure",))["pressure"])),
              flush=True)
        m.close()
    ctx.close()


def parent():
    outs = {}
    # the two runs side by side on the one GPU (two processes: well inside the box's limit; half the wall time of one after the other)
    procs = {name: subprocess.Popen([sys.executable, __file__, "child"], env=dict(os.environ, DMG_TB_SHAPE="stream", **env), stdout=subprocess.PIPE,
                                    stderr=subprocess.PIPE, text=True)
             for name, env in (("twin", {"DMG_ST_TWIN": "2", "DMG_ST_TWIN_ROWS": "4400"}), ("plain", {"DMG_ST_TWIN": "0"}))}
    for name, pr in procs.items():
        try:
            out, err = pr.communicate(timeout=1000)
        except subprocess.TimeoutExpired:
            for q in procs.values():
                q.kill()
            print(name, "TIMED OUT")
            sys.exit(2)
        if pr.returncode != 0:
            print(name, "FAILED", out[-1500:], err[-1500:])
            sys.exit(2)
        outs[name] = [l for l in out.splitlines() if "]: " in l]
    ok = len(outs["twin"]) == len(outs["plain"]) and len(outs["twin"]) > 0
    for a, b in zip(outs["twin"], outs["plain"]):
        same = a.rsplit(": ", 1)[1] == b.rsplit(": ", 1)[1]
        ok &= same
        print(("same  " if same else "DIFFER"), a, "|", b.split("[")[1].split("]")[0])
    band_hashes = {l.rsplit(": ", 1)[1] for l in outs["twin"] + outs["plain"] if l.startswith("16384x8192 in")}
    ok &= len(band_hashes) == 1                   # one band, 2, 4, 8 bands, every transport and order: the same field
    print("the 16384x8192 solve in 1, 2, 4 and 8 bands: %s" % ("one field" if len(band_hashes) == 1 else "DIFFERENT FIELDS %s" % sorted(band_hashes)))
    print("ALL IDENTICAL" if ok else "DIFFERENCES")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    child() if len(sys.argv) > 1 and sys.argv[1] == "child" else parent()
