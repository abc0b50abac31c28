"""CPU reference lines of BASELINE.md section 3, one JSON object per line.

  python tools/cpu_lines.py refgl <threads>     the reference's own GLSL on llvmpipe (needs oracle/_ref + Mesa)
  python tools/cpu_lines.py port <threads>      the OpenMP C restatement (oracle/), any host

TEST/BENCH INFRASTRUCTURE: times the checkers, never the product."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "oracle" / "refgl"))
from demiurge_amd import synth  # noqa: E402


def emit(**kw):
    print(json.dumps(kw), flush=True)


def refgl_lines(threads):
    os.environ["LP_NUM_THREADS"] = str(threads)
    import refgl
    rend = refgl.RefGL.get().renderer()
    # config 1: the whole Blur filter at 1024x512
    f = synth.terrain(1024, 512)
    for r in (1.0, 10.0):
        refgl.blur(f, r)
        t = []
        for _ in range(3):
            t0 = time.time(); refgl.blur(f, r); t.append(time.time() - t0)
        emit(line="llvmpipe blur 1024x512 radius %g" % r, seconds_per_filter=float(np.median(t)), threads=threads, renderer=rend)
    # Jacobi sweeps
    for (W, H), n in (((8192, 4096), 15 if threads > 1 else 4), ((16384, 8192), 10 if threads > 1 else 2)):
        img, b = synth.terrain(W, H), synth.divw(W, H)
        refgl.ocean_jacobi(img, b, 1)
        vals = []
        for _ in range(3 if threads > 1 else 1):
            t = []
            refgl.ocean_jacobi(img, b, n, timing=t)
            vals.append(W * H * n / t[0] / 1e6)
        emit(line="llvmpipe jacobi %dx%d" % (W, H), mcell_steps_per_s=float(np.median(vals)), sweeps=n, threads=threads, renderer=rend)
        del img, b
    # one ocean outer step at 1024x512, 100 Jacobi sweeps
    W, H = 1024, 512
    img = synth.terrain(W, H)
    v = synth.velocity(W, H); v[img > 0] = 0
    refgl.ocean_step(img, v, 1, 5, 5)
    t0 = time.time(); refgl.ocean_step(img, v, 1, 100, 50); dt = time.time() - t0
    emit(line="llvmpipe ocean outer step 1024x512 (advect + 50 diffusion + divergence + 100 jacobi + subDiv)", seconds=dt, threads=threads)


def port_lines(threads):
    import oracle
    oracle.set_threads(threads)
    for (W, H), n in (((8192, 4096), 10), ((16384, 8192), 6)):
        img, b = synth.terrain(W, H), synth.divw(W, H)
        g = oracle.grid(W, H)
        oracle.ocean_jacobi(g, img, b, 1, rows=True, nthreads=threads)
        t0 = time.time(); oracle.ocean_jacobi(g, img, b, n, rows=True, nthreads=threads); dt = time.time() - t0
        emit(line="C/OpenMP restatement jacobi %dx%d" % (W, H), mcell_steps_per_s=W * H * n / dt / 1e6, sweeps=n, threads=threads)
    W, H = 1024, 512
    f = synth.terrain(W, H)
    g = oracle.grid(W, H)
    for r in (1.0, 10.0):
        t0 = time.time(); oracle.blur(g, f, r); emit(line="C restatement blur 1024x512 radius %g" % r, seconds_per_filter=time.time() - t0, threads=threads)


if __name__ == "__main__":
    kind, threads = sys.argv[1], int(sys.argv[2])
    emit(host_cpus=os.cpu_count(), affinity=len(os.sched_getaffinity(0)))
    (refgl_lines if kind == "refgl" else port_lines)(threads)
