"""What does the GPU clock at while the headline's launches run back to back?  A thread samples rocm-smi (sclk, package power) every
half second while the main thread runs 5000-sweep solves at 16384x8192 (or, with `climate`, climate steps) for ~20 s.
usage (GPU box): python tools/clock_probe.py [jacobi|sweep1|climate|copy|idle]        (sweep1: the one-sweep-per-launch kernel)"""
import re
import subprocess
import sys
import threading
import time
from pathlib import Path

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from demiurge_amd import capi, synth  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "jacobi"
samples, stop = [], threading.Event()
t00 = time.perf_counter()


def watch():
    while not stop.is_set():
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk = re.search(r"GPU\[0\]\s*: sclk clock level: \S+ \((\d+)Mhz\)", out)
        pw = re.search(r"GPU\[0\]\s*: Current Socket Graphics Package Power \(W\): ([\d.]+)", out)
        samples.append((time.perf_counter() - t00, int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0))
        time.sleep(0.4)


W, H = 16384, 8192
ctx = capi.Context(0)
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100).astype(np.float32)
th = threading.Thread(target=watch)
marks = []
if what in ("jacobi", "sweep1"):
    prm = capi.default_ocean_params()
    if what == "sweep1":
        prm.jacobi_kernel = capi.JACOBI_SINGLE
    o = capi.Ocean(ctx, g, params=prm)
    o.set_terrain(h)
    o.set_divw(synth.lcg_uniform(W * H, 3).reshape(H, W).astype(np.float32) - 0.5)
    o.zero_pressure(); o.jacobi_sweeps(5000); ctx.sync()
    th.start()
    time.sleep(2.0)
    for i in range(100 if what == "jacobi" else 12):
        t0 = time.perf_counter()
        o.jacobi_sweeps(5000); ctx.sync()
        marks.append((t0 - t00, time.perf_counter() - t0))
elif what == "climate":
    lat = np.linspace(-1, 1, H, dtype=np.float32)[:, None]
    T0 = (15 - 40 * np.abs(lat) + 3 * np.sin(np.linspace(0, 12 * np.pi, W, dtype=np.float32))[None, :]).astype(np.float32)
    k = capi.Climate(ctx, g, T0, h, params=capi.climate_params(diffusivity=0.55e4))
    k.steps(0, 10); ctx.sync()
    th.start()
    time.sleep(2.0)
    for i in range(60):
        t0 = time.perf_counter()
        k.steps(10 + 1000 * i, 1000); ctx.sync()
        marks.append((t0 - t00, time.perf_counter() - t0))
elif what == "copy":
    a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    torch.cuda.synchronize()
    th.start()
    time.sleep(2.0)
    for i in range(60):
        t0 = time.perf_counter()
        for _ in range(500):
            b.copy_(a)
        torch.cuda.synchronize()
        marks.append((t0 - t00, time.perf_counter() - t0))
else:
    th.start()
    time.sleep(6.0)
time.sleep(2.0)
stop.set(); th.join()
print("work (start s, duration s):", " ".join("%.1f:%.3f" % m for m in marks[::10]))
for t, s, p in samples:
    busy = any(a <= t <= a + d for a, d in marks)
    print("t=%6.2f s  sclk %5d MHz  power %6.1f W  %s" % (t, s, p, "busy" if busy else "idle"))
