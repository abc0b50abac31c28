#!/usr/bin/env python3
"""The shader clock the headline kernel really runs at, measured INSIDE the kernel (MI355X_MICROARCH.md, DVFS: delta s_memtime /
delta s_memrealtime in a diagnostic build -- board power and sclk from rocm-smi are not the test): an experiments build
(make EXTRA=-DST_EXPERIMENTS) stamps both counters at every unit's start and end of ONE launch of the 5000-sweep solve at
16384x8192 after >= 2 s of back-to-back launches; rocm-smi is sampled beside it for comparison.
    DMG_LIB=demiurge_amd/libdemiurge_hip_exp.so python tools/kernel_clock.py [survey|ocean]"""
import csv, os, re, subprocess, sys, threading, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
terrain = sys.argv[1] if len(sys.argv) > 1 else "survey"
out = ROOT / "gpurun_out" / ("kernel_clock_%s.csv" % terrain)
os.environ["DMG_ST_PROF"] = str(out)
os.environ["DMG_ST_PROF_SKIP"] = "9000"                     # launches of more than 1024 units before the recorded one: ~2.5 s of them
sys.path.insert(0, str(ROOT))
import numpy as np
import torch  # noqa: F401
from demiurge_amd import capi, synth
W, H = 16384, 8192
ctx = capi.Context(0)
o = capi.Ocean(ctx, capi.make_grid(W, H))
o.set_terrain(synth.terrain(W, H) if terrain == "survey" else -np.ones((H, W), np.float32))
o.set_divw(synth.divw(W, H))
samples, stop = [], threading.Event()


def watch():
    while not stop.is_set():
        t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk = re.search(r"GPU\[0\]\s*: sclk clock level: \S+ \((\d+)Mhz\)", t)
        pw = re.search(r"GPU\[0\]\s*: Current Socket Graphics Package Power \(W\): ([\d.]+)", t)
        samples.append((time.perf_counter(), int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0))
        time.sleep(0.3)


th = threading.Thread(target=watch); th.start()
time.sleep(1.0)
t_busy0 = time.perf_counter()
times = []
for i in range(20):                                          # 20 solves = 12500 launches ~ 3.5 s, back to back
    t0 = time.perf_counter(); o.zero_pressure(); o.jacobi_sweeps(5000); ctx.sync(); times.append(time.perf_counter() - t0)
t_busy1 = time.perf_counter()
stop.set(); th.join()
o.close()
rows = [r for r in csv.DictReader(open(out)) if int(r["t1"]) > 0 and int(r["c1"]) > 0]
rt = np.array([int(r["t1"]) - int(r["t0"]) for r in rows], float)          # 100 MHz ticks
sc = np.array([int(r["c1"]) - int(r["c0"]) for r in rows], float)          # shader clocks
mhz = sc / rt * 100.0
span_rt = max(int(r["t1"]) for r in rows) - min(int(r["t0"]) for r in rows)
q = lambda x: " ".join("%.0f" % v for v in np.percentile(x, [0, 10, 50, 90, 100]))
print("terrain %s: solves %.1f .. %.1f ms (median %.1f); the recorded launch is the 9001st of more than 1024 units" % (terrain, min(times) * 1e3, max(times) * 1e3, np.median(times) * 1e3))
print("units %d; launch span %.1f us (s_memrealtime is one counter for the chip; s_memtime is not comparable between units, only within one)" % (len(rows), span_rt / 100.0))
print("shader clock per unit, delta s_memtime / delta s_memrealtime x 100 MHz (min p10 p50 p90 max): %s MHz" % q(mhz))
print("time-weighted mean over the units: %.0f MHz -> the launch's span is %.0f k shader clocks per SIMD" % (sc.sum() / rt.sum() * 100.0, span_rt / 100.0 * sc.sum() / rt.sum() * 100.0 / 1e3))
busy = [(s, p) for (t, s, p) in samples if t_busy0 + 0.5 < t < t_busy1]
if busy:
    print("rocm-smi beside it (%d samples inside the busy stretch): sclk median %.0f MHz (%.0f .. %.0f), package power median %.0f W" %
          (len(busy), np.median([b[0] for b in busy]), min(b[0] for b in busy), max(b[0] for b in busy), np.median([b[1] for b in busy])))
