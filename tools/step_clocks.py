"""Per-step clocks of k_jacobi_stream's units from a DMG_ST_STEPPROF file (experiments build, -DST_EXPERIMENTS):
medians over the units with the most common step count.  usage: python tools/step_clocks.py <file> [iterations]"""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], delimiter=",", dtype=np.uint64).astype(np.int64)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
top, arrived, done, end = a[:, 0:22], a[:, 22:43], a[:, 43:63], a[:, 63]
its = (top > 0).sum(1)
mode = np.bincount(its).argmax()
s = its == mode
print("units %d, of which %d with %d+ recorded iterations; unit time median %.2f us" % (len(a), s.sum(), mode, np.median((end[s] - top[s, 0]) / 100.0)))
top, arrived, done = top[s], arrived[s], done[s]
m = min(n, mode - 1)
f = lambda x: " ".join("%.2f" % v for v in x)
print("even step, wait            :", f(np.median(arrived[:, :m] - top[:, :m], 0) / 100.0))
print("even step, arrive..levels  :", f(np.median(done[:, :m] - arrived[:, :m], 0) / 100.0))
print("odd step, all of it        :", f(np.median(top[:, 1:m + 1] - done[:, :m], 0) / 100.0))
