"""One launch's units from a DMG_ST_PROF file: when they start and end (100 MHz clock), how long they take.
usage: python tools/unit_times.py <file>"""
import sys
import numpy as np
import csv
rows = [r for r in csv.DictReader(open(sys.argv[1])) if int(r["t1"]) > 0]      # (place holders of a twin list record nothing)
t0 = np.array([int(r["t0"]) for r in rows]) / 100.0
t1 = np.array([int(r["t1"]) for r in rows]) / 100.0
h = np.array([int(r["r1"]) - int(r["r0"]) for r in rows])
mixed = np.array([int(r["mixed"]) for r in rows])
base = t0.min()
t0 -= base; t1 -= base
d = t1 - t0
q = lambda x: " ".join("%.1f" % v for v in np.percentile(x, [0, 10, 50, 90, 99, 100]))
print("units %d, heights %s, span %.1f us" % (len(rows), np.unique(h)[:12], t1.max()))
print("start  (min p10 p50 p90 p99 max):", q(t0))
print("end    (min p10 p50 p90 p99 max):", q(t1))
print("length (min p10 p50 p90 p99 max):", q(d))
for lo, hi in ((0, 1), (1, 8), (8, 1000)):
    s = (mixed >= lo) & (mixed < hi)
    if s.any():
        print("  units with %d..%d coast rows: %d, length p50 %.1f p90 %.1f, rows p50 %d" % (lo, hi - 1, s.sum(), np.median(d[s]), np.percentile(d[s], 90), np.median(h[s])))
late = np.argsort(t1)[-8:]
print("last to end:", [(int(rows[i]["tx"]), int(rows[i]["r0"]), int(rows[i]["r1"]), int(rows[i]["mixed"]), round(float(t0[i]), 1), round(float(t1[i]), 1)) for i in late])
