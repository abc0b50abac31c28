"""One launch's units from a DMG_ST_PROF file, SIMD by SIMD (hardware ids): when the first and the last wave of every SIMD
end -- a launch of one round lasts as long as its slowest SIMD.
usage: python tools/unit_simds.py <file>"""
import csv
import sys

import numpy as np

rows = [r for r in csv.DictReader(open(sys.argv[1])) if int(r["t1"]) > 0]      # (place holders of a twin list record nothing)
t0 = np.array([int(r["t0"]) for r in rows]) / 100.0
t1 = np.array([int(r["t1"]) for r in rows]) / 100.0
base = t0.min()
t0 -= base
t1 -= base
hw = np.array([int(r["hwid"]) for r in rows])
xcc = np.array([int(r["xcc"]) for r in rows]) & 15
# HW_ID (gfx9): wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
h = np.array([int(r["r1"]) - int(r["r0"]) for r in rows])
mixed = np.array([int(r["mixed"]) for r in rows])
q = lambda x: " ".join("%.1f" % v for v in np.percentile(x, [0, 10, 50, 90, 99, 100]))
ends = {}
for k, e in zip(key, t1):
    ends.setdefault(k, []).append(e)
n = np.array([len(v) for v in ends.values()])
first = np.array([min(v) for v in ends.values()])
last = np.array([max(v) for v in ends.values()])
print("units %d on %d SIMDs (units per SIMD: %s), span %.1f us" % (len(rows), len(ends), np.bincount(n).tolist(), t1.max()))
print("start of a unit        (min p10 p50 p90 p99 max):", q(t0))
print("length of a unit       (min p10 p50 p90 p99 max):", q(t1 - t0))
print("a SIMD's first wave ends (min p10 p50 p90 p99 max):", q(first))
print("a SIMD's last wave ends  (min p10 p50 p90 p99 max):", q(last))
for x in np.unique(xcc):
    s = xcc == x
    print("  XCC %d: %d units, coast rows per unit %.1f, end p50 %.1f max %.1f" % (x, s.sum(), mixed[s].mean(), np.median(t1[s]), t1[s].max()))
