#!/usr/bin/env python3
"""What a rocprofv3 --kernel-trace of tools/overlap_probe.py shows about the halo exchange: how much of the RCCL kernels'
time runs while a Jacobi kernel is running (hidden) and how much with the Jacobi stream idle (exposed).
usage: overlap_trace.py <dir with *kernel_trace.csv> [label]"""
import csv, glob, sys

d = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else d
jac, rccl = [], []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        iv = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
        n = r["Kernel_Name"]
        if "k_jacobi" in n:
            jac.append(iv)
        elif "nccl" in n.lower() or "rccl" in n.lower():
            rccl.append(iv)
jac.sort(); rccl.sort()
# the second solve only (the first one tunes and builds lists): the last half of the Jacobi kernels
if jac:
    t_half = jac[len(jac) // 2][0]
    jac = [iv for iv in jac if iv[0] >= t_half]
    rccl = [iv for iv in rccl if iv[0] >= t_half and iv[1] <= jac[-1][1]]
# union of the Jacobi intervals
merged = []
for a, b in jac:
    if merged and a <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], b)
    else:
        merged.append([a, b])
def covered(a, b):
    c = 0
    for x, y in merged:
        if y <= a: continue
        if x >= b: break
        c += min(b, y) - max(a, x)
    return c
tot = sum(b - a for a, b in rccl)
hid = sum(covered(a, b) for a, b in rccl)
span = (jac[-1][1] - jac[0][0]) if jac else 0
busy = sum(y - x for x, y in merged)
print("%s: %d Jacobi kernels, %d RCCL kernels; span %.2f ms, Jacobi stream busy %.2f ms (idle %.2f ms = %.1f %%); RCCL kernels %.3f ms in all, "
      "%.3f ms (%.0f %%) of it while a Jacobi kernel runs" % (label, len(jac), len(rccl), span / 1e6, busy / 1e6, (span - busy) / 1e6,
      100.0 * (span - busy) / max(span, 1), tot / 1e6, hid / 1e6, 100.0 * hid / max(tot, 1)))
