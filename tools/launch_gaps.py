"""Gaps between consecutive Jacobi launches of a solve, from a rocprofv3 kernel trace: how much of a small band's time per launch is
not inside a kernel.  usage: python tools/launch_gaps.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
import numpy as np
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_jacobi_stream" in r["Kernel_Name"]))
if len(ks) < 10:
    raise SystemExit("no stream launches in the trace")
s = np.array([k[0] for k in ks], float) / 1e3
e = np.array([k[1] for k in ks], float) / 1e3
dur = e - s
gap = s[1:] - e[:-1]
gap = gap[gap < 200]                      # (between solves: host work)
print("%d launches: duration mean %.1f us (p10 %.1f, p50 %.1f, p90 %.1f); gap to the next launch mean %.2f us (p10 %.2f, p50 %.2f, p90 %.2f); "
      "a launch costs %.1f us of stream time of which %.1f %% is gap" %
      (len(ks), dur.mean(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), gap.mean(), np.percentile(gap, 10), np.median(gap),
       np.percentile(gap, 90), dur.mean() + gap.mean(), 100 * gap.mean() / (dur.mean() + gap.mean())))
