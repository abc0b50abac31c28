#!/usr/bin/env python3
"""Mean SQ counters per launch of one kernel from a rocprofv3 --pmc counter_collection.csv (+ durations from the
kernel trace next to it).  usage: pmc_sq.py <dir> <kernel-substring>"""
import collections
import csv
import glob
import sys

d, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {k: sum(v) / len(v) for k, v in acc.items()}
out["launches"] = len(dur)
out["avg_us"] = sum(dur) / max(len(dur), 1)
wc = out.get("SQ_WAVE_CYCLES")
if wc:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
              "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC"):
        if k in out:
            out[k + "/WAVE_CYCLES"] = round(out[k] / wc, 4)
for k, v in sorted(out.items()):
    print("%-36s %s" % (k, v))
