#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (separate FETCH_SIZE / WRITE_SIZE passes) into the per-launch
HBM traffic of one kernel, applying the gfx950 corrections of MI355X_MICROARCH.md (HBM section):
FETCH_SIZE is in KB and reports exactly 1/2 of a wide (16 B/lane) coalesced read stream -> doubled;
WRITE_SIZE is in KB and exact for 16 B/lane streaming stores.

usage: pmc_summary.py <kernel-substring> <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]
"""
import csv
import json
import sys


def mean_counter(path, kernel, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    kernel, fetch_csv, write_csv = sys.argv[1:4]
    f, nf = mean_counter(fetch_csv, kernel, "FETCH_SIZE")
    w, nw = mean_counter(write_csv, kernel, "WRITE_SIZE")
    out = dict(kernel=kernel, launches_sampled=dict(fetch=nf, write=nw),
               fetch_size_kb_raw=f, write_size_kb_raw=w,
               fetch_bytes_corrected=None if f is None else 2.0 * f * 1024,
               write_bytes=None if w is None else w * 1024)
    if f is not None and w is not None:
        out["hbm_bytes_per_launch"] = out["fetch_bytes_corrected"] + out["write_bytes"]
    s = json.dumps(out, indent=1)
    print(s)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(s + "\n")


if __name__ == "__main__":
    main()
