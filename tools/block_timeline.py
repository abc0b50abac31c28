"""One block of a banded solve from a rocprofv3 kernel trace: every kernel between two exchanges, relative times, by queue.
usage: python tools/block_timeline.py <dir> [which exchange]"""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 25
ks = sorted(({"n": r["Kernel_Name"], "s": int(r["Start_Timestamp"]) / 1e3, "e": int(r["End_Timestamp"]) / 1e3, "q": int(r["Queue_Id"]),
              "g": int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)} for r in rows), key=lambda k: k["s"])
marks = [i for i, k in enumerate(ks) if "fake_link" in k["n"] or "ncclDevKernel" in k["n"]]
links = [i for i, k in enumerate(ks) if "fake_link" in k["n"]] or marks
a, b = links[which], links[which + 1]
t0 = ks[a]["s"]
for k in ks[a:b + 1]:
    name = "fake_link" if "fake_link" in k["n"] else ("RCCL" if "ncclDev" in k["n"] else k["n"].split("(")[0][:24])
    print("q%-2d %-24s grid %8d  %8.1f .. %8.1f us  (%6.1f)" % (k["q"], name, k["g"], k["s"] - t0, k["e"] - t0, k["e"] - k["s"]))
