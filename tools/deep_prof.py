"""per-wave clocks of one deep launch (DMG_DEEP_PROF=<file>): how long units take, how much of it each wave waits for its partner"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["t0"]) for r in rows)
for w, name in ((0, "front"), (1, "back")):
    rr = [r for r in rows if int(r["wave"]) == w]
    dur = sorted((int(r["t1"]) - int(r["t0"])) / 100.0 for r in rr)
    wait = sorted(int(r["waited"]) / 100.0 for r in rr)
    end = sorted((int(r["t1"]) - t0) / 100.0 for r in rr)
    start = sorted((int(r["t0"]) - t0) / 100.0 for r in rr)
    q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
    print("%-5s waves %d: duration us p10/50/90 %.1f/%.1f/%.1f  waited us p10/50/90 %.1f/%.1f/%.1f (%.0f %% of the median duration)  starts p50 %.1f p99 %.1f  ends p50 %.1f max %.1f" %
          (name, len(rr), q(dur, .1), q(dur, .5), q(dur, .9), q(wait, .1), q(wait, .5), q(wait, .9), 100 * q(wait, .5) / q(dur, .5), q(start, .5), q(start, .99), q(end, .5), end[-1]))
# which SIMD the two waves of a workgroup sit on: HW_ID bits 4-5 SIMD, 8-11 CU
same = sum(1 for i in range(0, len(rows), 2) if ((int(rows[i]["hwid"]) >> 4) & 3) == ((int(rows[i + 1]["hwid"]) >> 4) & 3))
print("workgroups whose two waves share a SIMD: %d of %d" % (same, len(rows) // 2))
