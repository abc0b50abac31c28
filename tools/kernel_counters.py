"""What binds each of the tools' kernels at 16384x8192: per kernel, from rocprofv3 passes over
tools/kernel_table_run.py -- average duration (--stats pass), HBM bytes really moved per launch (FETCH_SIZE x 2 + WRITE_SIZE, the
gfx950 correction of MI355X_MICROARCH.md) and their share of the 8 TB/s peak, the bytes the launch cannot avoid, and the SQ view:
vector instructions per cell, the share of a wave's cycles in which it executes a vector instruction, is parked (s_waitcnt) or
stalls at issue.  usage: python tools/kernel_counters.py <dir with stats/ sq1/ sq2/ fetch/ write/ sub-directories>"""
import collections, csv, glob, sys
N = 16384 * 8192
d = sys.argv[1]
kernels = [  # substring, what, unavoidable bytes per cell and launch
    ("k_jacobi_stream<1>", "Jacobi 8 sweeps/launch", 13 * 0.627),
    ("k_jacobi_sweep_v4", "Jacobi 1 sweep/launch", 16),
    ("k_divergence", "divergence", 13),
    ("k_diffusion_tb", "diffusion 7-8 sweeps/launch", 17),
    ("k_advect", "advect", 16),
    ("k_subdiv", "subDiv", 22),
    ("k_temperature_v6", "climate step", 8.25),
    ("k_temperature_edges", "climate step: seam cells", 0.01),
    ("k_temperature_v5", "climate step (round 5, first form)", 8.25),
    ("k_temperature_v4", "climate step (round 4)", 8.25),
    ("k_erosion(", "erosion pass", 16),
    ("k_blur13_axis<0>", "blur13 x", 8),
    ("k_blur13_axis<1>", "blur13 y", 8),
    ("k_flow_direction", "flow direction", 5),
    ("k_flow_jump", "pointer jumping", 16),
    ("k_blur13_axis2<0>", "blur13 x", 8),
    ("k_blur13_axis2<1>", "blur13 y", 8),
    ("k_blur13_ytile", "blur13 y, streamed", 8),
    ("k_flow_passes", "flow: pass candidates", 12),
    # (round 6: the map is W x W texels of 16 B -- two per cell of the W x H canvas: written once = 32 B per cell -- and each of the
    # two plates' 16-byte texels is gathered about once = 32 B; round 4's 16 B per cell made the fold look 4.6 x over its bytes)
    ("k_tect_fold", "tectonics fold", 64),
    # per plate: the old plate read, the new one written, one gather of the map per texel (a plate is W x H, or W x W in one launch of three)
    ("k_tect_unfold", "tectonics unfold", 48),
    ("k_tect_ring_gather", "ocean spreading, one ring pass", 0.5),
]
def pmc(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (d, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
def of(acc, sub, name):
    v = [x for k, c in acc.items() if sub in k for x in c.get(name, [])]
    return sum(v) / len(v) if v else None
stats = {}
for f in glob.glob(d + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        stats[r["Name"]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
sq1, sq2, fe, wr = pmc("sq1"), pmc("sq2"), pmc("fetch"), pmc("write")
print("%-20s %-28s %8s %9s %7s %9s %7s | %9s %6s %6s %6s %6s %6s %5s | %s" % ("kernel", "what", "us", "HBM MB", "of 8TB", "must MB", "of 8TB", "VALU/cell", "valu", "sca", "wait", "stall", "SIMD", "waves", "reading"))
for sub, what, b in kernels:
    us = next((v[0] for k, v in stats.items() if sub in k), None)
    if us is None:
        continue
    f, w = of(fe, sub, "FETCH_SIZE"), of(wr, sub, "WRITE_SIZE")
    hbm = (2.0 * f + w) * 1024 if f is not None and w is not None else None
    wc = of(sq2, sub, "SQ_WAVE_CYCLES")
    g = lambda n: (of(sq2, sub, n) / wc) if wc and of(sq2, sub, n) is not None else float("nan")
    valu_n = of(sq1, sub, "SQ_INSTS_VALU")
    must = b * N
    fr_h = hbm / us / 1e6 / 8.0 if hbm else float("nan")        # bytes per us = 1e6 bytes per s: / 1e6 = TB/s, / 8 = of the peak
    fr_m = must / us / 1e6 / 8.0
    valu, sca, wait, stall = g("SQ_ACTIVE_INST_VALU"), g("SQ_ACTIVE_INST_SCA"), g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY")
    # the chip's view: SQ cycle counters tick once per four clocks; 1024 SIMDs at 2.4 GHz
    av = of(sq2, sub, "SQ_ACTIVE_INST_VALU")
    simd_busy = av * 4.0 / (us * 2400.0 * 1024.0) if av else float("nan")
    waves = wc * 4.0 / (us * 2400.0 * 1024.0) if wc else float("nan")     # resident waves per SIMD, averaged over the launch
    if fr_h >= 0.55: reading = "HBM: at the ceiling of its access shape (a copy reaches 0.79, 1-KB rows 0.69-0.70)"
    elif simd_busy >= 0.55: reading = "vector ALU: the SIMDs execute a vector instruction most of the time"
    elif wait >= 0.5: reading = "latency: waves parked in s_waitcnt on gathers / dependent loads, neither roof near"
    elif stall >= 0.45: reading = "issue stalls: dependent long-latency (transcendental) chains"
    else: reading = "between: ALU %.0f %%, HBM %.0f %% of their roofs" % (100 * simd_busy, 100 * fr_h / 0.7)
    print("%-20s %-28s %8.1f %9.0f %7.2f %9.0f %7.2f | %9.1f %6.2f %6.2f %6.2f %6.2f %6.2f %5.1f | %s" %
          (sub.rstrip("("), what, us, (hbm or 0) / 1e6, fr_h, must / 1e6, fr_m, (valu_n or 0) * 64 / N, valu, sca, wait, stall, simd_busy, waves, reading))
