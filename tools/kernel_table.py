"""Achieved bytes per second of the tools' kernels at 16384x8192 against the 8 TB/s HBM peak: the bytes a launch cannot avoid
(per cell: what it must read once and write once) over its average duration under `rocprofv3 --kernel-trace --stats`.
usage: python tools/kernel_table.py <dir with *kernel_stats.csv>"""
import csv, glob, sys
N = 16384 * 8192
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0])))
def avg(sub):
    r = [x for x in rows if sub in x["Name"]]
    return (float(r[0]["AverageNs"]) / 1e3, int(r[0]["Calls"])) if r else (None, 0)
table = [  # kernel substring, what, unavoidable bytes per cell and launch, cells
    ("k_jacobi_stream<1>", "Jacobi, 8 sweeps per launch (water cells: 62.7 %; 13 B each)", 13 * 0.627, N),
    ("k_jacobi_sweep_v4", "Jacobi, one sweep per launch (SURVEY 8d: 16 B)", 16, N),
    ("k_divergence", "divergence of v: v 8 + codes 1 in, divw 4 out", 13, N),
    ("k_diffusion_tb", "velocity diffusion, 7-8 sweeps per launch: v 8 in, 8 out, codes 1", 17, N),
    ("k_advect", "semi-Lagrangian advection: v 8 in (gathered), 8 out", 16, N),
    ("k_subdiv", "v -= grad p: p 4 + v 8 + codes 2 in, v 8 out", 22, N),
    ("k_temperature_v4", "climate step: T 4 in, 4 out, land sign 0.25", 8.25, N),
    ("k_erosion(", "erosion / uplift pass: h 4 + flow 4 + uplift 4 in, h 4 out", 16, N),
    ("k_blur13_axis<0>", "blur13 along x: 4 in, 4 out", 8, N),
    ("k_blur13_axis<1>", "blur13 along y: 4 in, 4 out", 8, N),
    ("k_blur13_ytile", "blur13 along y, streamed down the columns: 4 in, 4 out", 8, N),
    ("k_flow_direction", "flow direction codes: h 4 in, code 1 out", 5, N),
    ("k_flow_jump", "pointer jumping round: parent 4 + dist 4 in (+ gathers), 8 out", 16, N),
    ("k_tect_fold", "tectonics fold: map 16 out (+ plate gathers)", 16, N),
    ("k_tect_unfold", "tectonics back to plates: map 16 in (gathered), plate 16 out", 32, N),
]
print("%-22s %-78s %9s %9s %8s" % ("kernel", "bytes a launch cannot avoid, per cell", "us/launch", "GB/s", "of 8 TB/s"))
for sub, what, b, cells in table:
    us, calls = avg(sub)
    if us is None:
        continue
    gbs = b * cells / us / 1e3
    print("%-22s %-78s %9.1f %9.0f %8.2f" % (sub.rstrip("("), what, us, gbs, gbs / 8000.0))
