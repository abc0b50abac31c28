"""Per-workgroup timeline of one temporally blocked Jacobi launch (DMG_TB_TRACE).  usage: python tools/tb_trace_probe.py [W H]
Prints phase statistics: load phase, sweep phase, how co-resident workgroups overlap."""
import os
import sys
from pathlib import Path

import numpy as np
import torch  # noqa: F401

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
out = ROOT / "gpurun_out" / "tb_trace.bin"
out.parent.mkdir(exist_ok=True)
os.environ["DMG_TB_TRACE"] = str(out)
from demiurge_amd import capi, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = int(sys.argv[2]) if len(sys.argv) > 2 else W // 2
ctx = capi.Context(0)
oc = capi.Ocean(ctx, capi.make_grid(W, H))
oc.set_terrain(synth.terrain(W, H) if os.environ.get("TERRAIN", "survey") == "survey" else -np.ones((H, W), np.float32))
oc.set_divw(synth.divw(W, H))
oc.jacobi_sweeps(8)
oc.jacobi_sweeps(8)          # the file holds the last launch (warm)
ctx.sync()
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 6)
start, loaded, swept, hw = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2].astype(np.int64), t[:, 3]
live = loaded > 0
t0 = start.min()
print("workgroups %d, swept %d (rest all-land early exit)" % (len(t), live.sum()))
print("launch span %.1f us" % ((np.maximum(swept, start).max() - t0) / 100.0))
ld = (loaded[live] - start[live]) / 100.0
sw = (swept[live] - loaded[live]) / 100.0
print("load phase  us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f" % (ld.mean(), *np.percentile(ld, [10, 50, 90])))
print("sweep phase us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f" % (sw.mean(), *np.percentile(sw, [10, 50, 90])))
clk = (t[live, 5].astype(np.int64) - t[live, 4].astype(np.int64)) / np.maximum(swept[live] - loaded[live], 1) * 100.0 / 1e3
print("shader clock during the sweep phase (s_memtime / s_memrealtime): median %.2f GHz  p10 %.2f  p90 %.2f" % (np.median(clk), *np.percentile(clk, [10, 90])))
# per-CU occupancy over time: identify CU by (xcc, se, sh, cu)
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 0x1; se = (hwid >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("distinct CUs seen:", len(np.unique(key[live])))
# fraction of time both co-resident workgroups are in the same phase
busy_load = busy_sweep = both_load = both_sweep = mixed = 0
for k in np.unique(key[live])[:64]:
    idx = np.where(live & (key == k))[0]
    ev = []
    for i in idx:
        ev += [(start[i], 'L', 1), (loaded[i], 'L', -1), (loaded[i], 'S', 1), (swept[i], 'S', -1)]
    ev.sort()
    nL = nS = 0; last = ev[0][0]
    for tt, ph, d in ev:
        dt = tt - last
        if nL >= 2 and nS == 0: both_load += dt
        elif nS >= 2 and nL == 0: both_sweep += dt
        elif nL >= 1 and nS >= 1: mixed += dt
        elif nL == 1: busy_load += dt
        elif nS == 1: busy_sweep += dt
        last = tt
        if ph == 'L': nL += d
        else: nS += d
tot = both_load + both_sweep + mixed + busy_load + busy_sweep
print("per-CU time: both loading %.0f%%  both sweeping %.0f%%  one loads while one sweeps %.0f%%  single load %.0f%%  single sweep %.0f%%"
      % tuple(100.0 * v / tot for v in (both_load, both_sweep, mixed, busy_load, busy_sweep)))
