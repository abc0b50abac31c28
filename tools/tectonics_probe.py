"""Times one Tectonics step (fold / spread / unfold) on the GPU.  usage: python tools/tectonics_probe.py [W H [steps]]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: torch first)

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from demiurge_amd import capi, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else W // 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ctx = capi.Context(0)
img = (synth.terrain(W, H) * 3).astype(np.float32)
t = capi.Tectonics(ctx, capi.make_grid(W, H), img)
t.step(3)
ctx.sync()
acc = dict(fold=0.0, spread=0.0, unfold=0.0)
for _ in range(steps):
    for name in ("fold", "spread", "unfold"):
        ctx.timer_start()
        getattr(t, name)()
        acc[name] += ctx.timer_stop()
tot = sum(acc.values()) / steps
print("tectonics %dx%d (map %dx%d): %.2f ms/step  [fold %.2f  spread(10 passes) %.2f  unfold(2 plates) %.2f]  %.1f Mtexel-steps/s"
      % (W, H, W, W, tot, acc["fold"] / steps, acc["spread"] / steps, acc["unfold"] / steps, W * W / tot / 1e3))
t0 = time.time(); t.step(10); ctx.sync()
print("10 steps wall %.1f ms/step" % ((time.time() - t0) * 100))
