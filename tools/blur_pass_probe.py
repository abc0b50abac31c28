#!/usr/bin/env python3
"""One blur pass at a time on a device-resident field (dmg_field_blur_pass): ms per (0, r) and per (r, 0) pass for a list of radii at
4096x2048 and 16384x8192, and the SHA-256 of what the passes leave.  DMG_BLUR_NO_YTILE=1: the (0, r) pass by round 5's kernel.
    DMG_TEST_HOOKS=1 python3 tools/blur_pass_probe.py"""
import hashlib, os, sys
import numpy as np
os.environ.setdefault("DMG_TEST_HOOKS", "1")
sys.path.insert(0, ".")
import torch  # noqa: F401
from demiurge_amd import capi, synth
ctx = capi.Context(0)
for W, H in ((4096, 2048), (16384, 8192)):
    g = capi.make_grid(W, H)
    h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
    f = capi.Field(ctx, g, 0, 1, 0, data=h)
    line = []
    for r in (0.05, 0.12, 1.0, 1.94, 3.0, 5.0, 7.5):
        t = {}
        for vertical in (True, False):
            f.blur_pass(r, vertical); ctx.sync()
            ctx.timer_start()
            for _ in range(10):
                f.blur_pass(r, vertical)
            t[vertical] = ctx.timer_stop() / 10
        line.append("r %.2f: (0,r) %.3f (r,0) %.3f" % (r, t[True], t[False]))
    out = np.zeros((H, W), np.float32)
    f.get(out); f.close()
    print("%dx%d ms per pass  %s  sha %s" % (W, H, "  ".join(line), hashlib.sha256(out.tobytes()).hexdigest()[:16]), flush=True)
ctx.close()
