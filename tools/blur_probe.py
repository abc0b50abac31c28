"""The blur's passes at 16384x8192 (radius 50: the erosion round's pre-blur; six passes per axis) for rocprofv3 --stats, and the
SHA-256 of the result.  DMG_BLUR_AXIS_V1=1: the round-3 axis kernels; DMG_BLUR_LITERAL=1: the general kernel.
usage: rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/blur -o x -- python3 tools/blur_probe.py [W]"""
import hashlib
import sys
import numpy as np
sys.path.insert(0, ".")
from demiurge_amd import capi, synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = W // 2
ctx = capi.Context()
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
out = ctx.blur(g, h, 50.0)
out = ctx.blur(g, h, 50.0)
print("blur %dx%d radius 50: sha %s" % (W, H, hashlib.sha256(out.tobytes()).hexdigest()[:16]))
