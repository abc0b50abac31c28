"""The temporally blocked velocity-diffusion kernel (ocean_diffusion_tb.hip: up to 8 sweeps of OceanCurrents.cpp:664-693
per launch) against the oracle's literal one-sweep-at-a-time form, bit for bit: full globe (wrap, both poles with the
sign flip of get_velocity()), a belt (CLAMP_TO_EDGE rows), velocities seeping into land down to 1e-38 (where the
kernel's reciprocal-based division must give way to the IEEE one), every tile height the library builds."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("rows", ["default", "4", "6", "7", "8"])
def test_blocked_diffusion_bitwise_vs_oracle(rows):
    env = dict(os.environ)
    env.pop("DMG_DIFFUSION_SINGLE", None)
    if rows != "default":
        env["DMG_DTB_ROWS"] = rows
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "dtb_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout and "MISMATCH" not in r.stdout


def test_single_sweep_kernel_still_agrees():
    """DMG_DIFFUSION_SINGLE=1 keeps the one-sweep-per-launch kernel reachable (it is also the path of grids the
    blocked kernel does not take: W < 256, W % 8 != 0, no longitude wrap)"""
    env = dict(os.environ, DMG_DIFFUSION_SINGLE="1")
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "dtb_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("sweeps", [1, 5, 8, 13, 50])
def test_blocked_equals_single_sweep_kernel(sweeps):
    """same ocean, same inputs, the two kernels selected through dmg_ocean_params.jacobi_kernel: identical bits"""
    import numpy as np
    from demiurge_amd import capi, synth
    ctx = capi.Context(0)
    W, H = 768, 384
    grid = capi.make_grid(W, H)
    img = synth.terrain(W, H)
    v = (synth.lcg_uniform(W * H * 2, 9).reshape(H, W, 2) * 6).astype(np.float32)
    outs = []
    for kernel in (capi.JACOBI_SINGLE, capi.JACOBI_AUTO):
        prm = capi.default_ocean_params()
        prm.jacobi_kernel = kernel
        o = capi.Ocean(ctx, grid, params=prm)
        o.set_terrain(img); o.set_velocity(v); o.diffusion(sweeps)
        outs.append(o.get(want=("v",))["v"])
        o.close()
    ctx.close()
    assert np.array_equal(outs[0], outs[1])
