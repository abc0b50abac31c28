"""The C++ host side above the C ABI: a GL-free mirror of the reference's SubFilter/AsyncSubFilter surface
(demiurge_amd/host/Filter.h) and the HipOceanCurrents adapter a maintainer would drop into the editor."""
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden, relerr

HOST = ROOT / "demiurge_amd" / "host"


def build():
    subprocess.check_call(["make", "-C", str(ROOT / "demiurge_amd" / "csrc")])
    subprocess.check_call(["make", "-C", str(HOST)])


def test_adapter_builds_and_links_only_the_c_abi():
    build()
    out = subprocess.check_output(["nm", "-D", "--undefined-only", str(HOST / "host_demo")], text=True)
    used = sorted({l.split()[-1] for l in out.splitlines() if " dmg_" in l})
    assert "dmg_ocean_step" in used and "dmg_ocean_get" in used and "dmg_create" in used
    assert not [s for s in used if not s.startswith("dmg_")]
    assert "hip" not in out.lower().replace("demiurge_hip", "")       # no HIP/torch types or symbols cross the boundary


def test_adapter_reports_missing_device(tmp_path):
    """no GPU in the build container: the filter finishes with an error instead of falling back"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    build()
    t = tmp_path / "t.f32"
    np.zeros((8, 16), np.float32).tofile(t)
    r = subprocess.run([str(HOST / "host_demo"), "16", "8", str(t), "1", "1", "1", str(tmp_path / "o")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_adapter_matches_golden_and_oracle(oracle, tmp_path):
    """drive HipOceanCurrents like the editor (one step() per frame) and compare scratch1/scratch2 = v.x/v.y"""
    build()
    G = load_golden("ocean_step_128x64_full")
    W, H = 128, 64
    t = tmp_path / "terrain.f32"
    G["img"].astype(np.float32).tofile(t)
    r = subprocess.run([str(HOST / "host_demo"), str(W), str(H), str(t), str(int(G["n_outer"])), str(int(G["jacobi_iters"])),
                        str(int(G["diffusion_iters"])), str(tmp_path / "out")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    vx = np.fromfile(tmp_path / "out_vx.f32", np.float32).reshape(H, W)
    vy = np.fromfile(tmp_path / "out_vy.f32", np.float32).reshape(H, W)
    v = np.stack([vx, vy], -1)
    assert relerr(v, G["v"]) < 1e-6                                    # the reference's GLSL
    vo, _, _ = oracle.ocean_step(oracle.grid(W, H), G["img"], np.zeros_like(v), int(G["n_outer"]), int(G["jacobi_iters"]),
                                 int(G["diffusion_iters"]), rows=True)
    assert np.array_equal(v, vo)


@pytest.mark.gpu
def test_tectonics_and_noise_adapters(oracle, tmp_path):
    """HipTectonics (AsyncSubFilter, one terrain hand-over per step) and HipGradientNoise (one-step SubFilter)"""
    build()
    G = load_golden("tectonics_globe_64x32_s70")
    t = tmp_path / "terrain.f32"
    G["img"].astype(np.float32).tofile(t)
    r = subprocess.run([str(HOST / "host_demo"), "tectonics", "64", "32", str(t), "70", str(tmp_path / "tect")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.fromfile(tmp_path / "tect_terrain.f32", np.float32).reshape(32, 64), G["terrain"])
    r = subprocess.run([str(HOST / "host_demo"), "noise", "128", "64", "4", "3", str(tmp_path / "noise")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ref = oracle.gradient_noise(oracle.grid(128, 64), np.zeros((64, 128), np.float32), "SWISS", seed=3)
    assert np.array_equal(np.fromfile(tmp_path / "noise_terrain.f32", np.float32).reshape(64, 128), ref)
