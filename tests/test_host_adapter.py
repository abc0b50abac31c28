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
    assert "dmg_multi_step" in used and "dmg_multi_get" in used and "dmg_create" in used
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


def test_async_filter_cancel_semantics(tmp_path):
    """Filter.h's cooperative cancel (the reference's Cancel leaves the worker running, Filter.cpp:105-108) on the CPU: g++ builds
    tests/host_cancel_check.cpp against the mirror header alone"""
    exe = tmp_path / "cancel_check"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-I", str(HOST), str(ROOT / "tests" / "host_cancel_check.cpp"), "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "ok" in r.stdout and "destroyed while parked" in r.stdout, r.stdout + r.stderr
    # the ProgressFilter / BackupFilter mirror (reference Filter.cpp:12-23,70-115) -- Cancel restores the backup
    assert "progress filter: cancelled" in r.stdout and "terrain restored" in r.stdout, r.stdout


@pytest.mark.gpu
def test_adapter_cancel_ends_the_solve_at_the_next_outer_step(tmp_path):
    """HipOceanCurrents asked for the reference's 10000 outer steps (OceanCurrents.cpp:155) and cancelled once the first has been
    shown: it finishes a step or two later; one destroyed in mid-run joins at once"""
    import re
    from demiurge_amd import synth
    build()
    W, H = 512, 256
    t = tmp_path / "t.f32"
    synth.terrain(W, H).tofile(t)
    r = subprocess.run([str(HOST / "host_demo"), "cancel", str(W), str(H), str(t)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"cancelled at outer step (\d+) of 10000 .* finished at outer step (\d+)", r.stdout)
    assert m and 1 <= int(m.group(1)) <= int(m.group(2)) <= int(m.group(1)) + 3, r.stdout
    assert "destroyed in mid-run" in r.stdout


@pytest.mark.gpu
def test_progress_filter_cancel_restores_the_backup(tmp_path):
    """HipErosion inside the ProgressFilter / BackupFilter mirror, as cpufilterMenu::makeFilter builds it
    (cpufilter.cpp:30-31): cancelled once its first round's terrain has arrived, the terrain equals the backup taken when the filter
    was made and the filter is finalized; run to the end with half the canvas selected, restoreUnselected keeps the other half"""
    from demiurge_amd import synth
    build()
    W, H = 256, 128
    t = tmp_path / "t.f32"
    (synth.terrain(W, H) * 1000.0).astype(np.float32).tofile(t)
    r = subprocess.run([str(HOST / "host_demo"), "progress", str(W), str(H), str(t)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "the terrain equals the backup" in r.stdout and "unselected ones hold the backup" in r.stdout, r.stdout


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
    # the same filter split over two bands (HipOceanCurrents devices = {0, 0}: dmg_multi underneath): the same bits
    r = subprocess.run([str(HOST / "host_demo"), str(W), str(H), str(t), str(int(G["n_outer"])), str(int(G["jacobi_iters"])),
                        str(int(G["diffusion_iters"])), str(tmp_path / "out2"), "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    vx2 = np.fromfile(tmp_path / "out2_vx.f32", np.float32).reshape(H, W)
    vy2 = np.fromfile(tmp_path / "out2_vy.f32", np.float32).reshape(H, W)
    assert np.array_equal(vx2, vx) and np.array_equal(vy2, vy)


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


@pytest.mark.gpu
def test_remaining_tool_adapters(tmp_path):
    """HipErosion, HipTemperature, HipFlow (AsyncSubFilters) and the one-call tools (HipBlur, HipMorphological*,
    HipThermalErosion) driven frame by frame like the editor, against the same work through the Python binding"""
    build()
    from demiurge_amd import capi, synth
    W, H = 128, 64
    ctx = capi.Context(0)
    g = capi.make_grid(W, H)
    h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 99).reshape(H, W) * 8).astype(np.float32)
    t = tmp_path / "terrain.f32"
    h.tofile(t)

    def demo(*args):
        r = subprocess.run([str(HOST / "host_demo")] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        return np.fromfile(str(args[-1]) + "_terrain.f32", np.float32).reshape(H, W)

    # cpufilter: 3 rounds, lakes off / on
    assert np.array_equal(demo("erosion", W, H, t, 3, 0, tmp_path / "ero"), ctx.erosion_run(g, h, 50, 3, 0.5, 1.0, 1.0))
    e = capi.Erosion(ctx, g, h, 50, 0.5, 1.0, 1.0, lakes=1)
    e.rounds(2)
    assert np.array_equal(demo("erosion", W, H, t, 2, 1, tmp_path / "erol"), e.get())
    e.close()
    # Temperature: 3 frames of 10 steps from T = 50, step indices 0-9, 11-20, 22-31 (the reference's double increment)
    T = np.full((H, W), 50, np.float32)
    for first in (0, 11, 22):
        T = ctx.temperature_steps(g, T, h, first, 10)
    assert np.array_equal(demo("climate", W, H, t, 3, tmp_path / "clim"), T)
    # FlowFilter, lakeflag on
    assert np.array_equal(demo("flow", W, H, t, 1.0, 1, tmp_path / "flow"), ctx.flow_accumulate(g, h, 0.5, 1.0, lakes=1))
    # one-call tools
    assert np.array_equal(demo("blur", W, H, t, 3.0, tmp_path / "blur"), ctx.blur(g, h, 3.0))
    assert np.array_equal(demo("dilate", W, H, t, 5.5, tmp_path / "dil"), ctx.morphological(g, h, 5.5, "max"))
    assert np.array_equal(demo("erode", W, H, t, 5.5, tmp_path / "ero2"), ctx.morphological(g, h, 5.5, "min"))
    assert np.array_equal(demo("morphgrad", W, H, t, 4.0, tmp_path / "mg"), ctx.morphological_gradient(g, h, 4.0))
    assert np.array_equal(demo("thermal", W, H, t, tmp_path / "th"), ctx.thermal_erosion(g, h))
    ctx.close()


@pytest.mark.gpu
def test_handles_outlive_their_context():
    """C side of the round-1 core dump: dmg_destroy(ctx) before its ocean / tectonics / climate / erosion handles.
    The context is reference-counted by its handles; work after dmg_destroy still runs, teardown is clean."""
    build()
    r = subprocess.run([str(HOST / "host_demo"), "lifetime"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "lifetime ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
