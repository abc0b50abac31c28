"""world_size-2/3 latitude-band runs over torch.distributed (gloo) on the CPU: the exchange protocol of
demiurge_amd/bands.py reproduces the single-domain solve bit for bit."""
import socket

import numpy as np
import pytest

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_bands(mode, world, W, H, sweeps, ghost, tmp_path, step=None):
    import torch.multiprocessing as mp
    import sys
    sys.path.insert(0, str(ROOT / "tests"))
    import band_worker
    mp.spawn(band_worker.worker, args=(world, free_port(), mode, W, H, sweeps, ghost, str(tmp_path), step), nprocs=world, join=True)
    return np.concatenate([np.load(tmp_path / ("band%d.npy" % r)) for r in range(world)], axis=0)


@pytest.mark.parametrize("world,ghost,sweeps", [(2, 4, 19), (3, 1, 5), (2, 8, 8)])
def test_bands_gloo_cpu(oracle, tmp_path, world, ghost, sweeps):
    from demiurge_amd import synth
    W, H = 64, 48
    got = run_bands("oracle", world, W, H, sweeps, ghost, tmp_path)
    ref = oracle.ocean_jacobi(oracle.grid(W, H), synth.terrain(W, H), synth.divw(W, H), sweeps, rows=True)
    assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("world,ghost,sweeps,overlap", [(2, 8, 40, "0"), (4, 3, 10, "0"), (2, 8, 100, "1"), (3, 16, 200, "1")])
def test_bands_gloo_hip_ranks_share_gpu(oracle, tmp_path, monkeypatch, world, ghost, sweeps, overlap):
    """N ranks (one process each) drive the HIP library on the same GPU; host-staged gloo halos.  overlap = "1": the
    exchange posted between the edge and the interior tiles of each block's last launch (DMG_HALO_OVERLAP)."""
    from demiurge_amd import synth
    monkeypatch.setenv("DMG_HALO_OVERLAP", overlap)
    W, H = 512, 256
    got = run_bands("hip", world, W, H, sweeps, ghost, tmp_path)
    ref = oracle.ocean_jacobi(oracle.grid(W, H), synth.terrain(W, H), synth.divw(W, H), sweeps, rows=True)
    assert np.array_equal(got, ref)


@pytest.mark.gpu
def test_bands_bounce_buffers(oracle, tmp_path, monkeypatch):
    """the exchanger's fallback transport: slabs copied device-to-device through torch-owned buffers"""
    from demiurge_amd import synth
    monkeypatch.setenv("DMG_HALO_BOUNCE", "1")
    W, H, sweeps, ghost = 512, 256, 40, 8
    got = run_bands("hip", 2, W, H, sweeps, ghost, tmp_path)
    ref = oracle.ocean_jacobi(oracle.grid(W, H), synth.terrain(W, H), synth.divw(W, H), sweeps, rows=True)
    assert np.array_equal(got, ref)


@pytest.mark.gpu
def test_ocean_step_bands_gloo_hip(tmp_path):
    """the whole outer ocean step (advect, 50x diffusion, divergence, Jacobi, subDiv) on 3 bands = one band, bitwise"""
    from demiurge_amd import capi, synth
    W, H, ghost = 512, 192, 16
    step = (2, 40, 20)
    got = run_bands("hip", 3, W, H, 0, ghost, tmp_path, step=step)
    ctx = capi.Context(0)
    prm = capi.default_ocean_params()
    prm.jacobi_iters, prm.diffusion_iters = step[1], step[2]
    o = capi.Ocean(ctx, capi.make_grid(W, H), params=prm)
    img = synth.terrain(W, H)
    v0 = synth.velocity(W, H)
    v0[img > 0] = 0
    o.set_terrain(img)
    o.set_velocity(v0)
    o.step(step[0])
    r = o.get(want=("v", "pressure"))
    ref = np.concatenate([r["v"], r["pressure"][:, :, None]], axis=2)
    o.close()
    ctx.close()
    assert np.array_equal(got, ref)


def _bench_rehearsal(extra_env, extra_args=(), launch="bare", cpu_baseline=False):
    """launch = "bare": `python bench.py --gpus 2 ...`, the launcher starts the ranks itself; "torchrun": bench.py as a rank of a
    torch.distributed.run the caller started; "one-process": the launcher's fall-back child"""
    import json, os, subprocess, sys
    env = dict(os.environ, DMG_BENCH_REHEARSAL="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    tail = ["--gpus", "2", "--steps", "1", "--warmup", "1", "--width", "2048", "--height", "1024", "--sweeps", "96",
            *([] if cpu_baseline else ["--no-cpu-baseline"]), *extra_args]
    if launch == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), str(ROOT / "bench.py"), *tail]
    else:
        cmd = [sys.executable, str(ROOT / "bench.py"), *tail, *(["--one-process"] if launch == "one-process" else [])]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines, (json.loads(lines[0]) if len(lines) == 1 else None)


@pytest.mark.gpu
def test_bench_multi_rank_rehearsal(tmp_path):
    """`python bench.py --gpus 2` exactly as the driver types it: the bare process only launches -- a fresh
    torch.distributed.run child with one process per rank -- and relays rank 0's line.  With the rehearsal switch for a one-GPU
    box: rendezvous, RCCL-id broadcast, barrier, max-over-ranks and the per-band checksums run over gloo, and each rank drives
    both bands on GPU 0 through the library's one-process dmg_multi (RCCL refuses two ranks per GPU).  One timed leg per
    exchange order and transport, each leg's solved pressure checksummed band by band against the first leg's; the headline is
    the fastest, the others are listed; the line carries the band's roofline (PMC child on rank 0's band geometry) and the
    CPU baseline like an N = 1 line."""
    r, lines, d = _bench_rehearsal({}, cpu_baseline=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["roofline"]["sweeps_per_launch"] == 8.0
    assert "2 latitude bands" in d["config"]["parallelism"] and "one process per GPU" in d["config"]["parallelism"]
    assert d["halo_exchange"]["exchanges"] >= 3 and 0 < d["halo_exchange"]["share_of_stream_time"]
    legs = d["legs"]
    assert [(l["transport"], l["order"]) for l in legs] == [("rccl", "exchange-then-sweep"), ("rccl", "overlap"), ("store", "exchange-then-sweep"),
                                                             ("store", "overlap")]
    assert all(l.get("value", 0) > 0 and "error" not in l for l in legs), legs
    assert len({l["checksum"] for l in legs}) == 1 and "?" not in legs[0]["checksum"] and legs[0]["checksum"].count("-") == 1
    assert d["check"]["legs_different"] == [] and len(d["check"]["legs_equal"]) == 4
    assert d["value"] == max(l["value"] for l in legs) and isinstance(d["other_exchange_order"], dict)
    assert d["ranks_seen"]["processes"] == 2
    assert d["roofline"]["frac"] is not None and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["traffic"] > 0, d["roofline"]
    assert "band 0 of 2" in d["roofline"]["band"] and "band 0 of 2" in d["roofline"]["traffic_source"]
    assert d["roofline"]["copy_ceiling"]["value"] > 1000
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1


@pytest.mark.gpu
def test_bench_as_a_rank_of_torchrun_and_as_one_process(tmp_path):
    """the two children the launcher can start, each started directly: bench.py as a rank of torch.distributed.run (how a
    driver that brings its own launcher runs it), and --one-process (one process, a thread and a stream per band)"""
    r, lines, d = _bench_rehearsal({}, ("--no-extra-legs",), launch="torchrun")
    assert r.returncode == 0 and len(lines) == 1, r.stdout + r.stderr[-2000:]
    assert d["n_gpus"] == 2 and d["value"] > 0 and len(d["legs"]) == 1 and "one process per GPU" in d["config"]["parallelism"]
    r, lines, d1 = _bench_rehearsal({}, ("--transport", "store"), launch="one-process")
    assert r.returncode == 0 and len(lines) == 1, r.stdout + r.stderr[-2000:]
    assert d1["n_gpus"] == 2 and d1["value"] > 0 and "ONE process" in d1["config"]["parallelism"] and d1["ranks_seen"]["processes"] == 1
    assert [l["checksum"] for l in d1["legs"]] == [d["legs"][0]["checksum"]] * 2      # the same solve, whoever drives the bands


@pytest.mark.gpu
def test_bench_prints_its_line_when_the_overlapped_legs_fail(tmp_path):
    """the first multi-GPU run must not lose its measurement to a leg that has never run between two
    devices.  DMG_MULTI_FAIL=1:3:16 makes band 1 of every handle in the OVERLAPPED order give up in front of its third
    exchange: those legs fail on every rank (each inside its own try/except, on a handle that is destroyed afterwards), the
    exchange-then-sweep legs before and after them stand, and rank 0 prints its one line with the failures as strings."""
    r, lines, d = _bench_rehearsal({"DMG_MULTI_FAIL": "1:3:16", "DMG_MULTI_TIMEOUT_MS": "3000"})
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    legs = {(l["transport"], l["order"]): l for l in d["legs"]}
    assert len(legs) == 4
    for t in ("rccl", "store"):
        assert legs[(t, "exchange-then-sweep")].get("value", 0) > 0, legs
        assert "error" in legs[(t, "overlap")] and "injected" in legs[(t, "overlap")]["error"] or "failed" in legs[(t, "overlap")]["error"], legs
    assert d["value"] > 0 and "exchange-then-sweep" in d["config"]["parallelism"]
    assert isinstance(d["other_exchange_order"], str) and "failed" in d["other_exchange_order"]


@pytest.mark.gpu
def test_bench_falls_back_to_plain_units_when_every_twin_leg_fails(tmp_path):
    """Round 4: bands run the streaming kernel's twin units, which no run between two devices has seen.  DMG_ST_TWIN_FAIL makes
    every launch that would run them fail: all four legs of the first pass fail on every rank, bench.py switches the twin
    units off (DMG_ST_TWIN=0, read by the library whenever it cuts a launch) and runs its plan once more on fresh handles:
    the line carries the second pass's numbers and says so."""
    r, lines, d = _bench_rehearsal({"DMG_ST_TWIN_FAIL": "1", "DMG_TB_SHAPE": "stream"})
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    assert d["value"] > 0 and "plain units" in d.get("note", ""), d
    assert len(d["legs"]) == 4 and all(l.get("units") == "plain (fall-back)" and l.get("value", 0) > 0 for l in d["legs"]), d["legs"]
    assert "every leg failed" in r.stderr and "injected failure" in r.stderr


@pytest.mark.gpu
def test_bench_prints_a_line_even_when_every_leg_fails(tmp_path):
    """... and when nothing works -- not with plain units either, which the bench tries once every leg has failed -- the line
    still appears (value null, the reasons spelled out) with a non-zero exit status"""
    r, lines, d = _bench_rehearsal({"DMG_MULTI_FAIL": "1:2", "DMG_MULTI_TIMEOUT_MS": "3000"}, ("--transport", "rccl"), launch="torchrun")
    assert r.returncode != 0
    assert len(lines) == 1, r.stdout + r.stderr[-1500:]
    assert d["value"] is None and "no leg completed" in d["error"]
    assert [l.get("units") for l in d["legs"]] == ["twin", "twin", "plain (fall-back)", "plain (fall-back)"], d["legs"]


@pytest.mark.gpu
def test_overlapped_solve_over_rccl():
    """backend nccl (= RCCL) on this box's single GPU: two bands of one process exchange their halos with
    ncclSend/ncclRecv to the process's own rank, so the transfers run on RCCL's own stream next to the interior tiles --
    the stream ordering the gloo tests cannot reach.  All three orders (exchange then sweep, exchange overlapped with the
    last launch's interior, edge chain pipelined on a second stream) must reproduce the single-band solve bit for bit."""
    import os, subprocess, sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "nccl_self_probe.py"), "1024", "512", "200", "16"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if "bit-identical" in l]      # exchange-then-sweep, overlapped last launch, pipelined
    assert len(lines) == 3 and all(l.rstrip().endswith("True") for l in lines), r.stdout
