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


def run_bands(mode, world, W, H, sweeps, ghost, tmp_path):
    import torch.multiprocessing as mp
    import sys
    sys.path.insert(0, str(ROOT / "tests"))
    import band_worker
    mp.spawn(band_worker.worker, args=(world, free_port(), mode, W, H, sweeps, ghost, str(tmp_path)), nprocs=world, join=True)
    return np.concatenate([np.load(tmp_path / ("band%d.npy" % r)) for r in range(world)], axis=0)


@pytest.mark.parametrize("world,ghost,sweeps", [(2, 4, 19), (3, 1, 5), (2, 8, 8)])
def test_bands_gloo_cpu(oracle, tmp_path, world, ghost, sweeps):
    from demiurge_amd import synth
    W, H = 64, 48
    got = run_bands("oracle", world, W, H, sweeps, ghost, tmp_path)
    ref = oracle.ocean_jacobi(oracle.grid(W, H), synth.terrain(W, H), synth.divw(W, H), sweeps, rows=True)
    assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("world,ghost,sweeps", [(2, 8, 40), (4, 3, 10)])
def test_bands_gloo_hip_ranks_share_gpu(oracle, tmp_path, world, ghost, sweeps):
    """N ranks (one process each) drive the HIP library on the same GPU; host-staged gloo halos."""
    from demiurge_amd import synth
    W, H = 512, 256
    got = run_bands("hip", world, W, H, sweeps, ghost, tmp_path)
    ref = oracle.ocean_jacobi(oracle.grid(W, H), synth.terrain(W, H), synth.divw(W, H), sweeps, rows=True)
    assert np.array_equal(got, ref)
