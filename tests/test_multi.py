"""dmg_multi: the library's own multi-GPU driver (demiurge_amd/csrc/multi.hip).  On the one-GPU test box the bands share
device 0: peer copies between per-band streams and threads, and the RCCL calls themselves through a loopback
communicator (every slab sent to the process's own rank).  Every split / transport / order must reproduce a single
dmg_ocean bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from demiurge_amd import capi
    ctx = capi.Context(0)
    yield capi, ctx
    ctx.close()


def single_jacobi(capi, ctx, grid, img, b, iters):
    oc = capi.Ocean(ctx, grid)
    oc.set_terrain(img); oc.set_divw(b)
    oc.jacobi(iters, zero_first=True)
    p = oc.get(want=("pressure",))["pressure"]
    oc.close()
    return p


@pytest.mark.parametrize("bands,ghost,transport", [(2, 32, "peer"), (3, 16, "peer"), (4, 8, "peer"), (2, 32, "loopback"), (3, 24, "loopback")])
def test_multi_jacobi_is_bitwise(dev, bands, ghost, transport):
    capi, ctx = dev
    from demiurge_amd import synth
    W, H, iters = 1024, 512, 200 + 5
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    ref = single_jacobi(capi, ctx, grid, img, b, iters)
    flags = capi.MULTI_PEER if transport == "peer" else capi.MULTI_RCCL_LOOPBACK
    for order in (capi.MULTI_EXCHANGE_THEN_SWEEP, capi.MULTI_OVERLAP_LAST_LAUNCH):
        m = capi.Multi(grid, devices=[0] * bands, ghost=ghost, flags=flags | order)
        m.set_terrain(img); m.set_divw(b)
        m.jacobi(iters, zero_first=True)
        got = m.get(want=("pressure",))["pressure"]
        st = m.stats()
        m.close()
        assert np.array_equal(got, ref), (bands, ghost, transport, order)
        assert st["bands_total"] == bands and st["exchanges"] >= iters // ghost and st["exchange_bytes"] > 0


@pytest.mark.parametrize("bands,transport", [(2, "peer"), (3, "loopback")])
def test_multi_step_is_bitwise(dev, bands, transport):
    """the whole outer step (advect -> 50 diffusion -> divergence -> Jacobi -> subDiv) over bands == dmg_ocean_step"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 512, 256
    grid = capi.make_grid(W, H)
    img = synth.terrain(W, H)
    prm = capi.default_ocean_params()
    prm.jacobi_iters = 120
    oc = capi.Ocean(ctx, grid, params=prm)
    oc.set_terrain(img)
    oc.step(2)
    ref = oc.get()
    oc.close()
    flags = capi.MULTI_PEER if transport == "peer" else capi.MULTI_RCCL_LOOPBACK
    m = capi.Multi(grid, devices=[0] * bands, params=prm, ghost=32, flags=flags)
    m.set_terrain(img)
    m.step(2)
    got = m.get()
    m.close()
    for k in ("v", "pressure", "divw"):
        assert np.array_equal(got[k], ref[k]), k


def test_multi_reach_failure_is_collective_and_harmless(dev):
    """a back-trace longer than the ghost slab: every band stops BEFORE advect (all-reduced reach), v untouched"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 512, 256
    grid = capi.make_grid(W, H)
    img = synth.terrain(W, H)
    m = capi.Multi(grid, devices=[0, 0], ghost=4, flags=capi.MULTI_PEER)
    m.set_terrain(img)
    v = np.zeros((H, W, 2), np.float32)
    v[100:110, :, 1] = 60.0                       # 60 km/h * 24 h = 1440 km = ~17 rows of 82 km, only band 0 sees it
    m.set_velocity(v)
    with pytest.raises(capi.DmgError) as e:
        m.step(1)
    assert "ghost" in str(e.value)
    got = m.get(want=("v",))["v"]
    assert np.array_equal(got, v)                 # nothing moved on either band
    m.close()


def test_multi_single_rank_and_rccl_self(dev):
    """dmg_multi_create_rank with one rank (what bench.py uses under torchrun, N = 1) and the RCCL transport with one band"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H, iters = 512, 256, 64
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    ref = single_jacobi(capi, ctx, grid, img, b, iters)
    for m in (capi.Multi(grid, rank=0, world=1, unique_id=None, device=0, ghost=32),
              capi.Multi(grid, devices=[0], ghost=32, flags=capi.MULTI_RCCL)):
        m.set_terrain(img); m.set_divw(b)
        m.jacobi(iters)
        assert np.array_equal(m.get(want=("pressure",))["pressure"], ref)
        m.close()
    assert len(capi.Multi.unique_id()) == 128


def test_multi_full_size_eight_bands(dev):
    """BASELINE config 4's split -- 16384x8192 over 8 latitude bands, 32 ghost rows -- on the one GPU of the test box (peer
    copies between 8 band threads / streams): bit-identical to one band"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H, iters = 16384, 8192, 72
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    ref = single_jacobi(capi, ctx, grid, img, b, iters)
    ctx.trim()
    m = capi.Multi(grid, devices=[0] * 8, ghost=32, flags=capi.MULTI_PEER)
    m.set_terrain(img); m.set_divw(b)
    m.jacobi(iters, zero_first=True)
    got = m.get(want=("pressure",))["pressure"]
    st = m.stats()
    m.close()
    assert np.array_equal(got, ref)
    assert st["exchanges"] == 3 and st["exchange_bytes"] == 3 * 32 * W * 4          # band 0 has one neighbour: 2 MiB per exchange
