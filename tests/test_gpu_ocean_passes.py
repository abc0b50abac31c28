"""Parity of the remaining OceanCurrents passes (advect, diffusion, divergence, subDiv, outer step) through the
C ABI: bit for bit against the oracle and against the llvmpipe goldens (the reference's own GLSL; its transcendentals
are evaluated as Mesa/gallivm evaluate them, csrc/glsl_math.h -- see tests/test_oracle_golden.py)."""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from demiurge_amd import capi
    ctx = capi.Context(0)
    yield capi, ctx
    ctx.close()


def make(capi, ctx, G, W=128, H=64, **prm):
    grid = capi.make_grid(W, H, tuple(G["coords"]), float(G["circ"]))
    params = capi.default_ocean_params()
    for k, v in prm.items():
        setattr(params, k, v)
    o = capi.Ocean(ctx, grid, params=params)
    o.set_terrain(G["img"])
    return grid, o


def ogrid(oracle, grid):
    return oracle.grid(grid.W, grid.H, tuple(grid.coords), grid.circumference)


@pytest.mark.parametrize("name", ["full", "region"])
def test_divergence(dev, oracle, name):
    capi, ctx = dev
    G = load_golden("ocean_passes_128x64_" + name)
    for pf, key in ((100.0, "divergence"), (7.5, "divergence_pf7")):
        grid, o = make(capi, ctx, G, pressurefactor=pf)
        o.set_velocity(G["v"])
        o.divergence()
        d = o.get(want=("divw",))["divw"]
        assert np.array_equal(d, G[key])                                   # the reference's GLSL, bit for bit
        # the oracle evaluates the neighbour's pixel area at offset()'s fp32 coordinate, the kernel at the row centre
        assert np.array_equal(d, oracle.ocean_divergence(ogrid(oracle, grid), G["img"], G["v"], pf))
        o.close()


@pytest.mark.parametrize("name", ["full", "region"])
def test_diffusion_bitwise(dev, oracle, name):
    capi, ctx = dev
    G = load_golden("ocean_passes_128x64_" + name)
    for iters, key in ((1, "diffusion1"), (50, "diffusion50")):
        grid, o = make(capi, ctx, G)
        o.set_velocity(G["v"])
        o.diffusion(iters)
        v = o.get(want=("v",))["v"]
        assert np.array_equal(v, G[key])                                   # the reference's GLSL, bit for bit (50 sweeps)
        assert np.array_equal(v, oracle.ocean_diffusion(ogrid(oracle, grid), G["img"], G["v"], iters))
        o.close()
    # without the reference's scalar-RHS quirk
    grid, o = make(capi, ctx, G, quirks=0)
    o.set_velocity(G["v"])
    o.diffusion(3)
    assert np.array_equal(o.get(want=("v",))["v"], oracle.ocean_diffusion(ogrid(oracle, grid), G["img"], G["v"], 3, scalar_rhs=False))
    o.close()


@pytest.mark.parametrize("name", ["full", "region"])
def test_subdiv(dev, oracle, name):
    capi, ctx = dev
    G = load_golden("ocean_passes_128x64_" + name)
    grid, o = make(capi, ctx, G)
    o.set_velocity(G["v"])
    o.set_pressure(G["pressure200"])
    o.subdiv()
    v = o.get(want=("v",))["v"]
    assert np.array_equal(v, G["subdiv"])                                  # the reference's GLSL, bit for bit
    ref = oracle.ocean_subdiv(ogrid(oracle, grid), G["img"], G["v"], G["pressure200"], 100.0)
    assert np.array_equal(v, ref)
    o.close()


@pytest.mark.parametrize("name", ["ocean_passes_128x64_full", "ocean_passes_128x64_region", "ocean_advect_fast_128x64_full"])
def test_advect(dev, oracle, name):
    capi, ctx = dev
    G = load_golden(name)
    grid, o = make(capi, ctx, G)
    o.set_velocity(G["v"])
    o.advect()
    v = o.get(want=("v",))["v"]
    assert np.array_equal(v, oracle.ocean_advect(ogrid(oracle, grid), G["img"], G["v"]))
    # the reference's own pixels: sin/cos/asin/atan2/pow, the bilinear lerp and the wind constants as Mesa/llvmpipe evaluate them
    assert np.array_equal(v, G["advect"])
    o.close()


def test_outer_step(dev, oracle):
    """OceanCurrents::run body x2 from v = 0: advect -> diffusion(50) -> divergence -> jacobi(150) -> subDiv"""
    capi, ctx = dev
    G = load_golden("ocean_step_128x64_full")
    grid, o = make(capi, ctx, G, jacobi_iters=int(G["jacobi_iters"]), diffusion_iters=int(G["diffusion_iters"]))
    o.step(int(G["n_outer"]))
    r = o.get()
    # the reference's three textures after two outer steps, bit for bit
    assert np.array_equal(r["v"], G["v"]) and np.array_equal(r["pressure"], G["pressure"]) and np.array_equal(r["divw"], G["divw"])
    v0 = np.zeros(G["v"].shape, np.float32)
    v, p, d = oracle.ocean_step(ogrid(oracle, grid), G["img"], v0, int(G["n_outer"]), int(G["jacobi_iters"]),
                                int(G["diffusion_iters"]), rows=True)
    assert np.array_equal(r["v"], v) and np.array_equal(r["pressure"], p) and np.array_equal(r["divw"], d)
    o.close()


def test_outer_step_config3_size_runs(dev):
    """BASELINE config 3 (8192x4096 ocean solve): one outer step with a short Jacobi; finite, land stays at rest."""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 8192, 4096
    grid = capi.make_grid(W, H)
    params = capi.default_ocean_params()
    params.jacobi_iters = 40
    params.diffusion_iters = 10
    o = capi.Ocean(ctx, grid, params=params)
    img = synth.terrain(W, H)
    o.set_terrain(img)
    o.step(1)
    r = o.get(want=("v", "pressure"))
    assert np.isfinite(r["v"]).all() and np.isfinite(r["pressure"]).all()
    assert not r["v"][img > 0].any() and not r["pressure"][img > 0].any()
    assert np.abs(r["v"]).max() > 0
    o.close()


def test_band_passes_fake_cluster(dev):
    """latitude bands of the full outer step on one GPU (halo slabs moved with dmg_copy_device) == one band, bitwise;
    and advect reports a departure point beyond its ghost rows instead of reading garbage"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H, NB, G = 512, 256, 4, 16
    grid = capi.make_grid(W, H)
    img = synth.terrain(W, H)
    v0 = synth.velocity(W, H)
    v0[img > 0] = 0
    prm = capi.default_ocean_params()
    prm.jacobi_iters, prm.diffusion_iters = 24, 20

    one = capi.Ocean(ctx, grid, params=prm)
    one.set_terrain(img)
    one.set_velocity(v0)
    one.step(1)
    ref = one.get()
    one.close()

    oc = [capi.Ocean(ctx, grid, params=prm, band_rank=r, band_count=NB, ghost=G) for r in range(NB)]
    for o in oc:
        o.set_terrain(img)
        o.set_velocity(v0)

    def exchange(field, rows=G):
        for r in range(NB - 1):
            s_send, s_recv, nb = oc[r].halo(field, capi.SIDE_NORTH, rows)
            n_send, n_recv, nb2 = oc[r + 1].halo(field, capi.SIDE_SOUTH, rows)
            assert nb == nb2 and nb == rows * W * (8 if field == capi.F_V else 4)
            ctx.copy_device(n_recv, s_send, nb)
            ctx.copy_device(s_recv, n_send, nb)

    assert max(o.advect_reach() for o in oc) <= G
    exchange(capi.F_V)
    for o in oc:
        o.advect()
    done = 0
    while done < prm.diffusion_iters:
        n = min(G, prm.diffusion_iters - done)
        exchange(capi.F_V)
        for o in oc:
            o.diffusion(n)
        done += n
    exchange(capi.F_V)
    for o in oc:
        o.divergence()
        o.zero_pressure()
    done = 0
    while done < prm.jacobi_iters:
        n = min(G, prm.jacobi_iters - done)
        exchange(capi.F_PRESSURE)
        for o in oc:
            o.jacobi_sweeps(n)
        done += n
    exchange(capi.F_PRESSURE)
    for o in oc:
        o.subdiv()
    got = {}
    for o in oc:
        o.get(out=got)
    for k in ("vx", "vy", "pressure", "divw"):
        assert np.array_equal(got[k], ref[k]), k

    # a fast current needs more halo than a 2-row ghost: loud error, not silent garbage
    small = capi.Ocean(ctx, grid, params=prm, band_rank=1, band_count=NB, ghost=2)
    small.set_terrain(img)
    small.set_velocity(v0 * 30)
    assert small.advect_reach() > 2
    with pytest.raises(capi.DmgError):
        small.advect()
    small.close()
    for o in oc:
        o.close()


def test_restore_accepts_a_round5_sidecar(dev, tmp_path):
    """round 5 set DMG_QUIRK_EXACT_LLVMPIPE (4) by default and wrote `quirks 5` into the sidecar; round 6 turned
    the bit's sense round (set bits mean tolerance; 4 is ignored) and writes `quirks 1` for the same run.  A stored round-5 sidecar
    restores into today's default handle, one that really describes other parameters is still refused, and a handle created with
    the old default value behaves like the default one."""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 256, 128
    grid = capi.make_grid(W, H)
    img = synth.terrain(W, H)
    prm = capi.default_ocean_params()
    assert prm.quirks == capi.QUIRK_DIFFUSION_SCALAR_RHS
    prm.jacobi_iters, prm.diffusion_iters = 16, 8
    a = capi.Ocean(ctx, grid, params=prm)
    a.set_terrain(img)
    a.step(1)
    a.checkpoint(tmp_path / "ck")
    a.step(1)
    ref = a.get()
    a.close()
    meta = (tmp_path / "ck.meta").read_text()
    assert " quirks 1\n" in meta
    (tmp_path / "ck.meta").write_text(meta.replace(" quirks 1\n", " quirks 5\n"))        # as round 5's default handle wrote it
    b = capi.Ocean(ctx, grid, params=prm)
    b.set_terrain(img)
    b.restore(tmp_path / "ck")
    b.step(1)
    got = b.get()
    b.close()
    for k in ("vx", "vy", "pressure", "divw"):
        assert np.array_equal(got[k], ref[k]), k
    old = capi.default_ocean_params()
    old.jacobi_iters, old.diffusion_iters = 16, 8
    old.quirks = capi.QUIRK_DIFFUSION_SCALAR_RHS | capi.QUIRK_EXACT_LLVMPIPE          # a caller that still passes round 5's default
    c = capi.Ocean(ctx, grid, params=old)
    c.set_terrain(img)
    c.restore(tmp_path / "ck")
    c.step(1)
    got = c.get()
    c.close()
    for k in ("vx", "vy", "pressure", "divw"):
        assert np.array_equal(got[k], ref[k]), k
    (tmp_path / "ck.meta").write_text(meta.replace(" quirks 1\n", " quirks 9\n"))        # another rendering: refused
    d = capi.Ocean(ctx, grid, params=prm)
    d.set_terrain(img)
    with pytest.raises(capi.DmgError):
        d.restore(tmp_path / "ck")
    d.close()


def test_checkpoint_resume(dev, tmp_path):
    """run 2 outer steps in one go == run 1, checkpoint to .npy, restore into a fresh ocean, run 1 more"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 256, 128
    grid = capi.make_grid(W, H)
    img = synth.terrain(W, H)
    prm = capi.default_ocean_params()
    prm.jacobi_iters, prm.diffusion_iters = 40, 10
    a = capi.Ocean(ctx, grid, params=prm)
    a.set_terrain(img)
    a.step(2)
    ref = a.get()
    a.close()
    b = capi.Ocean(ctx, grid, params=prm)
    b.set_terrain(img)
    b.step(1)
    b.checkpoint(tmp_path / "ck")
    mid = b.get()
    b.close()
    v = np.load(tmp_path / "ck.g1.v.npy")                   # plain NumPy files, generation 1
    assert v.shape == (H, W, 2) and v.dtype == np.float32 and np.array_equal(v, mid["v"])
    assert np.array_equal(np.load(tmp_path / "ck.g1.pressure.npy"), mid["pressure"])
    assert "generation 1" in (tmp_path / "ck.meta").read_text()
    c = capi.Ocean(ctx, grid, params=prm)
    c.set_terrain(img)
    c.restore(tmp_path / "ck")
    c.step(1)
    got = c.get()
    c.close()
    for k in ("vx", "vy", "pressure", "divw"):
        assert np.array_equal(got[k], ref[k]), k
    d = capi.Ocean(ctx, capi.make_grid(128, 128), params=prm)
    d.set_terrain(synth.terrain(128, 128))
    with pytest.raises(capi.DmgError):
        d.restore(tmp_path / "ck")                          # shape mismatch is refused
    d.close()
    # a checkpoint never damages the one before it.  A second checkpoint writes generation 2, flips the sidecar,
    # removes generation 1; a crash in the middle of a third (its arrays half written, the sidecar not yet replaced)
    # leaves generation 2 restorable; a restore that fails leaves the handle as it was.
    e = capi.Ocean(ctx, grid, params=prm)
    e.set_terrain(img)
    e.restore(tmp_path / "ck")
    e.step(1)
    e.checkpoint(tmp_path / "ck")
    names = sorted(f.name for f in tmp_path.iterdir() if f.name.startswith("ck."))
    assert names == ["ck.g2.divw.npy", "ck.g2.pressure.npy", "ck.g2.v.npy", "ck.meta"], names
    (tmp_path / "ck.g3.v.npy").write_bytes(b"\x93NUMPY half a file")          # the crash
    (tmp_path / "ck.g3.pressure.npy.tmp").write_bytes(b"junk")
    f2 = capi.Ocean(ctx, grid, params=prm)
    f2.set_terrain(img)
    f2.restore(tmp_path / "ck")
    assert all(np.array_equal(f2.get()[k], ref[k]) for k in ("vx", "vy", "pressure", "divw"))       # generation 2 = two outer steps
    before = f2.get()
    (tmp_path / "ck.g2.divw.npy").write_bytes((tmp_path / "ck.g2.divw.npy").read_bytes()[:1000])     # truncated array
    with pytest.raises(capi.DmgError):
        f2.restore(tmp_path / "ck")
    after = f2.get()
    assert all(np.array_equal(before[k], after[k]) for k in before)                                  # untouched
    e.close(); f2.close()


@pytest.mark.gpu
def test_outer_step_4096x2048_bitwise_vs_oracle(dev, oracle):
    """a whole outer step (advect, 50 diffusion sweeps, divergence, 200 Jacobi sweeps, subDiv) at 8.4 M cells against
    the oracle: every field bit for bit (the oracle takes ~20 s of host time)"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 4096, 2048
    grid = capi.make_grid(W, H)
    prm = capi.default_ocean_params()
    prm.jacobi_iters, prm.diffusion_iters = 200, 50
    img = synth.terrain(W, H)
    v0 = synth.velocity(W, H)
    v0[img > 0] = 0
    o = capi.Ocean(ctx, grid, params=prm)
    o.set_terrain(img)
    o.set_velocity(v0)
    o.step(1)
    r = o.get()
    o.close()
    v, p, d = oracle.ocean_step(oracle.grid(W, H), img, v0, 1, 200, 50, rows=True)
    assert np.array_equal(r["divw"], d) and np.array_equal(r["pressure"], p) and np.array_equal(r["v"], v)


@pytest.mark.parametrize("name", ["full", "belt"])
def test_blocked_kernels_vs_reference_hashes(dev, name):
    """the temporally blocked diffusion and Jacobi kernels (W = 512 takes them) against the reference's own pixels:
    SHA-256 of the llvmpipe output of 50 diffusion sweeps and of two outer steps (tests/golden/make_goldens.py blocked)"""
    import hashlib
    from demiurge_amd import synth
    capi, ctx = dev
    G = load_golden("blocked_512x256_" + name)
    W, H = 512, 256
    grid = capi.make_grid(W, H, tuple(G["coords"]), float(G["circ"]))
    img = synth.terrain(W, H)
    v = (synth.velocity(W, H) * np.float32(G["velocity_scale"])).astype(np.float32)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).hexdigest()
    prm = capi.default_ocean_params()
    prm.jacobi_iters, prm.diffusion_iters = int(G["jacobi_iters"]), int(G["diffusion_iters"])
    o = capi.Ocean(ctx, grid, params=prm)
    o.set_terrain(img); o.set_velocity(v); o.diffusion(50)
    d = o.get(want=("v",))["v"]
    assert np.array_equal(d[::8, ::8], G["diffusion50_sample"]) and sha(d) == str(G["diffusion50_sha256"])
    o.set_velocity(np.zeros_like(v))
    o.step(int(G["n_outer"]))
    r = o.get()
    o.close()
    for key, a in (("step_v", r["v"]), ("step_pressure", r["pressure"]), ("step_divw", r["divw"])):
        assert np.array_equal(np.ascontiguousarray(a)[::8, ::8], G[key + "_sample"]), key
        assert sha(a) == str(G[key + "_sha256"]), key


# ---- DMG_QUIRK_FAST_TRANSCENDENTALS: the trig-heavy passes on the hardware's transcendentals (the bit's sense
# inverted in round 6, ADVICE r5: set = tolerance) ----
def _fast(capi):
    return capi.QUIRK_DIFFUSION_SCALAR_RHS | capi.QUIRK_FAST_TRANSCENDENTALS


@pytest.mark.parametrize("name", ["ocean_passes_128x64_full", "ocean_passes_128x64_region", "ocean_advect_fast_128x64_full"])
def test_tolerance_mode_advect_and_divergence_against_the_reference_pixels(dev, name):
    """north_star asks for 1e-5 relative to the reference's GLSL output; the default rendering pays for every bit.  With
    DMG_QUIRK_FAST_TRANSCENDENTALS set advect and divergence keep the reference's expressions (Mesa's asin / atan polynomials included)
    and take sin, cos, log2, exp2, reciprocals and square roots from the hardware: within 1e-5 (max norm, relative to the field's
    largest value) of the llvmpipe goldens -- and not the same bits, or the switch would be doing nothing."""
    capi, ctx = dev
    G = load_golden(name)
    grid, o = make(capi, ctx, G, quirks=_fast(capi))
    o.set_velocity(G["v"])
    o.advect()
    v = o.get(want=("v",))["v"]
    assert relerr(v, G["advect"]) <= 1e-5, relerr(v, G["advect"])
    assert not np.array_equal(v, G["advect"])
    o.close()
    if "divergence" in G:
        for pf, key in ((100.0, "divergence"), (7.5, "divergence_pf7")):
            grid, o = make(capi, ctx, G, pressurefactor=pf, quirks=_fast(capi))
            o.set_velocity(G["v"])
            o.divergence()
            d = o.get(want=("divw",))["divw"]
            assert relerr(d, G[key]) <= 1e-5, (key, relerr(d, G[key]))
            o.close()


def test_tolerance_mode_after_a_whole_outer_step_at_4096x2048(dev):
    """... and after everything an outer step does with those two passes' outputs (50 diffusion sweeps, 200 Jacobi sweeps on the
    divergence, subDiv's coastline snapping on the result) at 8.4 M cells: velocity, pressure and divergence within 1e-5 of the
    exact rendering's.  (subDiv snaps a coastal cell's velocity to a compass direction by comparing angles -- a decision, not a
    value -- so cells whose angle sits within rounding of a threshold may snap differently: they are counted, and must be a
    handful.)"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 4096, 2048
    grid = capi.make_grid(W, H)
    img = synth.terrain(W, H)
    v0 = synth.velocity(W, H)
    v0[img > 0] = 0
    out = []
    for quirks in (None, _fast(capi)):
        prm = capi.default_ocean_params()
        prm.jacobi_iters, prm.diffusion_iters = 200, 50
        if quirks is not None:
            prm.quirks = quirks
        o = capi.Ocean(ctx, grid, params=prm)
        o.set_terrain(img)
        o.set_velocity(v0)
        o.step(1)
        out.append(o.get())
        o.close()
    exact, fast = out
    assert relerr(fast["divw"], exact["divw"]) <= 1e-5 and relerr(fast["pressure"], exact["pressure"]) <= 1e-5
    dv = np.abs(fast["v"].astype(np.float64) - exact["v"]).max(axis=2)
    snapped = dv > 1e-5 * np.abs(exact["v"]).max()
    assert snapped.sum() <= 50, int(snapped.sum())
    assert relerr(fast["v"][~snapped], exact["v"][~snapped]) <= 1e-5
    assert not np.array_equal(fast["v"], exact["v"])
