"""Parity of the whole-grid filters (blur13, climate step, erosion/uplift pass) through the C ABI."""
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


def grids(capi, oracle, G, shape):
    H, W = shape
    c, circ = tuple(G["coords"]), float(G["circ"])
    return capi.make_grid(W, H, c, circ), oracle.grid(W, H, c, circ)


@pytest.mark.parametrize("name", ["blur_128x64_r1", "blur_128x64_r10", "blur_128x64_region_r3"])
def test_blur(dev, oracle, name):
    """config 1 plumbing: Blur::Blur + Blur::step; all-IEEE arithmetic + host cos table -> bit-exact vs the oracle"""
    capi, ctx = dev
    G = load_golden(name)
    g, og = grids(capi, oracle, G, G["field"].shape)
    out = ctx.blur(g, G["field"], float(G["radius"]))
    assert np.array_equal(out, G["out"])                      # the reference's own pixels (llvmpipe golden)
    assert np.array_equal(out, oracle.blur(og, G["field"], float(G["radius"])))


def test_blur_config1_size(dev, oracle):
    """BASELINE config 1: 1024x512 Gaussian blur"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 1024, 512
    f = synth.terrain(W, H)
    out = ctx.blur(capi.make_grid(W, H), f, 1.0)
    assert np.array_equal(out, oracle.blur(oracle.grid(W, H), f, 1.0))


BLUR_CASES = [  # (W, H, canvas, radius): full globe (pole crossings at the large radii), a belt, a polar cap, a window
    (256, 128, None, 1.0), (256, 128, None, 7.5), (200, 100, None, 40.0), (334, 111, None, 3.0), (128, 64, None, 300.0),
    (256, 128, (-1.0, 1.2, -np.pi, np.pi), 5.0), (192, 96, (0.9, np.pi / 2, -np.pi, np.pi), 12.0), (160, 120, (-0.4, 0.3, 0.2, 1.4), 6.0),
    (2048, 1024, None, 3.0),   # two million cells: from here on the (0, r) pass is the streamed one by default
]


def _blur_cases(capi):
    rng = np.random.default_rng(20261004)
    for W, H, cv, r in BLUR_CASES:
        coords = capi.FULL_GLOBE if cv is None else tuple(np.float32(v) for v in cv)
        yield W, H, coords, r, (rng.standard_normal((H, W)) * 100).astype(np.float32)


def test_blur_axis_kernels_equal_the_literal_kernel_and_the_oracle(dev, oracle):
    """Blur::step's two directions go through the axis forms of blur13 (k_blur13_axis2: what is the same for a row in scalar
    registers; the (0, r) pass since round 6 k_blur13_ytile: a thread walks down a column with the rows' lerps in an LDS ring); DMG_BLUR_LITERAL=1 (read once per process: a child) sends every pass through the general kernel, DMG_BLUR_AXIS_V1=1
    through the axis kernels of round 3.  Same bits, and the oracle's, on random fields."""
    import hashlib
    import os
    import subprocess
    import sys
    from pathlib import Path
    capi, ctx = dev
    mine = []
    for W, H, coords, r, f in _blur_cases(capi):
        out = ctx.blur(capi.make_grid(W, H, coords), f, r)
        assert np.array_equal(out, oracle.blur(oracle.grid(W, H, coords, 42000.0), f, r)), (W, H, coords, r)
        mine.append(hashlib.sha256(out.tobytes()).hexdigest())
    code = ("import sys, hashlib\nsys.path.insert(0, %r); sys.path.insert(0, %r)\nimport numpy as np\nfrom demiurge_amd import capi\n"
            "import test_gpu_filters as t\nctx = capi.Context(0)\n"
            "for W, H, coords, r, f in t._blur_cases(capi):\n    print(hashlib.sha256(ctx.blur(capi.make_grid(W, H, coords), f, r).tobytes()).hexdigest())\n"
            % (str(Path(__file__).resolve().parents[1]), str(Path(__file__).resolve().parent)))
    # the general kernel; the axis kernels before round 5's row-uniform form; round 5's (0, r) pass instead of the one streamed down the columns
    # ... and the streamed pass with blocks of 8 and of 40 rows whatever the radius (every ring size, pole crossings inside the rings' rows)
    for hook, val in (("DMG_BLUR_LITERAL", "1"), ("DMG_BLUR_AXIS_V1", "1"), ("DMG_BLUR_NO_YTILE", "1"), ("DMG_BLUR_YT_ROWS", "8"), ("DMG_BLUR_YT_ROWS", "40")):
        res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{hook: val}), capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        assert res.stdout.split() == mine, hook


def test_blur_streamed_pass_equals_round5_kernel_at_size():
    """The (0, r) pass streamed down the columns (k_blur13_ytile, round 6) against round 5's k_blur13_axis2<1> at 16384x8192 and at config 2's
    4096x2048: the same passes on the same field in two processes (DMG_BLUR_NO_YTILE is read once per process), SHA-256 of what they
    leave -- every ring size (radii up to 5) with the block heights the sizes really get, and a radius beyond the rings (7.5: both take
    the old kernel)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    outs = []
    for extra in ({}, {"DMG_BLUR_NO_YTILE": "1"}):
        res = subprocess.run([sys.executable, str(root / "tools" / "blur_pass_probe.py")], cwd=str(root), env=dict(os.environ, DMG_TEST_HOOKS="1", **extra),
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        shas = [l.split("sha")[-1].strip() for l in res.stdout.splitlines() if " sha " in l]
        assert len(shas) == 2, res.stdout[-2000:]
        outs.append(shas)
    assert outs[0] == outs[1], outs


def test_temperature(dev, oracle):
    capi, ctx = dev
    G = load_golden("temperature_128x64_full")
    g, og = grids(capi, oracle, G, G["T0"].shape)
    for first, n, key in ((0, 1, "T1"), (0, 20, "T20"), (3000, 20, "T20_from3000")):
        out = ctx.temperature_steps(g, G["T0"], G["terrain"], first, n)
        assert np.array_equal(out, G[key])                                           # the reference's GLSL output
        assert np.array_equal(out, oracle.temperature(og, G["T0"], G["terrain"], first, n))


@pytest.mark.parametrize("name,cases", [("temperature_128x64_rough", (("T1_from7", 7, 1), ("T5_from7000", 7000, 5))),
                                        ("temperature_128x64_rough_region", (("T3_from11", 11, 3),))])
def test_temperature_rough_field(dev, oracle, name, cases):
    capi, ctx = dev
    G = load_golden(name)
    g, og = grids(capi, oracle, G, G["T0"].shape)
    for key, first, n in cases:
        out = ctx.temperature_steps(g, G["T0"], G["terrain"], first, n)
        assert np.array_equal(out, G[key]), key


def test_climate_handle_equals_one_shot(dev):
    """dmg_climate (field resident on the device, stepped in pieces as the editor does ten steps per frame) gives the
    field of one dmg_temperature_steps call"""
    capi, ctx = dev
    G = load_golden("temperature_128x64_rough")
    g = capi.make_grid(128, 64, tuple(G["coords"]), float(G["circ"]))
    k = capi.Climate(ctx, g, G["T0"], G["terrain"])
    k.steps(7000, 2)
    k.steps(7002, 3)
    got = k.get()
    k.close()
    assert np.array_equal(got, G["T5_from7000"])
    assert np.array_equal(got, ctx.temperature_steps(g, G["T0"], G["terrain"], 7000, 5))


def _extreme_field(W, H, seed):
    """a climate field that makes waves of the packed kernel take BOTH paths: ordinary temperatures, and patches of values whose
    Laplacian numerators leave the window in which division through the constant is exact -- tiny (2^-90 .. 2^-70), denormal,
    huge (2^70 .. 2^120), infinite, NaN, exactly zero, and -273.15 (OLR numerator zero)"""
    rng = np.random.default_rng(seed)
    T = (rng.standard_normal((H, W)) * 20).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    kinds = [lambda n: (2.0 ** rng.uniform(-90, -70, n) * rng.choice([-1, 1], n)), lambda n: np.full(n, 1e-42), lambda n: 2.0 ** rng.uniform(70, 120, n),
             lambda n: np.full(n, np.inf), lambda n: np.full(n, np.nan), lambda n: np.zeros(n), lambda n: np.full(n, -273.15), lambda n: np.full(n, -np.inf)]
    for i, kind in enumerate(kinds):
        cy, cx = int(rng.integers(4, H - 4)), int(rng.integers(0, W))
        m = (np.abs(yy - cy) <= 2) & (np.abs((xx - cx + W // 2) % W - W // 2) <= 5 + i)
        T[m] = kind(int(m.sum())).astype(np.float32)
    T[H // 2 - 3:H // 2 + 3, :] = 1.5                        # a band that is smooth along the rows: nx == 0 exactly (the fast path's zero)
    return T


@pytest.mark.parametrize("W,H,seed", [(512, 96, 1), (1024, 64, 2), (260, 48, 3)])
def test_temperature_window_edges_take_either_path_to_the_same_bits(dev, oracle, W, H, seed):
    """The climate kernel asks once per wave and row whether every division of its 256 cells may go through the constant; waves that hold
    a value outside the window recompute their cells with the IEEE sequences.  Fields with such patches, against the oracle's
    plain divisions, NaNs and all (W = 260: the one-cell-per-thread kernel's rows of a width that is no multiple of 4 ... 260 is:
    its seam threads then)"""
    capi, ctx = dev
    T = _extreme_field(W, H, seed)
    terr = (np.random.default_rng(seed + 10).standard_normal((H, W)) > 0.3).astype(np.float32) - 0.5
    g, og = capi.make_grid(W, H), oracle.grid(W, H)
    with np.errstate(all="ignore"):
        for first, n in ((0, 1), (5000, 3)):
            got = ctx.temperature_steps(g, T, terr, first, n)
            want = oracle.temperature(og, T, terr, first, n)
            # (bit for bit where the value is a number; a NaN is a NaN -- x86 and the GPU give the default NaN different signs)
            num = ~np.isnan(want)
            assert np.array_equal(np.isnan(got), ~num) and np.array_equal(got.view(np.uint32)[num], want.view(np.uint32)[num]), \
                (first, n, int((got.view(np.uint32)[num] != want.view(np.uint32)[num]).sum()), int((np.isnan(got) != ~num).sum()))
            assert np.isnan(want).any() and np.isinf(want).any()


def test_climate_kernel_forms_agree(dev, oracle):
    """Round 5's climate kernel gives a wave three rows (the rows it reads fetched once), lets a row's first and last wave run like the
    others and computes the cells whose taps cross the seam again in k_temperature_edges.  Every form -- 1, 2, 3 (default), 4 rows per
    wave, with and without the seam cells apart -- must produce the same bits (tests/climate_worker.py: globe and belt canvases, odd
    widths, finite and blown-up fields, three bands), and the default form the oracle's."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    from climate_worker import field
    capi, ctx = dev
    root = Path(__file__).resolve().parents[1]
    forms = [{}, {"DMG_CLIMATE_ROWS": "1", "DMG_CLIMATE_NO_SEAMS": "1"}, {"DMG_CLIMATE_ROWS": "2"}, {"DMG_CLIMATE_ROWS": "4"}, {"DMG_CLIMATE_ROWS": "1"}]
    procs = [subprocess.Popen([sys.executable, str(root / "tests" / "climate_worker.py")], env=dict(os.environ, **f), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for f in forms]
    outs = []
    for f, p in zip(forms, procs):
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, (f, e[-2000:])
        outs.append(o.strip().splitlines())
    assert len(outs[0]) == 9, outs[0]
    for f, o in zip(forms[1:], outs[1:]):
        assert o == outs[0], (f, [(a, b) for a, b in zip(outs[0], o) if a != b])
    # the default form against the oracle where the new paths are all in play (2048 wide: 8 waves a row, seams apart)
    W, H = 2048, 1024
    T, h = field(W, H, W)
    prm = capi.climate_params(diffusivity=0.55e2)
    got = ctx.temperature_steps(capi.make_grid(W, H), T, h, 0, 3, params=prm)
    want = oracle.temperature(oracle.grid(W, H), T, h, 0, 3, params=prm)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int((got != want).sum())


@pytest.mark.parametrize("kw", [
    dict(steps_per_year=730, diffusivity=1.0e4),                                        # a stable coarse year
    dict(steps_per_year=1500000, year_seconds=3.0e7, albedo=0.25, olr_coeff=200.0, olr_ref_kelvin=280.0, emissivity=0.9),
    dict(heat_capacity_land=2.0e7, heat_capacity_ocean=8.0e7, celsius_offset=273.0, eccentricity=0.05, obliquity_deg=10.0, perihelion_deg=120.0,
         solar_constant=1400.0),
    dict(steps_per_year=4096, heat_capacity_land=float(2 ** 24), heat_capacity_ocean=1.0e7 / 3),    # a power-of-two divisor (l == 0), an awkward one
])
def test_temperature_params_follow_the_oracle(dev, oracle, kw):
    """dmg_climate_params: every literal of Temperature.cpp moved, the arithmetic's form kept -- against the oracle's restatement
    with the same values (the defaults are pinned by the llvmpipe goldens above)"""
    capi, ctx = dev
    G = load_golden("temperature_128x64_rough")
    g, og = grids(capi, oracle, G, G["T0"].shape)
    prm = capi.climate_params(**kw)
    out = ctx.temperature_steps(g, G["T0"], G["terrain"], 100, 12, params=prm)
    want = oracle.temperature(og, G["T0"], G["terrain"], 100, 12, params=prm)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), int((out != want).sum())
    assert not np.array_equal(out, ctx.temperature_steps(g, G["T0"], G["terrain"], 100, 12))     # (the values did change something)
    # the same through a resident handle whose parameters change between two step calls
    k = capi.Climate(ctx, g, G["T0"], G["terrain"])
    k.steps(100, 5)
    k.set_params(prm)
    k.steps(105, 7)
    mixed = k.get()
    k.close()
    half = oracle.temperature(og, G["T0"], G["terrain"], 100, 5)
    assert np.array_equal(mixed, oracle.temperature(og, half, G["terrain"], 105, 7, params=prm))


def test_climate_acos_quirk(dev, oracle):
    """h0's acos(-tan(phi)*tan(delta)) (Temperature.cpp:85-89) meets an argument of 1 + 2^-23 on the row where |phi| is within an
    ulp of pi/2 - |delta|: NaN under llvmpipe, and a row of the field is NaN from then on -- at 2048x1024 row 122 between steps 320
    and 328.  Reproduced by default (the oracle has the same NaNs); with DMG_CLIMATE_QUIRK_ACOS_NAN cleared the argument is
    clamped and the field stays finite."""
    from demiurge_amd import synth
    capi, ctx = dev
    W, H = 2048, 1024
    g, og = capi.make_grid(W, H), oracle.grid(W, H)
    terrain = (synth.terrain(W, H) * 100).astype(np.float32)
    T0 = (15 - 40 * np.abs(np.linspace(-1, 1, H, dtype=np.float32))[:, None] + np.zeros((1, W), np.float32)).astype(np.float32)
    stable = dict(diffusivity=0.55e4)
    quirky = ctx.temperature_steps(g, T0, terrain, 318, 12, params=capi.climate_params(**stable))
    rows = np.unique(np.argwhere(np.isnan(quirky))[:, 0])
    assert len(rows) >= 1 and np.isnan(quirky[rows]).all()
    want = oracle.temperature(og, T0, terrain, 318, 12, params=capi.climate_params(**stable))
    assert np.array_equal(quirky, want, equal_nan=True)
    clean_prm = capi.climate_params(quirks=0, **stable)
    clean = ctx.temperature_steps(g, T0, terrain, 318, 12, params=clean_prm)
    assert np.isfinite(clean).all() and np.array_equal(clean, oracle.temperature(og, T0, terrain, 318, 12, params=clean_prm))
    # (rows next to a NaN row lose their Laplacian to the isnan guard -- a step further out per step: beyond that reach nothing changes)
    far = np.ones(H, bool)
    for r in rows:
        far[max(0, r - 13):r + 14] = False
    assert far.sum() > H - 200 and np.array_equal(clean[far], quirky[far])


def test_climate_without_a_field_starts_from_the_initial_temperature(dev, oracle):
    """Temperature::Temperature clears the layer (Temperature.cpp:26-44): a handle created with T == NULL starts at
    params.initial_temperature; bad parameter values are refused"""
    capi, ctx = dev
    G = load_golden("temperature_128x64_full")
    g, og = grids(capi, oracle, G, G["T0"].shape)
    k = capi.Climate(ctx, g, None, G["terrain"])
    k.steps(0, 4)
    assert np.array_equal(k.get(), oracle.temperature(og, np.zeros_like(G["T0"]), G["terrain"], 0, 4))
    k.close()
    prm = capi.climate_params(initial_temperature=12.5, steps_per_year=20000)
    k = capi.Climate(ctx, g, None, G["terrain"], params=prm)
    k.steps(0, 4)
    assert np.array_equal(k.get(), oracle.temperature(og, np.full_like(G["T0"], 12.5), G["terrain"], 0, 4, params=prm))
    for bad in (dict(steps_per_year=0), dict(heat_capacity_land=0.0), dict(heat_capacity_ocean=float("nan")), dict(olr_ref_kelvin=-1.0),
                dict(diffusivity=float("inf"))):
        with pytest.raises(capi.DmgError):
            k.set_params(capi.climate_params(**bad))
    k.steps(4, 1)                                            # a refused set_params leaves the handle as it was
    assert np.array_equal(k.get(), oracle.temperature(og, np.full_like(G["T0"], 12.5), G["terrain"], 0, 5, params=prm))
    k.close()


def test_temperature_tap_table(oracle, tmp_path):
    """the per-row tap displacements (verified against the literal offset() arithmetic at set-up, rows that disagree
    stay literal) give the fields of the all-literal path and of the oracle: the reference's own 1000x500 size,
    a power-of-two globe, a regional extent (clamped columns) and a belt"""
    import os, subprocess, sys
    import temperature_worker
    outs = {}
    for tag, extra in (("table", {}), ("literal", {"DMG_TEMPERATURE_LITERAL": "1"})):
        env = dict(os.environ, **extra)
        env.pop("DMG_TEMPERATURE_LITERAL", None) if not extra else None
        path = tmp_path / (tag + ".npz")
        r = subprocess.run([sys.executable, temperature_worker.__file__, str(path)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[tag] = np.load(path)
    for name, W, H, coords, T, terr in temperature_worker.cases():
        assert np.array_equal(outs["table"][name], outs["literal"][name], equal_nan=True), name
        want = oracle.temperature(oracle.grid(W, H, coords, 42000.0), T, terr, 11, 3)
        assert np.array_equal(outs["table"][name], want, equal_nan=True), (name, int((outs["table"][name] != want).sum()))


def test_erosion(dev, oracle):
    capi, ctx = dev
    G = load_golden("erosion_128x64_full")
    g, og = grids(capi, oracle, G, G["h"].shape)
    for (f, n, key) in ((1.0, 1.0, "out"), (2.0, 0.7, "out_f2_n07")):
        out = ctx.erosion_pass(g, G["h"], G["flow"], G["updrift"], f, n)
        assert np.array_equal(out, G[key])                                           # the reference's GLSL, bit for bit (incl. pow)
        assert np.array_equal(out, oracle.erosion_pass(og, G["h"], G["flow"], G["updrift"], f, n))


def test_erosion_config2_size(dev, oracle):
    """BASELINE config 2 grid (4096x2048): one erosion/uplift pass against the oracle"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 4096, 2048
    h = (synth.terrain(W, H) * 20).astype(np.float32)
    flow = (np.abs(synth.divw(W, H)) * 3).astype(np.float32)
    up = (np.maximum(h, 0) / 50).astype(np.float32)
    out = ctx.erosion_pass(capi.make_grid(W, H), h, flow, up, 1.0, 1.0)
    assert np.array_equal(out, oracle.erosion_pass(oracle.grid(W, H), h, flow, up, 1.0, 1.0))
    assert np.array_equal(out[h <= 0], h[h <= 0])


def test_large_grid_hashes(dev):
    """every tool through the C ABI at 512x256 against the SHA-256 of the reference's own output
    (tests/golden/large_512x256_full.npz, tests/golden/make_goldens.py large)"""
    from large_inputs import NOISE_MODES, check, large_inputs
    capi, ctx = dev
    G = load_golden("large_512x256_full")
    W, H = 512, 256
    g = capi.make_grid(W, H, tuple(G["coords"]), float(G["circ"]))
    I = large_inputs(W, H)
    up = (np.maximum(I["heights"], 0) / np.float32(50)).astype(np.float32)
    check(G, "blur_r3", ctx.blur(g, I["terrain"], 3.0))
    check(G, "temperature_3_from11", ctx.temperature_steps(g, I["T"], I["heights"], 11, 3))
    check(G, "erosion_pass", ctx.erosion_pass(g, I["heights"], I["flow"], up, 1.0, 1.0))
    check(G, "erosion_pass_f2_n07", ctx.erosion_pass(g, I["heights"], I["flow"], up, 2.0, 0.7))
    check(G, "thermal", ctx.thermal_erosion(g, I["relief"]))
    check(G, "erode_r5", ctx.morphological(g, I["terrain"], 5.5, "min"))
    check(G, "dilate_r2", ctx.morphological(g, I["terrain"], 2.0, "max"))
    codes, mask = ctx.flow_passes(g, I["heights"], 0.5)
    check(G, "flow_codes", codes)
    check(G, "flow_mask", mask)
    for m in NOISE_MODES:
        check(G, "noise_" + m, ctx.gradient_noise(g, np.zeros((H, W), np.float32), capi.noise_params(m, seed=5, warp=0.6)))
    t = capi.Tectonics(ctx, g, I["tect"])
    t.step(12)
    check(G, "tectonics_12_terrain", t.terrain())
    t.close()


def test_huge_grid_hashes(dev):
    """the same at 2048x1024 (2 M cells) through the C ABI: advect and a whole ocean outer step with 300 Jacobi sweeps
    (blocked kernels), the climate step, the flow passes, three noise modes"""
    from demiurge_amd import synth
    from large_inputs import check, large_inputs
    capi, ctx = dev
    G = load_golden("huge_2048x1024_full")
    W, H = 2048, 1024
    g = capi.make_grid(W, H, tuple(G["coords"]), float(G["circ"]))
    I = large_inputs(W, H)
    v = (synth.velocity(W, H) * np.float32(3)).astype(np.float32)
    v[I["terrain"] > 0] = 0
    prm = capi.default_ocean_params()
    prm.jacobi_iters, prm.diffusion_iters = 300, 50
    o = capi.Ocean(ctx, g, params=prm)
    o.set_terrain(I["terrain"]); o.set_velocity(v); o.advect()
    check(G, "advect", o.get(want=("v",))["v"], 32)
    o.set_velocity(v); o.step(1)
    r = o.get()
    o.close()
    check(G, "step_v", r["v"], 32); check(G, "step_pressure", r["pressure"], 32); check(G, "step_divw", r["divw"], 32)
    check(G, "temperature_2_from7000", ctx.temperature_steps(g, I["T"], I["heights"], 7000, 2), 32)
    codes, mask = ctx.flow_passes(g, I["heights"], 0.5)
    check(G, "flow_codes", codes, 32); check(G, "flow_mask", mask, 32)
    for m in ("DEFAULT", "SWISS", "Jordan"):
        check(G, "noise_" + m, ctx.gradient_noise(g, np.zeros((H, W), np.float32), capi.noise_params(m, seed=9, warp=0.4)), 32)


def test_large_grid_hashes_regional_extent(dev):
    """the same tools on the regional extent (CLAMP_TO_EDGE everywhere; the ocean passes take the one-sweep kernels)"""
    from demiurge_amd import synth
    from large_inputs import check, large_inputs
    capi, ctx = dev
    G = load_golden("large_512x256_region")
    W, H = 512, 256
    g = capi.make_grid(W, H, tuple(G["coords"]), float(G["circ"]))
    I = large_inputs(W, H)
    v = (synth.velocity(W, H) * np.float32(3)).astype(np.float32)
    v[I["terrain"] > 0] = 0
    prm = capi.default_ocean_params()
    prm.jacobi_iters, prm.diffusion_iters = 100, 50
    o = capi.Ocean(ctx, g, params=prm)
    o.set_terrain(I["terrain"]); o.set_velocity(v); o.step(2)
    r = o.get()
    o.close()
    check(G, "step_v", r["v"]); check(G, "step_pressure", r["pressure"]); check(G, "step_divw", r["divw"])
    check(G, "blur_r3", ctx.blur(g, I["terrain"], 3.0))
    check(G, "temperature_3_from11", ctx.temperature_steps(g, I["T"], I["heights"], 11, 3))
    check(G, "thermal", ctx.thermal_erosion(g, I["relief"]))
    check(G, "erode_r5", ctx.morphological(g, I["terrain"], 5.5, "min"))
    codes, mask = ctx.flow_passes(g, I["heights"], 0.5)
    check(G, "flow_codes", codes); check(G, "flow_mask", mask)
    check(G, "noise_SWISS", ctx.gradient_noise(g, np.zeros((H, W), np.float32), capi.noise_params("SWISS", seed=5, warp=0.6)))
    t = capi.Tectonics(ctx, g, I["tect"])
    t.step(8)
    check(G, "tectonics_8_terrain", t.terrain())
    t.close()
