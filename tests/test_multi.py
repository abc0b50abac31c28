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


TRANSPORT = {"peer": "MULTI_PEER", "loopback": "MULTI_RCCL_LOOPBACK", "store": "MULTI_STORE", "rccl": "MULTI_RCCL"}


def tflag(capi, name):
    return getattr(capi, TRANSPORT[name])


@pytest.mark.parametrize("bands,ghost,transport", [(2, 32, "peer"), (3, 16, "peer"), (4, 8, "peer"), (2, 32, "loopback"), (3, 24, "loopback"),
                                                   (2, 32, "store"), (3, 16, "store"), (4, 8, "store")])
def test_multi_jacobi_is_bitwise(dev, bands, ghost, transport):
    capi, ctx = dev
    from demiurge_amd import synth
    W, H, iters = 1024, 512, 200 + 5
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    ref = single_jacobi(capi, ctx, grid, img, b, iters)
    flags = tflag(capi, transport)
    for order in (capi.MULTI_EXCHANGE_THEN_SWEEP, capi.MULTI_OVERLAP_LAST_LAUNCH):
        m = capi.Multi(grid, devices=[0] * bands, ghost=ghost, flags=flags | order)
        m.set_terrain(img); m.set_divw(b)
        m.jacobi(iters, zero_first=True)
        got = m.get(want=("pressure",))["pressure"]
        st = m.stats()
        m.close()
        assert np.array_equal(got, ref), (bands, ghost, transport, order)
        assert st["bands_total"] == bands and st["exchanges"] >= iters // ghost and st["exchange_bytes"] > 0
        assert st["order"] == order and st["warnings"] == 0          # (conftest asks for eight hardware queues: no fall-back)


@pytest.mark.parametrize("bands,transport", [(2, "peer"), (3, "loopback"), (3, "store")])
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
    flags = tflag(capi, transport)
    m = capi.Multi(grid, devices=[0] * bands, params=prm, ghost=32, flags=flags)
    m.set_terrain(img)
    m.step(2)
    got = m.get()
    m.close()
    for k in ("v", "pressure", "divw"):
        assert np.array_equal(got[k], ref[k]), k


def test_multi_reach_failure_is_collective_and_harmless(dev):
    """a back-trace longer than the rows a band OWNS (a halo must come from one neighbour): every band stops BEFORE advect
    (all-reduced reach), v untouched"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 512, 256
    grid = capi.make_grid(W, H)
    img = synth.terrain(W, H)
    m = capi.Multi(grid, devices=[0] * 8, ghost=4, flags=capi.MULTI_PEER)        # 32 rows per band
    m.set_terrain(img)
    v = np.zeros((H, W, 2), np.float32)
    v[100:110, :, 1] = 150.0                      # 150 km/h * 24 h = 3600 km = ~44 rows of 82 km, only one band sees it
    m.set_velocity(v)
    with pytest.raises(capi.DmgError) as e:
        m.step(1)
    assert "owns only" in str(e.value)
    got = m.get(want=("v",))["v"]
    assert np.array_equal(got, v)                 # nothing moved on any band
    m.close()


@pytest.mark.parametrize("bands,ghost,transport", [(2, 4, "peer"), (3, 8, "store"), (4, 4, "loopback")])
def test_multi_step_with_a_reach_beyond_the_ghost_rows(dev, bands, ghost, transport):
    """VERDICT r3 item 3 (OceanCurrents.cpp:294-299: the back-trace travels |v| * 24 h): a velocity whose reach exceeds the
    bands' ghost rows no longer stops the step -- advect gathers from a copy of v that is `reach` rows deep, exchanged like any
    other field (through every transport; the store transport sends it in slot-sized pieces) -- and the bands still
    reproduce one band bit for bit, step after step as the currents change the reach"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 512, 256
    grid = capi.make_grid(W, H)
    img = synth.terrain(W, H)
    prm = capi.default_ocean_params()
    prm.jacobi_iters = 96
    v0 = synth.velocity(W, H) * 40.0                   # |v| up to ~60 km/h: ~17 rows of 82 km, far beyond the ghost rows
    v0[img > 0] = 0
    oc = capi.Ocean(ctx, grid, params=prm)
    oc.set_terrain(img); oc.set_velocity(v0)
    reach = oc.advect_reach()
    oc.step(3)
    ref = oc.get()
    oc.close()
    assert reach > ghost + 4, reach
    m = capi.Multi(grid, devices=[0] * bands, params=prm, ghost=ghost, flags=tflag(capi, transport))
    m.set_terrain(img); m.set_velocity(v0)
    m.step(3)
    got = m.get()
    m.close()
    for k in ("v", "pressure", "divw"):
        assert np.array_equal(got[k], ref[k]), k


def test_config4_whole_outer_step_eight_bands_at_size(dev):
    """BASELINE config 4 -- the 16384x8192 ocean solve split over 8 latitude bands -- as a WHOLE outer step (advect, 50
    diffusion sweeps, divergence, Jacobi, subDiv: OceanCurrents.cpp:155-170), with currents of |v| ~ 10 whose back-trace
    reaches ~96 rows where the bands keep 32 ghost rows: on the one GPU of the test box, bit-identical to one band.
    (The Jacobi solve runs 160 of its 5000 sweeps here: five exchange blocks; the full count is the bench's business.)"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 16384, 8192
    grid = capi.make_grid(W, H)
    img = synth.terrain(W, H)
    prm = capi.default_ocean_params()
    prm.jacobi_iters = 160
    v0 = (synth.velocity(W, H) * 7.0).astype(np.float32)     # |v| up to ~10 km/h: 10 * 24 h / 2.56 km per row
    v0[img > 0] = 0
    oc = capi.Ocean(ctx, grid, params=prm)
    oc.set_terrain(img); oc.set_velocity(v0)
    reach = oc.advect_reach()
    oc.step(2)
    ref = oc.get(want=("v", "pressure"))
    oc.close()
    ctx.trim()
    assert 80 <= reach <= 130, reach
    m = capi.Multi(grid, devices=[0] * 8, params=prm, ghost=32, flags=capi.MULTI_STORE)
    m.set_terrain(img); m.set_velocity(v0)
    m.step(2)
    got = m.get(want=("v", "pressure"))
    m.close()
    for k in ("v", "pressure"):
        assert np.array_equal(got[k], ref[k]), k


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


@pytest.mark.parametrize("transport", ["peer", "loopback", "store"])
def test_multi_overlap_uses_a_second_stream_and_is_bitwise_at_size(dev, transport):
    """VERDICT r2 item 1 / ADVICE r2: the overlapped order puts the exchange on the band's second stream between the two
    halves of the block's last launch.  16384-wide bands (above 4096 tiles: the streaming kernel with per-geometry unit
    heights, the halves of a launch sharing one cut), against one band, bit for bit."""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H, iters = 16384, 2048, 136
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    ref = single_jacobi(capi, ctx, grid, img, b, iters)
    flags = tflag(capi, transport)
    for bands, ghost in ((2, 32), (4, 8)):
        m = capi.Multi(grid, devices=[0] * bands, ghost=ghost, flags=flags | capi.MULTI_OVERLAP_LAST_LAUNCH)
        m.set_terrain(img); m.set_divw(b)
        m.jacobi(iters, zero_first=True)
        got = m.get(want=("pressure",))["pressure"]
        m.close()
        assert np.array_equal(got, ref), (transport, bands, ghost)


def _multi_child(code, env_extra, timeout=240):
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    env.update(env_extra)
    env = {k: v for k, v in env.items() if v is not None}
    return subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=timeout)


_FAIL_CHILD = """
import sys, time
import numpy as np
sys.path.insert(0, '.')
from demiurge_amd import capi, synth
W, H = 1024, 512
grid = capi.make_grid(W, H)
flags = {flags}
m = capi.Multi(grid, devices=[0] * 3, ghost=16, flags=flags | {order})
m.set_terrain(synth.terrain(W, H)); m.set_divw(synth.divw(W, H))
t0 = time.perf_counter()
try:
    m.jacobi(400, zero_first=True)
    m.sync()
    print('NO ERROR')
except capi.DmgError as e:
    print('error after %.2f s: %s' % (time.perf_counter() - t0, e))
try:
    m.jacobi(8)
    print('STILL ALIVE')
except capi.DmgError as e:
    print('second call refused:', e)
m.close()
print('closed')
"""


@pytest.mark.parametrize("transport", ["peer", "loopback", "store"])
@pytest.mark.parametrize("order", ["plain", "overlap"])
def test_multi_band_failure_returns_everywhere(transport, order):
    """one band gives up between exchanges (DMG_MULTI_FAIL: band 1, in front of its 5th exchange).
    Every band thread returns -- the call fails within seconds instead of hanging --, the communicators are aborted, the
    handle refuses further work and can still be destroyed."""
    code = _FAIL_CHILD.format(flags="capi." + TRANSPORT[transport],
                              order="capi.MULTI_OVERLAP_LAST_LAUNCH" if order == "overlap" else "0")
    # (the store transport's neighbours find out through their landing kernels: those give up after the same deadline)
    r = _multi_child(code, {"DMG_MULTI_FAIL": "1:5", "DMG_MULTI_TIMEOUT_MS": "3000" if transport == "store" else "20000"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "failure injected" in r.stdout and "second call refused" in r.stdout and "closed" in r.stdout, r.stdout[-2000:]


_LINK_CHILD = """
import sys, hashlib
import numpy as np
sys.path.insert(0, '.')
from demiurge_amd import capi, synth
W, H = 16384, 2048
grid = capi.make_grid(W, H)
for flags in (capi.MULTI_RCCL_LOOPBACK, capi.MULTI_PEER, capi.MULTI_STORE):
    for order in (capi.MULTI_EXCHANGE_THEN_SWEEP, capi.MULTI_OVERLAP_LAST_LAUNCH):
        m = capi.Multi(grid, devices=[0] * 2, ghost=32, flags=flags | order)
        m.set_terrain(synth.terrain(W, H)); m.set_divw(synth.divw(W, H))
        m.jacobi(200, zero_first=True)
        print(hashlib.sha256(m.get(want=('pressure',))['pressure'].tobytes()).hexdigest())
        m.close()
"""


def test_multi_slow_link_changes_no_bit():
    """The edge-first overlapped order exists to hide a link slower than a local copy.  DMG_MULTI_FAKE_LINK_US holds every
    exchange's stream for that long (the emulation behind profiles/r03_fake_link.log): whatever the interior units, the edge
    units and the exchange do meanwhile, the result is the one without a link, in both orders and both one-GPU transports."""
    fast = _multi_child(_LINK_CHILD, {})
    slow = _multi_child(_LINK_CHILD, {"DMG_MULTI_FAKE_LINK_US": "150"})
    assert fast.returncode == 0 and slow.returncode == 0, fast.stderr[-1500:] + slow.stderr[-1500:]
    import re
    sha = lambda out: [l for l in out.split() if re.fullmatch(r"[0-9a-f]{64}", l)]        # (RCCL prints a banner of its own)
    a, b = sha(fast.stdout), sha(slow.stdout)
    assert len(a) == 6 and len(set(a)) == 1 and a == b, (a, b)


def test_multi_wait_times_out_instead_of_hanging():
    """the host-side waits of the RCCL transports poll with a deadline: with an absurdly short one (1 ms against a solve
    of several hundred) the wait gives up, aborts the communicator and reports it -- what a rank does whose neighbour has
    died -- and the handle can still be destroyed once the kernels that were queued have run."""
    code = """
import sys
sys.path.insert(0, '.')
from demiurge_amd import capi, synth
W, H = 4096, 2048
grid = capi.make_grid(W, H)
m = capi.Multi(grid, devices=[0], ghost=32, flags=capi.MULTI_RCCL)
m.set_terrain(synth.terrain(W, H)); m.set_divw(synth.divw(W, H))
m.jacobi(20000, zero_first=True)
try:
    m.sync()
    print('NO TIMEOUT')
except capi.DmgError as e:
    print('timed out:', e)
m.close()
print('closed')
"""
    r = _multi_child(code, {"DMG_MULTI_TIMEOUT_MS": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "timed out" in r.stdout and "no progress" in r.stdout and "closed" in r.stdout, r.stdout[-2000:]


def test_multi_wait_is_patient_with_a_long_queue_that_progresses():
    """the deadline of a host-side wait is about PROGRESS, not wall time.  A queue of work far longer than
    DMG_MULTI_TIMEOUT_MS whose exchanges keep completing (dmg_multi_jacobi and dmg_multi_climate_steps are asynchronous; the
    reference's climate runs 500 000 steps) must not be mistaken for a dead neighbour."""
    code = """
import sys, time
sys.path.insert(0, '.')
from demiurge_amd import capi, synth
W, H = 16384, 2048
grid = capi.make_grid(W, H)
m = capi.Multi(grid, devices=[0, 0], ghost=32, flags=capi.MULTI_RCCL_LOOPBACK)
m.set_terrain(synth.terrain(W, H)); m.set_divw(synth.divw(W, H))
m.jacobi(64, zero_first=True); m.sync()
t0 = time.perf_counter()
m.jacobi(60000, zero_first=True)
m.sync()
print('solved in %.0f ms' % ((time.perf_counter() - t0) * 1e3))
m.close()
print('closed')
"""
    r = _multi_child(code, {"DMG_MULTI_TIMEOUT_MS": "60"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "closed" in r.stdout and "solved in" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    ms = float(r.stdout.split("solved in")[1].split()[0])
    assert ms > 300, ms                                       # (several deadlines long: the test means something)


def test_multi_overlap_without_eight_hardware_queues_falls_back_and_says_so():
    """VERDICT r3 weak 8 / ADVICE r3: the library no longer edits its host's environment when it is loaded.  A handle asked for
    the overlapped order in a process whose GPU_MAX_HW_QUEUES is unset (HIP's default of four serialises the two streams)
    runs exchange-then-sweep instead, reports it in its stats and on stderr -- and computes the same bits."""
    code = """
import sys, hashlib
sys.path.insert(0, '.')
from demiurge_amd import capi, synth
W, H = 1024, 512
grid = capi.make_grid(W, H)
m = capi.Multi(grid, devices=[0, 0], ghost=16, flags=capi.MULTI_PEER | capi.MULTI_OVERLAP_LAST_LAUNCH)
m.set_terrain(synth.terrain(W, H)); m.set_divw(synth.divw(W, H))
m.jacobi(100, zero_first=True)
st = m.stats()
print('order', st['order'], 'warnings', st['warnings'])
m.close()
"""
    r = _multi_child(code, {"GPU_MAX_HW_QUEUES": None})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "order 0 warnings 1" in r.stdout and "GPU_MAX_HW_QUEUES" in r.stderr, r.stdout[-1000:] + r.stderr[-1000:]
    r = _multi_child(code, {})
    assert "order 16 warnings 0" in r.stdout, r.stdout[-1000:] + r.stderr[-1000:]


@pytest.mark.parametrize("world,order", [(2, "plain"), (3, "overlap")])
def test_store_transport_between_processes(dev, tmp_path, world, order):
    """VERDICT r3 item 1 (peer-mapped direct stores, hipIpc handles per rank): one PROCESS per band, all on the one
    GPU of the test box.  Every rank maps its neighbours' landing slots and flag words through hipIpc handles and the halo rows
    cross the process boundary by direct stores and flags -- the path a rank-per-GPU job takes over xGMI.  The handles travel
    through files here (dmg_multi_store_blob / _connect: a host with its own bootstrap); with an RCCL id they travel through
    one ncclAllGather inside dmg_multi_create_rank."""
    import os, subprocess, sys
    capi, ctx = dev
    from demiurge_amd import synth
    W, H, iters, ghost = 2048, 768, 168, 16
    ref = single_jacobi(capi, ctx, capi.make_grid(W, H), synth.terrain(W, H), synth.divw(W, H), iters)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", HSA_ENABLE_IPC_MODE_LEGACY="0", DMG_MULTI_TIMEOUT_MS="30000")
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "store_worker.py"), str(r), str(world), str(tmp_path), str(W), str(H),
                               str(iters), str(ghost), order], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, o[-1500:] + e[-1500:]
    got = np.zeros((H, W), np.float32)
    for r in range(world):
        part = np.load(tmp_path / ("p%d.npy" % r))
        y0, y1 = r * H // world, (r + 1) * H // world
        got[y0:y1] = part[y0:y1]
    assert np.array_equal(got, ref)
    assert all(("order %d" % (16 if order == "overlap" else 0)) in o for o, _ in outs), outs


def test_multi_rccl_two_devices_smoke(dev):
    """the first box with two GPUs runs the real transport: two bands on two devices, grouped ncclSend/ncclRecv over xGMI,
    both orders, against one band"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU: RCCL refuses two ranks on one device (the loopback and peer transports cover the code paths)")
    capi, ctx = dev
    from demiurge_amd import synth
    W, H, iters = 2048, 1024, 200
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    ref = single_jacobi(capi, ctx, grid, img, b, iters)
    for order in (capi.MULTI_EXCHANGE_THEN_SWEEP, capi.MULTI_OVERLAP_LAST_LAUNCH):
        m = capi.Multi(grid, devices=[0, 1], ghost=32, flags=capi.MULTI_RCCL | order)
        m.set_terrain(img); m.set_divw(b)
        m.jacobi(iters, zero_first=True)
        got = m.get(want=("pressure",))["pressure"]
        m.close()
        assert np.array_equal(got, ref), order


@pytest.mark.parametrize("bands,ghost,transport", [(3, 8, "loopback"), (2, 5, "peer"), (4, 16, "loopback"), (1, 4, "rccl"), (3, 6, "store")])
def test_multi_climate_year_cycle_on_a_fake_cluster(dev, bands, ghost, transport):
    """the climate year cycle (Temperature.cpp:47-162) over latitude bands driven by the library -- the
    bands are dmg_climate_create_band handles, the exchange is dmg_multi's band_exchange (RCCL calls through the loopback
    communicator, peer copies, a one-band RCCL communicator) -- against one band, bit for bit, over several exchanges and a
    terrain change in between"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H, steps = 512, 256, 45
    grid = capi.make_grid(W, H)
    terrain = (synth.terrain(W, H) * 100).astype(np.float32)
    T0 = (15 - 40 * np.abs(np.linspace(-1, 1, H, dtype=np.float32))[:, None] + 3 * synth.divw(W, H)).astype(np.float32)
    flipped = np.ascontiguousarray(-terrain)                  # (a named array: the binding hands out a bare pointer)
    k = capi.Climate(ctx, grid, T0, terrain)
    k.steps(0, steps)
    from demiurge_amd.capi import lib, check, _p
    check(lib().dmg_climate_set_terrain(k.h, _p(flipped)))
    k.steps(steps, 7)
    ref = k.get()
    k.close()
    flags = tflag(capi, transport)
    m = capi.MultiClimate(grid, T0, terrain, devices=[0] * bands, ghost=ghost, flags=flags)
    m.steps(0, steps)
    m.set_terrain(flipped)
    m.steps(steps, 7)
    got = m.get()
    m.close()
    assert np.array_equal(got, ref)
