#!/usr/bin/env python3
"""bench.py -- Mcell-steps/s of the ocean-current Jacobi pressure stencil (BASELINE.json metric).

A "step" is one pressure solve of the reference (OceanCurrents::jacobi, src/filter/OceanCurrents.cpp:437-509):
pressure := 0 then `--sweeps` Jacobi sweeps (reference value 5000, the loop of :500-508) over the synthetic 16384x8192
full-globe grid of SURVEY.md section 8d, inputs resident in HBM before the timed region.

    python bench.py [--gpus N --steps K --warmup W]

N > 1 splits the SAME grid into N latitude bands (strong scaling); the bands and their halo exchange are driven by the
library itself (dmg_multi, demiurge_amd/csrc/multi.hip).  How the N bands get their processes:

  * started bare (no WORLD_SIZE in the environment), this process is only a LAUNCHER: it imports neither torch nor the library
    and makes no HIP call.  It starts `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a
    fresh child -- one process per GPU -- relays rank 0's line and exits with the child's code.  If that child prints no line
    with a value, a second fresh child drives all N devices from ONE process (`--one-process`: dmg_multi_create(N, [0..N-1]),
    RCCL through ncclCommInitAll, one host thread and stream per band) and its line is printed with a note saying so.
  * started under torch.distributed.run (WORLD_SIZE = N), it is one rank of that launch: torch.distributed (gloo, CPU tensors)
    only hands out the RCCL id, the barrier, the max-over-ranks and the per-band checksums.

An N > 1 run times one leg (W warm-ups + K solves) per exchange order and transport, each on a fresh handle inside its own
try/except, the form that has run before first (exchange, then sweep, over RCCL).  After each leg every band's solved pressure is
checksummed (CRC-32 of its owned rows); a leg whose checksums differ from the first completed leg's is marked with an `error` and
cannot become the headline.  The headline is the fastest remaining leg that every rank got through; the others are listed under
`legs`, and a watchdog prints the line with what there is if a later leg hangs.

ONE JSON line on rank 0.  Besides the contract's fields:
  roofline          the default kernel (8 sweeps per launch on chip).  `achieved` = HBM bytes the launch really moves
                    (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, collected by a child of this run; for N > 1 on rank 0's band
                    geometry) / its HIP-event duration, `frac` = that / 8 TB/s  (<= 1).  The SURVEY 8d figure -- 16
                    algorithmic bytes per cell-sweep, which a temporally blocked kernel undercuts ~10x -- is kept as
                    `algorithmic_x_peak`; `fp32` is the vector-ALU side (13 flop per executed cell-sweep, water cells only,
                    against the 157.3 TFLOP/s fp32 peak); `copy_ceiling` = a 1 GiB device copy timed in this run.
  roofline_single   a second, short timed leg with one sweep per launch: the HBM-bound form the north_star's ">= 60 % of
                    HBM roofline" is stated for; achieved = 16 B x cells / launch duration.
  all_ocean         the same solve on a terrain without land (nothing to skip: the kernel's worst case), with its fp32 side
  legs              N > 1: every timed leg {transport, order, value | error, halo_exchange, checksum}
  other_exchange_order   N > 1: the headline transport's other order (value, or "... failed: <why>")
  ranks_seen        N > 1: what ncclCommCount reports for the data communicator, and the distinct devices behind the bands
  check             64 sweeps by the default kernel against 64 single-sweep launches, compared bit for bit (untimed)
  cpu_baseline      the reference GLSL on llvmpipe on this host (oracle/_ref), else the C port
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time
import zlib
from pathlib import Path

# dmg_multi's overlapped order runs two streams per band; HIP's default of four hardware queues serialises them on this
# runtime (profiles/r03_overlap_probe.log).  Read when HIP initialises, so set before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# a rank whose neighbour has died notices when its stream stops making progress for this long (multi.hip, multi_wait); a solve
# of the bench takes well under a second per rank
os.environ.setdefault("DMG_MULTI_TIMEOUT_MS", "60000")

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
HBM_ROWS_GBS = 5550.0          # what whole 1-KB rows fetched once from all over a large buffer reach (same guide: 5.5-5.6 TB/s):
                               # the streaming kernel's access shape (a wave reads 1 KB of a row, the next row is 64 KB further on)
FP32_PEAK_TFLOPS = 157.3       # vector fp32 peak (same guide)
FP32_ISSUABLE_TFLOPS = 106.0   # what packed FMAs really issue at two waves per SIMD (tools/pk_probe.hip, profiles/r03_valu_issue_probe.log)
ALGO_BYTES_PER_CELL_STEP = 16  # read p, read divw, read terrain, write p'
FLOP_PER_CELL_STEP = 13        # what the default kernel executes per cell-sweep: 4 add/sub + 3 divisions of (1 mul + 1 fma) each, an fma = 2
                               # (round 2's three-operation division: 19)
JACOBI_KERNELS = ["k_jacobi_stream", "k_jacobi_stream_twin", "k_jacobi_tb8", "k_jacobi_sweep_v4"]
# what torch.distributed.run puts into its workers' environment: a child of a rank (PMC passes, CPU baseline) or of the launcher's
# fall-back must not look like a rank itself
DIST_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
            "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_RUN_ID",
            "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_ERROR_FILE", "TORCH_NCCL_ASYNC_ERROR_HANDLING")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=16384)
    ap.add_argument("--height", type=int, default=8192)
    ap.add_argument("--sweeps", type=int, default=5000, help="Jacobi sweeps per solve (reference: 5000)")
    ap.add_argument("--ghost", type=int, default=32, help="ghost rows per band side = sweeps between halo exchanges")
    ap.add_argument("--kernel", choices=["auto", "single", "blocked"], default="auto",
                    help="Jacobi kernel: one sweep per launch, or 8 sweeps per launch temporally blocked (bit-identical)")
    ap.add_argument("--terrain", choices=["survey", "ocean"], default="survey",
                    help="survey: the ~40%% land field of SURVEY.md 8d (the metric's workload); ocean: no land at all "
                         "(worst case for the kernel, which skips work on land)")
    ap.add_argument("--order", choices=["auto", "exchange-then-sweep", "overlap"], default="auto",
                    help="N>1: refresh the ghost rows, then sweep a block; or send them while the interior of the block's last "
                         "launch is swept (its edge units first, alone).  auto (default): one timed leg each, exchange-then-sweep "
                         "first; the headline is the faster leg that every rank got through")
    ap.add_argument("--transport", choices=["auto", "rccl", "store"], default="auto",
                    help="N>1: ncclSend/ncclRecv of the halo slabs, or the edge rows stored straight into the neighbour's memory "
                         "(peer-mapped / IPC pointers, flags instead of an RCCL kernel per block).  auto: RCCL legs first, then store legs")
    ap.add_argument("--leg-timeout", type=int, default=240,
                    help="N>1: seconds a leg after the first may take before the line is printed without it")
    ap.add_argument("--launch-timeout", type=int, default=320,
                    help="N>1, bare start: seconds the torch.distributed.run child may take before the one-process child is tried "
                         "(which gets two thirds of it: both fit the 600 s a driver allows a bench run)")
    ap.add_argument("--one-process", action="store_true",
                    help="N>1: drive all N devices from this one process (dmg_multi_create) instead of one process per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="headline leg only (no single-sweep / all-ocean / PMC / check)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-bands", type=int, default=1, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def child_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in DIST_ENV}
    env.update(extra)
    return env


# --------------------------------------------------------------------------------------------------
# The launcher: `python bench.py --gpus N` started bare.  Nothing in here may import torch or the library or touch HIP: the
# processes that do are fresh children (tests/test_bench_launcher.py holds this function to that).
# --------------------------------------------------------------------------------------------------
def json_lines(text):
    out = []
    for line in text.splitlines():
        if line.startswith('{"metric"'):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


def run_child(cmd, timeout, env):
    """-> (rc or None on timeout, stdout).  stderr passes through.  A child that outlives `timeout` is ended with its whole
    process group (the ranks are grandchildren), by the group id this launcher created -- never by pattern."""
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=str(ROOT), start_new_session=True)
    try:
        out, _ = p.communicate(timeout=timeout)
        return p.returncode, out
    except subprocess.TimeoutExpired:
        import signal
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(p.pid, sig)
            except ProcessLookupError:
                break
            try:
                out, _ = p.communicate(timeout=15)
                return None, out
            except subprocess.TimeoutExpired:
                continue
        return None, ""


def launcher(args, argv, popen=run_child):
    n = args.gpus
    me = str(ROOT / "bench.py")
    first = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
             "--master-port", str(free_port()), me] + argv
    rc, out = popen(first, args.launch_timeout, child_env())
    lines = json_lines(out or "")
    good = [d for d in lines if d.get("value")]
    if good:
        print(json.dumps(good[-1]), flush=True)
        return rc if rc is not None else 0
    why = ("torch.distributed.run child: %s, %s" %
           ("no exit within %d s" % args.launch_timeout if rc is None else "rc %d" % rc,
            ("its line: " + str(lines[-1].get("error"))[:300]) if lines else "no line"))
    print("bench: %s; once more from ONE process driving %d devices" % (why, n), file=sys.stderr, flush=True)
    t2 = max(60, args.launch_timeout * 2 // 3)
    rc2, out2 = popen([sys.executable, me] + argv + ["--one-process"], t2, child_env())
    lines2 = json_lines(out2 or "")
    good2 = [d for d in lines2 if d.get("value")]
    if good2:
        d = good2[-1]
        d["note"] = ("one process driving %d devices (dmg_multi_create, ncclCommInitAll): the one-process-per-GPU launch gave no "
                     "value (%s). " % (n, why) + d.get("note", "")).strip()
        print(json.dumps(d), flush=True)
        return rc2 if rc2 is not None else 0
    # nothing measured: the line says why, with both children's reasons
    d = (lines or lines2 or [{"metric": "Mcell-steps/s, ocean Jacobi stencil @%dx%d fp32" % (args.width, args.height), "value": None,
                              "unit": "Mcell-steps/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                              "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": "ocean-current Jacobi pressure solve, %dx%d full globe" % (args.width, args.height)}}])[-1]
    d["error"] = ("%s; one-process child: %s, %s" %
                  (why, "no exit within %d s" % t2 if rc2 is None else "rc %d" % rc2,
                   ("its line: " + str(lines2[-1].get("error"))[:300]) if lines2 else "no line"))
    print(json.dumps(d), flush=True)
    return 3


# --------------------------------------------------------------------------------------------------
# CPU baseline (rank 0): runs in a child process so a missing/fragile Mesa cannot take the bench down
# --------------------------------------------------------------------------------------------------
def cpu_baseline_child(args):
    """Times a bounded sample of the same workload on the host cores and prints a JSON object."""
    from demiurge_amd import synth
    W, H = args.width, args.height
    img, b = synth.terrain(W, H), synth.divw(W, H)
    # a one-GPU box grants ~16 cores whatever the host shows; oversubscribed OpenMP/llvmpipe threads crawl
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    res = None
    sys.path.insert(0, str(ROOT / "oracle" / "refgl"))
    sys.path.insert(0, str(ROOT / "oracle"))
    try:
        os.environ.setdefault("LP_NUM_THREADS", str(cores))
        import refgl
        if not refgl.available():
            raise RuntimeError("oracle/_ref/librefgl.so not built")
        t = []
        refgl.ocean_jacobi(img, b, 2, timing=t)                       # includes shader compile + first-touch
        per = max(t[0] / 2, 1e-3)
        n = int(max(3, min(40, 15.0 / per)))
        t = []
        refgl.ocean_jacobi(img, b, n, timing=t)
        res = dict(value=W * H * n / t[0] / 1e6, unit="Mcell-steps/s", cores=int(os.environ["LP_NUM_THREADS"]),
                   kind="reference",
                   sample="%d sweeps of the reference Jacobi GLSL (OceanCurrents.cpp:450-481) at %dx%d on %s, %.1f s"
                          % (n, W, H, refgl.RefGL.get().renderer().split("|")[0].strip(), t[0]))
    except Exception as e:  # Mesa / reference harness unavailable on this host: fall back to the C port
        note = "llvmpipe reference unavailable (%s)" % str(e)[:80]
        import oracle
        g = oracle.grid(W, H)
        t0 = time.time()
        oracle.ocean_jacobi(g, img, b, 2, rows=True, nthreads=cores)
        per = max((time.time() - t0) / 2, 1e-3)
        n = int(max(3, min(200, 15.0 / per)))
        t0 = time.time()
        oracle.ocean_jacobi(g, img, b, n, rows=True, nthreads=cores)
        dt = time.time() - t0
        res = dict(value=W * H * n / dt / 1e6, unit="Mcell-steps/s", cores=cores, kind="port",
                   sample="%d sweeps of oracle/dmg_oracle.c orc_ocean_jacobi_rows (OpenMP) at %dx%d, %.1f s; %s"
                          % (n, W, H, dt, note))
    print("CPU_BASELINE_JSON " + json.dumps(res))


def cpu_baseline(args):
    cmd = [sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-only", "--width", str(args.width),
           "--height", str(args.height)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=child_env())
        for line in out.stdout.splitlines():
            if line.startswith("CPU_BASELINE_JSON "):
                return json.loads(line[len("CPU_BASELINE_JSON "):])
        return dict(value=None, unit="Mcell-steps/s", cores=0, kind="port",
                    sample="cpu baseline failed: " + (out.stderr or out.stdout)[-200:])
    except Exception as e:
        return dict(value=None, unit="Mcell-steps/s", cores=0, kind="port", sample="cpu baseline failed: %s" % e)


# --------------------------------------------------------------------------------------------------
# HBM traffic of the two Jacobi kernels, measured in this run: a child under rocprofv3 --pmc, one counter per pass
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots)
# --------------------------------------------------------------------------------------------------
def pmc_child(args):
    """a few launches of each kernel, nothing else: what the profiler passes run.  --pmc-bands N > 1: the launches of band 0
    of N (rank 0's band geometry: its owned rows plus `ghost` rows towards band 1), blocks of `ghost` sweeps as dmg_multi
    issues them between two halo refreshes -- the ghost rows are simply not refreshed here, which moves no byte differently."""
    import numpy as np
    from demiurge_amd import capi, synth
    W, H, nb = args.width, args.height, args.pmc_bands
    ctx = capi.Context(0)
    grid = capi.make_grid(W, H)
    img = synth.terrain(W, H) if args.terrain == "survey" else -np.ones((H, W), np.float32)
    b = synth.divw(W, H)
    # (a small grid's first twelve launches are the tile-shape tuner's trials: enough launches for the chosen kernel to show)
    small = ((W + 239) // 240) * ((H // nb + 47) // 48) <= 4096
    for kern, sweeps in ((capi.JACOBI_AUTO, 192 if small else 40), (capi.JACOBI_SINGLE, 6)):
        prm = capi.default_ocean_params()
        prm.jacobi_kernel = kern
        if nb == 1:
            oc = capi.Ocean(ctx, grid, params=prm)
            oc.set_terrain(img); oc.set_divw(b)
            oc.jacobi(sweeps, zero_first=True)
        else:
            ghost = max(1, min(args.ghost, H // nb))
            blk = ghost - ghost % 8 if ghost >= 8 else ghost
            oc = capi.Ocean(ctx, grid, params=prm, band_rank=0, band_count=nb, ghost=ghost)
            oc.set_terrain(img); oc.set_divw(b)
            oc.zero_pressure()
            for _ in range(max(1, (sweeps if kern == capi.JACOBI_AUTO else 4 * blk) // blk)):
                oc.jacobi_sweeps(blk)
        ctx.sync()
        oc.close()
    ctx.close()


POWER_HELPER = r"""
import re, subprocess, sys, time
gpu = int(sys.argv[1])
pat_c = re.compile(r"GPU\[%d\]\s*: sclk clock level: \S+ \((\d+)Mhz\)" % gpu)
pat_p = re.compile(r"GPU\[%d\]\s*: Current Socket Graphics Package Power \(W\): ([\d.]+)" % gpu)
while True:
    t = time.time()
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    c, w = pat_c.search(out), pat_p.search(out)
    if c and w:
        print("%.3f %.3f %s %s" % (t, time.time(), c.group(1), w.group(1)), flush=True)
    time.sleep(0.1)
"""


class PowerWatch:
    """The shader clock and the package power the timed launches really run at: a small helper process (started before the
    warm-ups, so that nothing of it lies in the timed region; it touches no GPU runtime) loops over `rocm-smi --showclocks
    --showpower`, and the samples that fall between begin() and end() are kept.  The 8-sweep kernel holds an MI355X at its package
    power limit, and the clock it is given there, not the 2.4 GHz of the data sheet, is what its instruction rates have to be read
    against (profiles/r05_clock_power.txt).  Never raises; `stop()` returns None when the tool is not there or no sample fell
    into the window."""

    def __init__(self, gpu_index=0):
        self.gpu, self.p, self.t0, self.t1 = int(gpu_index), None, None, None

    def start(self):
        try:
            self.p = subprocess.Popen([sys.executable, "-c", POWER_HELPER, str(self.gpu)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                      env=child_env())
        except Exception:
            self.p = None
        return self

    def begin(self):
        self.t0 = time.time()

    def end(self):
        self.t1 = time.time()

    def stop(self):
        if self.p is None:
            return None
        try:
            self.p.terminate()
            out, _ = self.p.communicate(timeout=20)
        except Exception:
            try:
                self.p.kill()
            except Exception:
                pass
            return None
        if self.t0 is None or self.t1 is None:
            return None
        body = []
        for line in out.splitlines():
            try:
                a, b, c, w = line.split()
                if float(a) >= self.t0 and float(b) <= self.t1:      # (the whole query inside the timed region)
                    body.append((int(c), float(w)))
            except ValueError:
                pass
        if not body:
            return None
        clk, pw = sorted(c for c, _ in body), sorted(w for _, w in body)
        return {"sclk_MHz": clk[len(clk) // 2], "package_W": pw[len(pw) // 2], "samples": len(body), "spec_sclk_MHz": 2400,
                "source": "rocm-smi --showclocks --showpower looped by a helper process, the queries that lie inside the timed steps, medians (GPU[%d])" % self.gpu}


def measure_traffic(args, kernels, bands=1, ghost=32):
    """-> {kernel name: dict(hbm_bytes_per_launch, fetch_kb_raw, write_kb_raw, launches)} or {} with a reason"""
    exe = shutil.which("rocprofv3")
    if not exe:
        return {}, "rocprofv3 not on PATH"
    out = {}
    tmp = tempfile.mkdtemp(prefix="dmg_pmc_")
    try:
        vals = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "--output-format", "csv", "--",
                   sys.executable, str(ROOT / "bench.py"), "--pmc-child", "--width", str(args.width), "--height", str(args.height),
                   "--terrain", args.terrain, "--pmc-bands", str(bands), "--ghost", str(ghost)]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=child_env(TMPDIR=tmp), cwd=tmp)
            files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if r.returncode != 0 or not files:
                return {}, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, (r.stderr or r.stdout)[-160:])
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != counter:
                    continue
                hits = [k for k in kernels if k in row["Kernel_Name"]]
                if hits:                                  # (the longest name: k_jacobi_stream_twin is not k_jacobi_stream)
                    vals.setdefault((max(hits, key=len), counter), []).append(float(row["Counter_Value"]))
        for k in kernels:
            f, w = vals.get((k, "FETCH_SIZE")), vals.get((k, "WRITE_SIZE"))
            if not f or not w:
                continue
            # the first launches of a geometry may be the tuner's trials of other kernels/shapes: keep the steady tail
            f, w = f[len(f) // 2:], w[len(w) // 2:]
            fk, wk = sum(f) / len(f), sum(w) / len(w)
            # gfx950: FETCH_SIZE (KB) reports half of a wide coalesced read stream -> doubled; WRITE_SIZE (KB) is exact
            out[k] = dict(hbm_bytes_per_launch=2.0 * fk * 1024 + wk * 1024, fetch_kb_raw=fk, write_kb_raw=wk, launches=len(f))
        src = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (FETCH_SIZE x2: gfx950 correction)"
        if bands > 1:
            src += "; band 0 of %d (rank 0's geometry: %d owned + %d ghost rows), whole launches" % (bands, args.height // bands, ghost)
        return out, src
    except Exception as e:
        return {}, "PMC collection failed: %s" % str(e)[:160]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class Control:
    """The bench's control plane between the ranks: rendezvous, the RCCL id, barriers, max over ranks.  It runs over gloo (CPU
    tensors), so that it survives whatever happens to the library's RCCL communicators -- those carry the data path, and a
    leg that loses one must still be able to tell the other ranks; backend nccl is the fall-back if gloo does not come up."""

    def __init__(self, torch, world, rank, device):
        self.torch, self.world, self.rank, self.dist, self.dev = torch, world, rank, None, "cpu"
        if world == 1:
            return
        import datetime
        import torch.distributed as dist
        self.dist = dist
        try:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
        except Exception as e:
            print("bench: gloo control plane failed (%s); using nccl" % str(e)[:120], file=sys.stderr)
            dist.init_process_group("nccl", device_id=device)
            self.dev = device

    def allmax(self, vals):
        if self.world == 1:
            return list(vals)
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in t.cpu()]

    def barrier(self):
        self.allmax([0.0])

    def fresh_id(self, capi):
        """a new 128-byte RCCL id from rank 0 to everybody: one per communicator (an id serves ONE ncclCommInitRank round)"""
        if self.world == 1:
            return None
        idt = self.torch.zeros(128, dtype=self.torch.uint8, device=self.dev)
        if self.rank == 0:
            idt = self.torch.tensor(list(capi.Multi.unique_id()), dtype=self.torch.uint8, device=self.dev)
        self.dist.broadcast(idt, 0)
        return bytes(idt.cpu().tolist())

    def close(self):
        if self.dist is not None:
            try:
                self.dist.destroy_process_group()
            except Exception:
                pass


class Watchdog:
    """fire() on a thread of its own if arm(s) is not followed by disarm() within s seconds"""

    def __init__(self, fire):
        import threading
        self.fire, self.timer, self.threading = fire, None, threading

    def arm(self, seconds):
        self.disarm()
        self.timer = self.threading.Timer(seconds, self.fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def main():
    os.environ.setdefault("DMG_TEST_HOOKS", "1")      # the bench is harness: its children hand DMG_* hooks to the library too (capi._is_harness)
    argv = sys.argv[1:]
    args = parse(argv)
    if args.cpu_baseline_only:
        cpu_baseline_child(args)
        return
    if args.pmc_child:
        pmc_child(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.one_process:
        raise SystemExit(launcher(args, argv))
    worker(args)


def worker(args):
    import numpy as np
    import torch
    from demiurge_amd import capi, synth

    nb = args.gpus                                             # latitude bands = GPUs of the job
    world = int(os.environ.get("WORLD_SIZE", "1"))             # processes: N (one per GPU), or 1 (N = 1, or --one-process)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.one_process:
        world, rank, local_rank = 1, 0, 0
    elif world != nb:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (nb, world))
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: the HIP backend has no CPU fallback")
    # rehearsal on a one-GPU box (tests only): DMG_BENCH_REHEARSAL=1 keeps the N-process plumbing (rendezvous, id broadcast,
    # barrier, max over ranks) but RCCL refuses two ranks on one GPU, so every rank then drives all N bands itself on GPU 0
    # through the library's one-process form; with --one-process the N bands likewise all sit on GPU 0
    rehearsal = os.environ.get("DMG_BENCH_REHEARSAL") == "1"
    one_proc = world == 1 and nb > 1
    dev_index = 0 if (rehearsal or one_proc) else local_rank
    if one_proc and not rehearsal and torch.cuda.device_count() < nb:
        raise SystemExit("--one-process --gpus %d but %d devices visible" % (nb, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    ctl = Control(torch, world, rank, device)

    W, H, sweeps = args.width, args.height, args.sweeps
    ghost = max(1, min(args.ghost, H // nb))
    grid = capi.make_grid(W, H)
    kernel_of = {"auto": capi.JACOBI_AUTO, "single": capi.JACOBI_SINGLE, "blocked": capi.JACOBI_BLOCKED}
    order_flag = {"overlap": capi.MULTI_OVERLAP_LAST_LAUNCH, "exchange-then-sweep": capi.MULTI_EXCHANGE_THEN_SWEEP}
    transport_flag = {"rccl": capi.MULTI_RCCL, "store": capi.MULTI_STORE}
    row0 = [capi.lib().dmg_band_row0(H, nb, r) for r in range(nb + 1)]
    fields = {}

    def field(name):
        """the synthetic inputs, built once per process (every leg uploads the same bytes)"""
        if name not in fields:
            fields[name] = {"survey": lambda: synth.terrain(W, H), "ocean": lambda: -np.ones((H, W), np.float32),
                            "divw": lambda: synth.divw(W, H)}[name]()
        return fields[name]

    def make(kernel, terrain, order="exchange-then-sweep", transport="rccl", rccl_id=None, bands=None, quirks=0):
        prm = capi.default_ocean_params()
        prm.jacobi_kernel = kernel_of[kernel]
        prm.quirks |= quirks
        bands = nb if bands is None else bands
        if bands > 1 and (one_proc or rehearsal):
            # one process holds all the bands: a device each (ncclCommInitAll / peer-mapped stores), or -- rehearsal -- all of
            # them on GPU 0, where peer copies stand in for RCCL and the store transport is the real one
            if rehearsal:
                devs, tf = [0] * bands, (capi.MULTI_STORE if transport == "store" else capi.MULTI_PEER)
            else:
                devs, tf = list(range(bands)), transport_flag[transport]
            m = capi.Multi(grid, devices=devs, ghost=ghost, params=prm, flags=tf | order_flag[order])
        else:
            m = capi.Multi(grid, rank=rank if bands > 1 else 0, world=bands, unique_id=rccl_id, device=dev_index, ghost=ghost, params=prm,
                           flags=(transport_flag[transport] if bands > 1 else 0) | order_flag[order])
        m.set_terrain(field(terrain))
        m.set_divw(field("divw"))
        return m

    def timed(m, n_sweeps, steps, warmup):
        """one GPU: -> wall seconds, HIP-event ms on the kernels' stream, launches, sweeps"""
        for _ in range(warmup):
            m.jacobi(n_sweeps, zero_first=True)                   # OceanCurrents.cpp:495-508
        m.sync(); torch.cuda.synchronize(device)
        s0 = m.stats()
        t0 = time.perf_counter()
        m.timer_start()
        for _ in range(steps):
            m.jacobi(n_sweeps, zero_first=True)
        ev_ms = m.timer_stop()
        m.sync(); torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        s1 = m.stats()
        return dt, ev_ms, max(s1["jacobi_launches"] - s0["jacobi_launches"], 1), max(s1["jacobi_sweeps"] - s0["jacobi_sweeps"], 1), s0, s1

    def agree(local_err, vals=()):
        """A barrier that also tells every rank whether ANY rank has failed so far (and takes the max of `vals`).  Every rank
        calls it at the same points of a leg whatever happened to it locally, so the ranks' collectives never get out of step."""
        out = ctl.allmax([1.0 if local_err else 0.0] + list(vals))
        return out[0] > 0, out[1:]

    def band_checksums(m):
        """CRC-32 of every band's owned rows of the solved pressure, as a vector over ALL bands: this process fills the entries
        of the bands it holds (+1, so that an entry nobody filled stays 0), the max over ranks assembles the rest"""
        p = m.get(want=("pressure",))["pressure"]
        mine = range(nb) if (one_proc or rehearsal or nb == 1) else [rank]
        v = [0.0] * nb
        for r in mine:
            v[r] = float(zlib.crc32(memoryview(p[row0[r]:row0[r + 1]]).cast("B")) + 1)
        del p
        return v

    def leg(transport, order_name):
        """One timed leg (W warm-ups, then K solves between two barriers) on a handle of its own.  Never raises: -> dict with
        `value`, or with `error`.  A failure on one rank fails the leg on all of them; the handle is destroyed, never reused."""
        res = {"transport": transport, "order": order_name}
        m, err, dt, ev_ms, s0, s1 = None, None, None, None, None, None
        sums = [0.0] * nb

        def note(e):
            return "%s: %s" % (type(e).__name__, str(e)[:300])

        rccl_id = ctl.fresh_id(capi) if not rehearsal else None   # (collective: every rank, before anything can fail locally)
        watch = PowerWatch(dev_index).start() if rank == 0 and not args.no_extra_legs else None
        try:
            m = make(args.kernel, args.terrain, order_name, transport, rccl_id)
        except Exception as e:
            err = note(e)
        bad, _ = agree(err)
        if not bad:
            try:
                for _ in range(args.warmup):
                    m.jacobi(sweeps, zero_first=True)                 # OceanCurrents.cpp:495-508
                m.sync(); torch.cuda.synchronize(device)
            except Exception as e:
                err = note(e)
            bad, _ = agree(err)                                       # the barrier in front of the timed region
        if not bad:
            t0 = time.perf_counter()
            try:
                torch.cuda.synchronize(device)
                s0 = m.stats()
                if watch is not None:
                    watch.begin()
                t0 = time.perf_counter()
                m.timer_start()
                for _ in range(args.steps):
                    m.jacobi(sweeps, zero_first=True)
                ev_ms = m.timer_stop()
                m.sync(); torch.cuda.synchronize(device)
                if watch is not None:
                    watch.end()
            except Exception as e:
                err, ev_ms = note(e), 0.0
            bad, _ = agree(err)                                       # the barrier behind it
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            _, (dt, ev_ms) = agree(None, [dt, ev_ms or 0.0])          # max over ranks
        if not bad:
            try:
                s1 = m.stats()
                n_launch, n_sw = max(s1["jacobi_launches"] - s0["jacobi_launches"], 1), max(s1["jacobi_sweeps"] - s0["jacobi_sweeps"], 1)
                res.update(dt=dt, ev_ms=ev_ms, n_launch=n_launch, n_sweeps=n_sw, kernel=m.band(0).kernel_name(),
                           value=float(W) * H * sweeps * args.steps / dt / 1e6, ms_per_step=dt / args.steps * 1e3)
                if nb > 1:
                    x_ms, n_x = s1["exchange_ms"] - s0["exchange_ms"], s1["exchanges"] - s0["exchanges"]
                    res["exchange_ms"] = x_ms
                    res["halo_exchange"] = dict(
                        transport=capi.MULTI_TRANSPORT_NAMES.get(s1["transport"], "?"), order=order_name, exchanges=n_x, ms_total=x_ms,
                        share_of_stream_time=x_ms / max(ev_ms, 1e-9), bytes_per_exchange=(s1["exchange_bytes"] - s0["exchange_bytes"]) / max(n_x, 1),
                        warnings=s1["warnings"],
                        note="rank 0's HIP events around each exchange on the stream that carries it: transfer + the wait for the neighbours")
                    res["ranks_seen"] = dict(zip(("rccl_comm_count", "devices_of_this_process"), m.comm_ranks()), processes=world)
                    # what the leg computed (untimed): the last solve's pressure, band by band
                    sums = band_checksums(m)
            except Exception as e:
                err = note(e)
        try:
            if m is not None:
                m.close()
        except Exception as e:
            err = err or "close: %s" % e
        if watch is not None:
            res["clock"] = watch.stop()
        bad, sums = agree(err, sums)
        if bad:
            res = {"transport": transport, "order": order_name, "error": err or "another rank failed in this leg"}
        elif nb > 1:
            res["checksum"] = "-".join("%08x" % (int(x) - 1) if x > 0 else "????????" for x in sums)
        return res

    # ---- the timed legs ----
    # N = 1: one leg.  N > 1: the order and transport that have run before go first (exchange, then sweep, over RCCL), then the
    # others, each on a fresh handle inside its own try/except; the headline is the fastest leg that every rank got through.
    # A watchdog prints the line with what there is if a later leg hangs instead of failing.
    if nb == 1:
        plan = [("rccl", "exchange-then-sweep")]
    else:
        orders = ["exchange-then-sweep", "overlap"] if args.order == "auto" else [args.order]
        transports = ["rccl", "store"] if args.transport == "auto" else [args.transport]
        plan = [(t, o) for t in transports for o in orders]
        if args.no_extra_legs:
            plan = plan[:1]
    legs = []                  # every leg of every pass, in order: what the watchdog and the line read
    state = {"out": None, "drop": 0}

    def shown():
        return legs[state["drop"]:]

    def emit(partial_reason=None):
        """rank 0: the ONE JSON line, from whatever legs there are"""
        out = state["out"] if state["out"] is not None else headline(shown(), partial_reason)
        if partial_reason and state["out"] is not None:
            out["note"] = (out.get("note", "") + " " + partial_reason).strip()
        print(json.dumps(out), flush=True)

    def watchdog_fire():
        good = any("value" in l for l in shown())
        if rank == 0:
            emit("watchdog: a leg neither finished nor failed within %d s; the line holds the legs before it" % args.leg_timeout)
        os._exit(0 if good else 3)

    dog = Watchdog(watchdog_fire)

    def headline(legs_, partial_reason=None):
        good = [l for l in legs_ if "value" in l and "error" not in l]
        out = {
            "metric": "Mcell-steps/s, ocean Jacobi stencil @%dx%d fp32" % (W, H),
            "value": None, "unit": "Mcell-steps/s", "n_gpus": nb, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ocean-current Jacobi pressure solve, %dx%d full globe, %d sweeps per step "
                                   "(pressure zeroed first)%s" % (W, H, sweeps, "" if args.terrain == "survey" else ", all-ocean terrain"),
                       "parallelism": "1 GPU"},
        }
        if partial_reason:
            out["note"] = partial_reason
        if state.get("fallback"):
            out["note"] = (out.get("note", "") + " " + state["fallback"]).strip()
        if nb > 1:
            out["legs"] = [{k: v for k, v in l.items() if k in ("transport", "order", "value", "ms_per_step", "error", "halo_exchange", "units",
                                                                  "checksum")} for l in legs_]
        if not good:
            out["error"] = "no leg completed: " + "; ".join("%s/%s: %s" % (l["transport"], l["order"], l.get("error")) for l in legs_)
            return out
        best = max(good, key=lambda l: l["value"])
        out["value"], out["ms_per_step"] = best["value"], best["ms_per_step"]
        if nb > 1:
            blk = ghost - ghost % 8 if ghost >= 8 else ghost
            how = ("one process per GPU" if world > 1 else "ONE process, a host thread and stream per band") + (", REHEARSAL: all bands on GPU 0" if rehearsal else "")
            out["config"]["parallelism"] = ("%d latitude bands (dmg_multi, %s), %d ghost rows, halo every %d sweeps, %s, transport %s"
                                            % (nb, how, ghost, blk, best["order"], best["transport"]))
            out["halo_exchange"] = best.get("halo_exchange")
            out["ranks_seen"] = best.get("ranks_seen")
            rest = [l for l in legs_ if l is not best]
            if rest:
                # (the name round 3's line used for its second leg)
                o = next((l for l in rest if l["transport"] == best["transport"]), rest[0])
                out["other_exchange_order"] = ({"order": o["order"], "transport": o["transport"], "value": o["value"], "unit": "Mcell-steps/s",
                                                "ms_per_step": o["ms_per_step"]} if "value" in o and "error" not in o else
                                               "%s/%s failed: %s" % (o["transport"], o["order"], o["error"]))
        return out

    def run_plan(units=None):
        """the legs of one pass; each leg's band checksums must equal those of the pass's first completed leg (the order and
        transport that have run before), or the leg is kept with an `error` and cannot become the headline"""
        first = len(legs)
        for i, (t, o) in enumerate(plan):
            if i > 0:
                dog.arm(args.leg_timeout)       # the first leg has nothing to fall back on; later ones may not take the line with them
            l = leg(t, o)
            dog.disarm()
            if units:
                l["units"] = units
            ref = next((x for x in legs[first:] if "checksum" in x and "error" not in x), None)
            if ref is not None and "checksum" in l and l["checksum"] != ref["checksum"]:
                l["error"] = ("solved pressure differs from the %s/%s leg's (band CRCs %s, there %s): not a candidate for the headline"
                              % (ref["transport"], ref["order"], l["checksum"], ref["checksum"]))
            legs.append(l)
        return [l for l in legs[first:] if "value" in l and "error" not in l]

    good = run_plan()
    if not good and os.environ.get("DMG_ST_TWIN") != "0":
        # Bands of up to 4400 rows run the streaming kernel's twin units (round 4), which no run between two devices has seen
        # yet.  If EVERY leg failed, the plan is run once more with plain units (the library reads DMG_ST_TWIN whenever it cuts
        # a launch into units; every handle of the second pass is new): the legs are agreed between the ranks, so all of them
        # take this branch together.
        first_errors = "; ".join("%s/%s: %s" % (l["transport"], l["order"], l.get("error")) for l in legs)
        os.environ["DMG_ST_TWIN"] = "0"
        if rank == 0:
            print("bench: every leg failed (%s); once more with plain units (DMG_ST_TWIN=0)" % first_errors[:600], file=sys.stderr, flush=True)
        for l in legs:
            l["units"] = "twin"
        n_first = len(legs)
        good = run_plan("plain (fall-back)")
        if good:
            state["fallback"] = "every leg with twin units failed (%s); these are plain units (DMG_ST_TWIN=0)" % first_errors[:600]
            state["drop"] = n_first

    if not good:
        # nothing to report but the reasons: rank 0 prints them; the others leave only once it has (a rank that exits non-zero
        # makes the launcher end the rest)
        if rank == 0:
            emit()
        try:
            ctl.barrier()
        except Exception:
            pass
        ctl.close()
        raise SystemExit(3)
    if rank != 0:
        ctl.close()
        return
    best = max(good, key=lambda l: l["value"])
    dt, ev_ms, n_launch, n_sweeps, kname = best["dt"], best["ev_ms"], best["n_launch"], best["n_sweeps"], best["kernel"]
    band_cells = float(W) * (H / nb)
    sweeps_per_launch = n_sweeps / n_launch
    # the kernels' share of band 0's stream: with exchange-then-sweep the exchanges sit on the same stream and are taken out;
    # in the overlapped order they run beside the interior units on the second stream
    own_ms = ev_ms - (best.get("exchange_ms", 0.0) if best["order"] == "exchange-then-sweep" else 0.0)
    launch_s = (max(own_ms, 1e-6) / 1e3) / n_launch

    algo_bytes = ALGO_BYTES_PER_CELL_STEP * band_cells * sweeps_per_launch
    algo_gbs = algo_bytes / launch_s / 1e9
    # the kernel skips cells that are land (they hold +0): the arithmetic side counts the water cells only
    water = 1.0 if args.terrain != "survey" else float((field("survey")[row0[0]:row0[1]] <= 0).mean())
    tflops = FLOP_PER_CELL_STEP * band_cells * water * sweeps_per_launch / launch_s / 1e12
    out = headline(shown())
    state["out"] = out            # (from here on a watchdog would print this)
    extra = not args.no_extra_legs and args.kernel == "auto"
    if nb > 1:
        dog.arm(args.leg_timeout + 300)      # the profiled children and the CPU leg may not take the measured line with them

    # ---- HBM traffic of both kernels, measured by a profiled child of this run (N > 1: on rank 0's band geometry) ----
    traffic, traffic_src = ({}, "not collected")
    if extra:
        traffic, traffic_src = measure_traffic(args, JACOBI_KERNELS, bands=nb, ghost=ghost)

    def traffic_of(k):          # measured in this run, or nothing: a committed figure would be another build's
        if k in traffic:
            return traffic[k]["hbm_bytes_per_launch"], traffic_src
        return None, traffic_src

    tb, tsrc = traffic_of(kname)
    roof = {"kernel": kname, "launch_us": launch_s * 1e6, "sweeps_per_launch": sweeps_per_launch, "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "traffic": tb, "traffic_source": tsrc,
            "algorithmic_bytes_per_launch": algo_bytes, "algorithmic_GBps": algo_gbs, "algorithmic_x_peak": algo_gbs / HBM_PEAK_GBS,
            "fp32": {"flop_per_cell_sweep": FLOP_PER_CELL_STEP, "cells_counted": "water cells (%.1f %% of the %s): land is skipped"
                                                                                 % (100 * water, "grid" if nb == 1 else "band"),
                     "achieved": tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP32_PEAK_TFLOPS,
                     "issuable_peak": FP32_ISSUABLE_TFLOPS, "frac_of_issuable": tflops / FP32_ISSUABLE_TFLOPS,
                     "issuable_note": "packed FMAs measured at two waves per SIMD (this kernel's occupancy) issue 106 TFLOP/s, 121 at four "
                                      "(tools/pk_probe.hip, profiles/r03_valu_issue_probe.log): no instruction stream reaches the spec peak"}}
    clock = best.get("clock")
    if clock:
        # what the launches really ran at: the package power limit holds the shader clock below the data sheet's 2.4 GHz
        roof["clock"] = clock
        at = FP32_PEAK_TFLOPS * clock["sclk_MHz"] / clock["spec_sclk_MHz"]
        roof["fp32"].update(peak_at_clock=at, frac_at_clock=tflops / at, issuable_at_clock=FP32_ISSUABLE_TFLOPS * clock["sclk_MHz"] / clock["spec_sclk_MHz"],
                            frac_of_issuable_at_clock=tflops / (FP32_ISSUABLE_TFLOPS * clock["sclk_MHz"] / clock["spec_sclk_MHz"]))
    if algo_gbs > HBM_PEAK_GBS:
        # the 16 B per cell-sweep model stops bounding a launch that keeps 8 sweeps on chip: the vector side is the figure of merit
        fm = roof["fp32"].get("frac_of_issuable_at_clock", roof["fp32"]["frac_of_issuable"])
        roof["figure_of_merit"] = {"name": "fp32.frac_of_issuable_at_clock" if "frac_of_issuable_at_clock" in roof["fp32"] else "fp32.frac_of_issuable",
                                   "value": fm, "frac_at_clock": roof["fp32"].get("frac_at_clock"),
                                   "why": "algorithmic_x_peak > 1: the byte model does not bound the launch"}
    if nb > 1:
        roof["band"] = ("band 0 of %d (rank 0): %d owned + %d ghost rows; launch_us = band 0's stream time per launch%s"
                        % (nb, row0[1] - row0[0], ghost, ", halo exchanges taken out" if best["order"] == "exchange-then-sweep" else
                           " (overlapped order: the exchange runs beside the interior units)"))
    if sweeps_per_launch > 1:
        # (the label says what the note says.  8 sweeps per HBM pass: the launch is bound by its vector
        # instructions at the clock the package power limit leaves it, not by its bytes; `achieved` / `frac` stay the HBM side)
        roof["bound"] = "valu (fp32 vector issue at the power-limited clock); hbm side: achieved / frac"
        if tb:
            roof["achieved"] = tb / launch_s / 1e9
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
            roof["shape_ceiling"] = HBM_ROWS_GBS
            roof["frac_of_shape_ceiling"] = roof["achieved"] / HBM_ROWS_GBS
            roof["note"] = ("8 sweeps per launch on chip: `achieved` is the HBM traffic the launch really moves (PMC) over its "
                            "duration.  It is not what the launch waits for alone: the kernel is CO-LIMITED at the package power limit.  "
                            "These launches hold the chip at ~1400 W; the shader clock measured INSIDE the kernel (s_memtime / s_memrealtime "
                            "per unit, profiles/r06_clock_power.txt) is 1.96 GHz (`clock`, rocm-smi sampled beside the timed steps, reads "
                            "~2.05; a device copy runs at 2.4 GHz and 850 W), at which the vector pipe is 76-80 % occupied, nine tenths of "
                            "it by the formula's own instructions, 90 % with the launch's tail -- while the launch moves 0.8 of the bytes per "
                            "second the one-sweep kernel moves in this same run (`demonstrated_GBps`).  Halving the bytes (round 5's deep form: "
                            "16 sweeps per trip, bit-identical) buys nothing in the exact arithmetic; halving the arithmetic (the `tolerance` "
                            "rendering, within north_star's 1e-5) buys 9 % with this kernel and 25 % with the deep form (DESIGN.md 4.1b, 4.1c).  "
                            "The SURVEY 8d model (16 B per cell-sweep) stops bounding a kernel that does 8 sweeps per HBM pass: "
                            "`algorithmic_x_peak`; the HBM-bound form the 60 % target is stated for is `roofline_single`")
        else:
            roof["achieved"], roof["frac"] = None, None
            roof["note"] = "no PMC traffic for this configuration (%s); see algorithmic_x_peak and fp32" % tsrc
    else:
        roof.update(bound="hbm", achieved=algo_gbs, frac=algo_gbs / HBM_PEAK_GBS, note="one sweep per launch: HBM-bound, 16 algorithmic bytes per cell-sweep")
    out["roofline"] = roof

    if extra:
        # ---- this box's copy ceiling: a 1 GiB device-to-device copy (read + write = 2 GiB of traffic), HIP events ----
        try:
            a = torch.empty(1 << 28, dtype=torch.float32, device=device)
            bb = torch.empty_like(a)
            bb.copy_(a); torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                bb.copy_(a)
            e1.record(); torch.cuda.synchronize(device)
            gbs = 5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) / 1e3) / 1e9
            roof["copy_ceiling"] = {"value": gbs, "unit": "GB/s", "sample": "5 device copies of 1 GiB (read + write counted), after one warm-up"}
            if roof.get("achieved"):
                roof["frac_of_copy_ceiling"] = roof["achieved"] / gbs
            del a, bb
            torch.cuda.empty_cache()
        except Exception as e:
            roof["copy_ceiling"] = {"value": None, "error": str(e)[:120]}

    if extra and nb == 1:
        # ---- one sweep per launch: the HBM-bound form the north_star's 60 % target is stated for ----
        m1 = make("single", args.terrain)
        n1 = min(sweeps, 1500)
        dt1, ev1, l1, sw1, _, _ = timed(m1, n1, 1, 1)
        ls1 = (ev1 / 1e3) / l1
        a1 = ALGO_BYTES_PER_CELL_STEP * band_cells / ls1 / 1e9
        k1 = m1.band(0).kernel_name()
        t1, t1src = traffic_of(k1)
        out["roofline_single"] = {"bound": "hbm", "kernel": k1, "achieved": a1, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a1 / HBM_PEAK_GBS,
                                  "traffic": t1, "traffic_source": t1src, "launch_us": ls1 * 1e6, "sweeps_per_launch": sw1 / l1,
                                  "value": float(W) * H * n1 / dt1 / 1e6, "sample": "1 solve of %d sweeps, one per launch, after 1 warm-up" % n1}
        if t1:
            # what this box's HBM demonstrably delivers to a streaming kernel in this very run: the one-sweep kernel's real traffic rate
            # (a torch device copy -- copy_ceiling -- reads lower than that on these boxes)
            roof["demonstrated_GBps"] = {"value": t1 / ls1 / 1e9, "what": "%s: PMC traffic / launch time, this run" % k1}
            if roof.get("achieved"):
                roof["frac_of_demonstrated"] = roof["achieved"] / (t1 / ls1 / 1e9)
        # ---- untimed check: 64 sweeps of the default kernel == 64 single-sweep launches, bit for bit ----
        md = make("auto", args.terrain)
        md.jacobi(64, zero_first=True)
        m1.jacobi(64, zero_first=True)
        pa, pb = md.get(want=("pressure",))["pressure"], m1.get(want=("pressure",))["pressure"]
        same = bool(np.array_equal(pa, pb))
        out["check"] = {"what": "64 sweeps: %s vs %s" % (md.band(0).kernel_name(), k1), "bit_identical": same,
                        "max_abs_diff": 0.0 if same else float(np.abs(pa - pb).max()), "nonzero_cells": int(np.count_nonzero(pa))}
        del pa, pb
        m1.close(); md.close()
        # ---- the tolerance rendering (DMG_QUIRK_JACOBI_TOLERANCE: fused row coefficients, within north_star's 1e-5; NOT the headline,
        # which stays the bit-exact rendering): its rate, and how far its solved pressure is from the exact rendering's ----
        try:
            mt = make("auto", args.terrain, quirks=capi.QUIRK_JACOBI_TOLERANCE)
            dtt, evt, lt_, swt, _, _ = timed(mt, sweeps, 1, 1)
            me = make("auto", args.terrain)
            me.jacobi(sweeps, zero_first=True)
            pt, pe = mt.get(want=("pressure",))["pressure"], me.get(want=("pressure",))["pressure"]
            pmax = float(np.abs(pe).max())
            dmax = max(float(np.abs(pt[y:y + 512].astype(np.float64) - pe[y:y + 512]).max()) for y in range(0, H, 512))
            out["tolerance"] = {"value": float(W) * H * sweeps / dtt / 1e6, "unit": "Mcell-steps/s", "ms_per_step": dtt * 1e3,
                                "launch_us": evt / lt_ * 1e3, "sweeps_per_launch": swt / lt_,
                                "kernel": ("k_jacobi_deep<3> (16 sweeps per trip through HBM, two waves per unit) + k_jacobi_stream<3> for the last 8"
                                           if swt / lt_ > 8.5 else mt.band(0).kernel_name() + "<3>"), "max_rel": dmax / max(pmax, 1e-30),
                                "max_rel_what": "max|p_tol - p_exact| / max|p_exact| after %d sweeps from zero pressure, this run" % sweeps,
                                "bound": 1e-5, "sample": "1 solve after 1 warm-up",
                                "what": "DMG_QUIRK_JACOBI_TOLERANCE: a(pW+pE) + b(pS+pN) - c bC with split row coefficients (three fused "
                                        "multiply-adds per cell, bit-equal to the oracle's form 3); opt-in, the headline above is the exact rendering"}
            del pt, pe
            mt.close(); me.close()
        except Exception as e:
            out["tolerance"] = {"value": None, "error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        # ---- all-ocean terrain: nothing to skip ----
        mo = make("auto", "ocean")
        dto, evo, lo_, swo, _, _ = timed(mo, sweeps, 1, 1)
        out["all_ocean"] = {"value": float(W) * H * sweeps / dto / 1e6, "unit": "Mcell-steps/s", "ms_per_step": dto * 1e3,
                            "launch_us": evo / lo_ * 1e3, "kernel": mo.band(0).kernel_name(), "sample": "1 solve after 1 warm-up",
                            "fp32_TFLOPs": FLOP_PER_CELL_STEP * float(W) * H * sweeps / dto / 1e12,
                            "fp32_frac": FLOP_PER_CELL_STEP * float(W) * H * sweeps / dto / 1e12 / FP32_PEAK_TFLOPS}
        mo.close()
    elif nb > 1:
        # the bands' solve against ONE band's: what `check` is for N = 1 is the per-leg checksum here (every leg equals the
        # first); the first itself is pinned to the single-band kernels by the -m gpu tests (tests/test_multi.py)
        ref = next((l for l in shown() if "checksum" in l and "error" not in l), None)
        out["check"] = {"what": "CRC-32 of every band's owned rows of the solved pressure, each leg against the first completed leg",
                        "reference_leg": "%s/%s" % (ref["transport"], ref["order"]) if ref else None,
                        "legs_equal": [l["transport"] + "/" + l["order"] for l in shown() if ref and l.get("checksum") == ref["checksum"] and "error" not in l],
                        "legs_different": [l["transport"] + "/" + l["order"] for l in shown() if ref and "checksum" in l and l["checksum"] != ref["checksum"]]}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    dog.disarm()
    emit()
    ctl.close()


if __name__ == "__main__":
    main()
