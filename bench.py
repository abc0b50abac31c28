#!/usr/bin/env python3
"""bench.py -- Mcell-steps/s of the ocean-current Jacobi pressure stencil (BASELINE.json metric).

A "step" is one pressure solve of the reference (OceanCurrents::jacobi, src/filter/OceanCurrents.cpp:437-509):
pressure := 0 then `--sweeps` Jacobi sweeps (reference value 5000) over the synthetic 16384x8192 full-globe
grid of SURVEY.md section 8d, inputs resident in HBM before the timed region.

    python bench.py [--gpus N --steps K --warmup W]            one process per GPU; N>1 via torch.distributed.run

N>1 splits the SAME grid into N latitude bands (strong scaling), ghost rows refreshed with RCCL send/recv.
Prints ONE JSON line on rank 0 with `roofline` (algorithmic bytes / measured launch time vs HBM peak) and, at N=1,
`cpu_baseline` (the reference GLSL on llvmpipe when oracle/_ref + Mesa are usable on this host, else the C port).
"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
ALGO_BYTES_PER_CELL_STEP = 16  # read p, read divw, read terrain, write p'


def parse():
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


# --------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N=1): runs in a child process so a missing/fragile Mesa cannot take the bench down
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
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        for line in out.stdout.splitlines():
            if line.startswith("CPU_BASELINE_JSON "):
                return json.loads(line[len("CPU_BASELINE_JSON "):])
        return dict(value=None, unit="Mcell-steps/s", cores=0, kind="port",
                    sample="cpu baseline failed: " + (out.stderr or out.stdout)[-200:])
    except Exception as e:
        return dict(value=None, unit="Mcell-steps/s", cores=0, kind="port", sample="cpu baseline failed: %s" % e)


def main():
    args = parse()
    if args.cpu_baseline_only:
        cpu_baseline_child(args)
        return
    import torch
    from demiurge_amd import bands, capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: the HIP backend has no CPU fallback")
    # rehearsal switches (tests only): DMG_BENCH_SHARE_GPU=1 puts every rank on GPU 0, DMG_DIST_BACKEND=gloo stages
    # the halos through the host -- together they run the N-rank code path on a one-GPU box
    dev_index = 0 if os.environ.get("DMG_BENCH_SHARE_GPU") == "1" else local_rank
    backend = os.environ.get("DMG_DIST_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    W, H, sweeps = args.width, args.height, args.sweeps
    ghost = max(1, min(args.ghost, H // world))
    stream = torch.cuda.current_stream(device)
    ctx = capi.Context(dev_index, stream=stream.cuda_stream)
    grid = capi.make_grid(W, H)
    prm = capi.default_ocean_params()
    prm.jacobi_kernel = {"auto": capi.JACOBI_AUTO, "single": capi.JACOBI_SINGLE, "blocked": capi.JACOBI_BLOCKED}[args.kernel]
    oc = capi.Ocean(ctx, grid, params=prm, band_rank=rank, band_count=world, ghost=ghost)
    img = synth.terrain(W, H) if args.terrain == "survey" else -np.ones((H, W), np.float32)
    oc.set_terrain(img)
    del img
    b = synth.divw(W, H)
    oc.set_divw(b)
    del b

    exch = bands.HaloExchanger(dist, rank, world, device) if world > 1 else None

    def solve():
        bands.jacobi_solve(oc, exch, sweeps, ghost, zero_first=True)   # OceanCurrents.cpp:495-508

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        solve()
    fence()
    l0, s0 = oc.counters()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        solve()
    ev_ms = ctx.timer_stop()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt, ev_ms], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, ev_ms = float(tt[0]), float(tt[1])

    if rank == 0:
        cell_steps = float(W) * H * sweeps * args.steps
        value = cell_steps / dt / 1e6
        # dominant kernel = one Jacobi sweep launch over this rank's band; average launch time from HIP events
        l1, s1 = oc.counters()
        n_launch, n_sweeps = max(l1 - l0, 1), max(s1 - s0, 1)
        sweeps_per_launch = n_sweeps / n_launch
        band_cells = float(W) * (H / world)
        launch_s = (ev_ms / 1e3) / n_launch
        algo_bytes = ALGO_BYTES_PER_CELL_STEP * band_cells * sweeps_per_launch
        achieved = algo_bytes / launch_s / 1e9
        kname = "k_jacobi_tb8" if sweeps_per_launch > 1 else "k_jacobi_sweep_v4"
        traffic = None                     # measured separately with rocprofv3 --pmc (profiles/README.md)
        try:
            t = json.load(open(ROOT / "profiles" / "pmc_traffic.json")).get(kname)
            if t and world == 1 and (t["W"], t["H"]) == (W, H) and sweeps_per_launch in (1.0, 8.0) \
                    and args.terrain == "survey":
                traffic = t["hbm_bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "Mcell-steps/s, ocean Jacobi stencil @%dx%d fp32" % (W, H),
            "value": value, "unit": "Mcell-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ocean-current Jacobi pressure solve, %dx%d full globe, %d sweeps per step "
                                   "(pressure zeroed first)%s" % (W, H, sweeps, "" if args.terrain == "survey" else ", all-ocean terrain"),
                       "parallelism": "1 GPU" if world == 1 else "%d latitude bands, %d ghost rows, RCCL halo every %d sweeps"
                                      % (world, ghost, ghost),
                       "device": ctx.name()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kname, "launch_us": launch_s * 1e6, "sweeps_per_launch": sweeps_per_launch,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": ("8 sweeps per launch on chip: real HBM traffic (`traffic`) is ~1/10 of the algorithmic "
                                  "bytes, so frac > 1; the sweep phase is VALU-bound (DESIGN.md 4.1)") if sweeps_per_launch > 1
                                 else "one sweep per launch: HBM-bound"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    oc.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
