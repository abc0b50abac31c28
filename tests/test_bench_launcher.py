"""`python bench.py --gpus N` started bare (the way the driver starts it) is a launcher: it must hand the work to FRESH child
processes -- torch.distributed.run with one process per GPU first, one process driving all N devices if that gives no value --
and must itself import neither torch nor the library nor touch HIP (a process that has initialised the GPU may not be
replaced or forked into ranks on this pool).  CPU only: the children are faked, or fail for want of a GPU."""
import json
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

LINE = {"metric": "Mcell-steps/s, ocean Jacobi stencil @16384x8192 fp32", "value": 123.0, "unit": "Mcell-steps/s", "n_gpus": 2}

DRIVER = textwrap.dedent("""
    import json, sys
    sys.path.insert(0, %(root)r)
    sys.argv = ["bench.py"] + %(argv)r
    import bench
    script = %(script)r
    calls = []
    def fake(cmd, timeout, env):
        calls.append({"cmd": cmd, "timeout": timeout, "world_in_env": "WORLD_SIZE" in env, "rank_in_env": "RANK" in env})
        rc, out = script[len(calls) - 1]
        return rc, out
    args = bench.parse(sys.argv[1:])
    rc = bench.launcher(args, sys.argv[1:], popen=fake)
    loaded = sorted(m for m in sys.modules if m == "torch" or m.startswith("torch.") or m.startswith("demiurge_amd") or m == "numpy")
    print("RESULT " + json.dumps({"rc": rc, "calls": calls, "loaded": loaded}))
""")


def drive(argv, script, env=None):
    code = DRIVER % {"root": str(ROOT), "argv": argv, "script": script}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=str(ROOT),
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(next(l for l in r.stdout.splitlines() if l.startswith("RESULT "))[7:])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    return res, lines, r


def test_launcher_starts_torchrun_in_a_fresh_child_and_relays_the_line():
    argv = ["--gpus", "2", "--steps", "1", "--warmup", "0"]
    res, lines, _ = drive(argv, [[0, "noise\n" + json.dumps(LINE) + "\n"]], env={"WORLD_SIZE": "7", "RANK": "3"})
    assert res["rc"] == 0 and len(res["calls"]) == 1 and lines == [LINE]
    cmd = res["calls"][0]["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-len(argv):] == argv and cmd[-len(argv) - 1].endswith("bench.py")
    # the launcher itself has loaded neither torch nor the library (nor numpy), and a stray rank environment does not leak
    assert res["loaded"] == [] and not res["calls"][0]["world_in_env"] and not res["calls"][0]["rank_in_env"]


def test_launcher_falls_back_to_one_process_when_the_ranks_give_no_value():
    argv = ["--gpus", "4", "--steps", "1", "--warmup", "0", "--launch-timeout", "90"]
    failed = dict(LINE, value=None, error="no leg completed: rccl/exchange-then-sweep: boom")
    res, lines, r = drive(argv, [[3, json.dumps(failed) + "\n"], [0, json.dumps(dict(LINE, n_gpus=4)) + "\n"]])
    assert res["rc"] == 0 and len(res["calls"]) == 2 and res["loaded"] == []
    second = res["calls"][1]["cmd"]
    assert second[1].endswith("bench.py") and second[-1] == "--one-process" and second[2:-1] == argv and res["calls"][1]["timeout"] == 60
    assert len(lines) == 1 and lines[0]["value"] == 123.0 and "one process driving 4 devices" in lines[0]["note"] and "boom" in lines[0]["note"]
    assert "once more from ONE process" in r.stderr


def test_launcher_times_out_a_hung_child_and_still_prints_one_line():
    argv = ["--gpus", "2"]
    res, lines, _ = drive(argv, [[None, ""], [None, ""]])
    assert res["rc"] == 3 and len(lines) == 1 and lines[0]["value"] is None and lines[0]["n_gpus"] == 2
    assert "no exit within" in lines[0]["error"] and "one-process child" in lines[0]["error"]


def test_bare_command_end_to_end_without_a_gpu():
    """the driver's command on a box without a GPU: both children start for real, neither finds a device, and the launcher
    prints its one line with the reasons and exits non-zero -- no usage message, no traceback of its own"""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--width", "256", "--height", "128",
                        "--sweeps", "16"], capture_output=True, text=True, timeout=600, cwd=str(ROOT),
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert r.returncode == 3 and len(lines) == 1, r.stdout[-500:] + r.stderr[-1500:]
    assert lines[0]["value"] is None and lines[0]["n_gpus"] == 2 and "torch.distributed.run child: rc" in lines[0]["error"]
    assert "no GPU visible" in r.stderr


def test_run_child_ends_the_process_group_it_started():
    sys.path.insert(0, str(ROOT))
    import importlib
    bench = importlib.import_module("bench")
    rc, out = bench.run_child([sys.executable, "-c", "import time,sys; print('x', flush=True); time.sleep(60)"], 2, dict(os.environ))
    assert rc is None and out.startswith("x")
    rc, out = bench.run_child([sys.executable, "-c", "print('{\"metric\": 1}')"], 30, dict(os.environ))
    assert rc == 0 and "metric" in out


def test_power_watch_reads_a_looping_rocm_smi(tmp_path):
    """bench.PowerWatch: a helper process loops over `rocm-smi --showclocks --showpower`; the queries that lie between begin() and end()
    give the medians the bench line carries as roofline.clock.  Here against a stand-in for the tool (this container has no GPU);
    without the tool stop() answers None and nothing is raised."""
    import importlib.util
    import os
    import stat
    import time
    spec = importlib.util.spec_from_file_location("bench_under_test", str(ROOT / "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fake = tmp_path / "rocm-smi"
    fake.write_text("#!/bin/sh\n"
                    "echo '============ ROCm System Management Interface ============'\n"
                    "echo 'GPU[0]\t\t: sclk clock level: 1: (2051Mhz)'\n"
                    "echo 'GPU[1]\t\t: sclk clock level: 1: (111Mhz)'\n"
                    "echo 'GPU[0]\t\t: Current Socket Graphics Package Power (W): 1388.0'\n"
                    "echo 'GPU[1]\t\t: Current Socket Graphics Package Power (W): 90.0'\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    old = os.environ["PATH"]
    os.environ["PATH"] = str(tmp_path) + os.pathsep + old
    try:
        w = bench.PowerWatch(0).start()
        time.sleep(0.5)
        w.begin()
        time.sleep(1.5)
        w.end()
        got = w.stop()
    finally:
        os.environ["PATH"] = old
    assert got and got["sclk_MHz"] == 2051 and got["package_W"] == 1388.0 and got["samples"] >= 2, got
    # no tool at all: nothing to report, nothing raised
    os.environ["PATH"] = str(tmp_path / "nowhere")
    try:
        w = bench.PowerWatch(0).start()
        w.begin(); time.sleep(0.3); w.end()
        assert w.stop() is None
    finally:
        os.environ["PATH"] = old
