import os
import subprocess
import sys
from pathlib import Path

# dmg_multi's overlapped exchange order needs eight hardware queues, asked for before HIP initialises; the
# library does not edit the environment itself
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import pytest

# the tests are the harness: their child processes (workers, `python -c` probes) hand DMG_* hooks to the library too (capi._is_harness)
os.environ.setdefault("DMG_TEST_HOOKS", "1")

# torch bundles its own libamdhip64; it must be loaded BEFORE libdemiurge_hip.so so that both resolve to ONE HIP
# runtime in this process (loading /opt/rocm's first leaves torch unable to see the GPU: "No HIP GPUs are available")
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure).  Built on demand with plain gcc."""
    lib = ROOT / "oracle" / "liboracle.so"
    src = ROOT / "oracle" / "dmg_oracle.c"
    src2 = ROOT / "oracle" / "flow_oracle.cpp"
    srcs = [src, src2, ROOT / "oracle" / "tect_oracle.c", ROOT / "oracle" / "noise_oracle.c"]
    if not lib.exists() or lib.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
        subprocess.check_call(["make", "-C", str(ROOT / "oracle"), "liboracle.so"])
    import oracle as orc
    orc.set_linear_taps(False)
    return orc


def load_golden(name):
    z = np.load(GOLDEN / (name + ".npz"))
    return {k: z[k] for k in z.files}


def relerr(a, ref):
    """max |a-ref| / max |ref|: the norm every parity tolerance in this suite is stated in."""
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30))
