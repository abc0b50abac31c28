"""ctypes binding of libdemiurge_hip.so (the C ABI in include/demiurge_hip.h).

This module is plumbing for the tests and bench.py; the product is the shared library.  It fails
loudly when the library is missing or no MI355X is visible -- there is no CPU fallback.
"""
import ctypes
import os
import sys
import weakref
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
# (DMG_LIB: another build of the same library, for A/B measurements of two builds in one GPU call -- tools only)
LIB_PATH = Path(os.environ["DMG_LIB"]) if os.environ.get("DMG_LIB") else _HERE / "libdemiurge_hip.so"
ROOT = _HERE.parent

OK, EINVAL, ENOMEM, EHIP, ENODEVICE, ESTATE = 0, -1, -2, -3, -4, -5
F_PRESSURE, F_DIVW, F_TERRAIN, F_V = range(4)
SIDE_SOUTH, SIDE_NORTH = 0, 1
QUIRK_DIFFUSION_SCALAR_RHS = 1
QUIRK_LLVMPIPE_RASTER = 2
QUIRK_EXACT_LLVMPIPE = 4            # round 5's bit: accepted and ignored
QUIRK_FAST_TRANSCENDENTALS = 8
QUIRK_JACOBI_TOLERANCE = 16
QUIRK_TOLERANCE = QUIRK_FAST_TRANSCENDENTALS | QUIRK_JACOBI_TOLERANCE
PRIVATE_STREAM = 1
JACOBI_AUTO, JACOBI_SINGLE, JACOBI_BLOCKED = 0, 1, 2
FULL_GLOBE = (np.float32(-np.pi / 2), np.float32(np.pi / 2), np.float32(-np.pi), np.float32(np.pi))  # Project.h:97


class DmgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dmg error %d: %s" % (code, msg))
        self.code = code


class Grid(ctypes.Structure):
    _fields_ = [("W", ctypes.c_int), ("H", ctypes.c_int), ("coords", ctypes.c_float * 4),
                ("circumference", ctypes.c_float)]


class OceanParams(ctypes.Structure):
    _fields_ = [("pressurefactor", ctypes.c_float), ("jacobi_iters", ctypes.c_int),
                ("diffusion_iters", ctypes.c_int), ("timestep_h", ctypes.c_float), ("scale", ctypes.c_float),
                ("wind", ctypes.c_float), ("quirks", ctypes.c_int), ("jacobi_kernel", ctypes.c_int)]


class ClimateParams(ctypes.Structure):
    """dmg_climate_params: every literal of Temperature.cpp as a field"""
    _fields_ = [("steps_per_year", ctypes.c_int), ("year_seconds", ctypes.c_float), ("diffusivity", ctypes.c_float), ("albedo", ctypes.c_float),
                ("olr_coeff", ctypes.c_float), ("olr_ref_kelvin", ctypes.c_float), ("emissivity", ctypes.c_float), ("celsius_offset", ctypes.c_float),
                ("heat_capacity_land", ctypes.c_float), ("heat_capacity_ocean", ctypes.c_float), ("eccentricity", ctypes.c_float),
                ("obliquity_deg", ctypes.c_float), ("perihelion_deg", ctypes.c_float), ("solar_constant", ctypes.c_float),
                ("initial_temperature", ctypes.c_float), ("quirks", ctypes.c_int)]


CLIMATE_QUIRK_ACOS_NAN = 1


class WorldParams(ctypes.Structure):
    _fields_ = [("tectonics_steps", ctypes.c_int), ("erosion_rounds", ctypes.c_int), ("climate_steps", ctypes.c_int),
                ("erosion_N", ctypes.c_int), ("exponent", ctypes.c_float), ("slope_exponent", ctypes.c_float), ("factor", ctypes.c_float),
                ("lakes", ctypes.c_int), ("climate", ClimateParams)]


class MultiStats(ctypes.Structure):
    _fields_ = [("bands_total", ctypes.c_int), ("bands_local", ctypes.c_int), ("transport", ctypes.c_int), ("order", ctypes.c_int),
                ("ghost", ctypes.c_int), ("warnings", ctypes.c_int), ("exchanges", ctypes.c_longlong), ("exchange_ms", ctypes.c_double),
                ("exchange_bytes", ctypes.c_longlong), ("jacobi_launches", ctypes.c_longlong), ("jacobi_sweeps", ctypes.c_longlong)]


MULTI_AUTO, MULTI_RCCL, MULTI_PEER, MULTI_RCCL_LOOPBACK, MULTI_STORE = 0, 1, 2, 3, 4
MULTI_EXCHANGE_THEN_SWEEP, MULTI_OVERLAP_LAST_LAUNCH = 0, 16
MULTI_TRANSPORT_NAMES = {1: "rccl", 2: "peer-copies", 3: "rccl-loopback", 4: "store"}
MULTI_STORE_BLOB_BYTES = 192
MULTI_WARN_HWQ_FALLBACK = 1
F_PRESSURE, F_DIVW, F_TERRAIN, F_V, F_VADVECT = 0, 1, 2, 3, 4


class NoiseParams(ctypes.Structure):
    _fields_ = [("scale", ctypes.c_float), ("lacunarity", ctypes.c_float), ("persistence", ctypes.c_float),
                ("min", ctypes.c_float), ("max", ctypes.c_float), ("octaves", ctypes.c_int), ("seed", ctypes.c_int),
                ("warp", ctypes.c_float), ("mode", ctypes.c_int), ("blend", ctypes.c_int)]


NOISE_MODES = ["DEFAULT", "RIDGED", "BILLOWY", "IQNoise", "SWISS", "Jordan", "Plateaus"]
BLEND_MODES = ["replace", "add", "subtract", "multiply", "divide", "max", "min"]


def noise_params(mode="DEFAULT", blend="replace", **kw):
    """GradientNoiseMenu defaults (GradientNoise.h:39) with overrides by field name"""
    p = NoiseParams()
    lib().dmg_noise_params_default(ctypes.byref(p))
    p.mode, p.blend = NOISE_MODES.index(mode), BLEND_MODES.index(blend)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def noise_seed_offset(seed):
    out = (ctypes.c_float * 3)()
    check(lib().dmg_noise_seed_offset(int(seed), out))
    return np.array(out[:], np.float32)


def make_grid(W, H, coords=FULL_GLOBE, circumference=42000.0):
    g = Grid()
    g.W, g.H = int(W), int(H)
    for i in range(4):
        g.coords[i] = float(np.float32(coords[i]))
    g.circumference = float(circumference)
    return g


_lib = None


def lib():
    """Loads the library (raises if it was not built: run __graft_entry__.build()).

    In a process that also uses torch, import torch FIRST: torch ships its own libamdhip64 and the two must
    resolve to one HIP runtime (device pointers are shared with torch for the RCCL halo exchange)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FileNotFoundError("%s missing -- build it with `make -C demiurge_amd/csrc` "
                                "(or __graft_entry__.build()); there is no fallback path" % LIB_PATH)
    L = ctypes.CDLL(str(LIB_PATH))
    c_int, c_float, vp, sz = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
    PG, PP = ctypes.POINTER(Grid), ctypes.POINTER(OceanParams)
    sig = {
        "dmg_version": (c_int, []),
        "dmg_last_error": (ctypes.c_char_p, []),
        "dmg_runtime_init": (c_int, []),
        "dmg_debug_set": (c_int, [ctypes.c_char_p, ctypes.c_char_p]),
        "dmg_create": (c_int, [c_int, vp, c_int, ctypes.POINTER(vp)]),
        "dmg_destroy": (None, [vp]),
        "dmg_sync": (c_int, [vp]),
        "dmg_trim": (c_int, [vp]),
        "dmg_device_name": (c_int, [vp, ctypes.c_char_p, sz]),
        "dmg_copy_device": (c_int, [vp, vp, vp, sz]),
        "dmg_timer_start": (c_int, [vp]),
        "dmg_timer_stop": (c_int, [vp, ctypes.POINTER(c_float)]),
        "dmg_band_row0": (c_int, [c_int, c_int, c_int]),
        "dmg_ocean_params_default": (None, [PP]),
        "dmg_ocean_create": (c_int, [vp, PG, PP, c_int, c_int, c_int, ctypes.POINTER(vp)]),
        "dmg_ocean_destroy": (None, [vp]),
        "dmg_ocean_set_terrain": (c_int, [vp, vp]),
        "dmg_ocean_set_velocity": (c_int, [vp, vp, vp]),
        "dmg_ocean_set_divw": (c_int, [vp, vp]),
        "dmg_ocean_set_pressure": (c_int, [vp, vp]),
        "dmg_ocean_jacobi": (c_int, [vp, c_int, c_int]),
        "dmg_ocean_jacobi_sweeps": (c_int, [vp, c_int]),
        "dmg_ocean_zero_pressure": (c_int, [vp]),
        "dmg_ocean_jacobi_sweeps_begin": (c_int, [vp, c_int]),
        "dmg_ocean_jacobi_sweeps_end": (c_int, [vp]),
        "dmg_ocean_jacobi_geometry": (c_int, [vp, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
        "dmg_ocean_band_rows": (c_int, [vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
        "dmg_ocean_jacobi_launch": (c_int, [vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp]),
        "dmg_ocean_pressure_buffer": (c_int, [vp, ctypes.POINTER(c_int)]),
        "dmg_ocean_set_pressure_buffer": (c_int, [vp, c_int, c_int, c_int]),
        "dmg_ocean_pressure_halo": (c_int, [vp, c_int, c_int, c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(sz)]),
        "dmg_ocean_counters": (c_int, [vp, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]),
        "dmg_ocean_jacobi_kernel_name": (c_int, [vp, ctypes.c_char_p, sz]),
        "dmg_ocean_halo": (c_int, [vp, c_int, c_int, c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(sz)]),
        "dmg_ocean_step": (c_int, [vp, c_int]),
        "dmg_ocean_advect": (c_int, [vp]),
        "dmg_ocean_diffusion": (c_int, [vp, c_int]),
        "dmg_ocean_divergence": (c_int, [vp]),
        "dmg_ocean_subdiv": (c_int, [vp]),
        "dmg_ocean_advect_reach": (c_int, [vp, ctypes.POINTER(c_int)]),
        "dmg_ocean_advect_prepare": (c_int, [vp, c_int]),
        "dmg_ocean_get": (c_int, [vp, vp, vp, vp, vp]),
        "dmg_ocean_checkpoint": (c_int, [vp, ctypes.c_char_p]),
        "dmg_ocean_restore": (c_int, [vp, ctypes.c_char_p]),
        "dmg_blur": (c_int, [vp, PG, vp, c_float]),
        "dmg_temperature_steps": (c_int, [vp, PG, vp, vp, ctypes.POINTER(ClimateParams), c_int, c_int]),
        "dmg_climate_params_default": (None, [ctypes.POINTER(ClimateParams)]),
        "dmg_climate_set_params": (c_int, [vp, ctypes.POINTER(ClimateParams)]),
        "dmg_multi_climate_set_params": (c_int, [vp, ctypes.POINTER(ClimateParams)]),
        "dmg_erosion_pass": (c_int, [vp, PG, vp, vp, vp, c_float, c_float]),
        "dmg_flow_accumulate": (c_int, [vp, PG, vp, vp, c_float, c_float, c_int, vp]),
        "dmg_erosion_run": (c_int, [vp, PG, vp, c_int, c_int, c_float, c_float, c_float]),
        "dmg_flow_passes": (c_int, [vp, PG, vp, vp, c_float, vp, vp]),
        "dmg_climate_create": (c_int, [vp, PG, vp, vp, ctypes.POINTER(vp)]),
        "dmg_climate_destroy": (None, [vp]),
        "dmg_climate_steps": (c_int, [vp, c_int, c_int]),
        "dmg_climate_get": (c_int, [vp, vp]),
        "dmg_erosion_create": (c_int, [vp, PG, vp, c_int, c_float, c_float, c_float, c_int, ctypes.POINTER(vp)]),
        "dmg_erosion_destroy": (None, [vp]),
        "dmg_erosion_rounds": (c_int, [vp, c_int]),
        "dmg_erosion_get": (c_int, [vp, vp]),
        "dmg_thermal_erosion": (c_int, [vp, PG, vp]),
        "dmg_morphological": (c_int, [vp, PG, vp, c_float, c_int]),
        "dmg_morphological_gradient": (c_int, [vp, PG, vp, c_float]),
        "dmg_field_write_png": (c_int, [ctypes.c_char_p, c_int, c_int, vp, c_int]),
        "dmg_ocean_speed_shade": (c_int, [vp, c_float, vp]),
        "dmg_noise_params_default": (None, [ctypes.POINTER(NoiseParams)]),
        "dmg_noise_seed_offset": (c_int, [c_int, ctypes.POINTER(c_float)]),
        "dmg_div2_split": (c_int, [vp, c_float, ctypes.POINTER(c_float), ctypes.POINTER(c_float)]),
        "dmg_stream_units_probe": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp, sz,
                                           ctypes.POINTER(sz), ctypes.POINTER(c_int)]),
        "dmg_gradient_noise": (c_int, [vp, PG, vp, vp, ctypes.POINTER(NoiseParams)]),
        "dmg_ocean_generate_terrain": (c_int, [vp, ctypes.POINTER(NoiseParams)]),
        "dmg_tectonics_create": (c_int, [vp, PG, vp, ctypes.POINTER(vp)]),
        "dmg_tectonics_destroy": (None, [vp]),
        "dmg_tectonics_update_rotation": (c_int, [vp, c_int, c_float, ctypes.POINTER(c_float)]),
        "dmg_tectonics_step": (c_int, [vp, c_int]),
        "dmg_tectonics_fold": (c_int, [vp]),
        "dmg_tectonics_spread": (c_int, [vp]),
        "dmg_tectonics_render": (c_int, [vp]),
        "dmg_tectonics_unfold": (c_int, [vp]),
        "dmg_tectonics_get_terrain": (c_int, [vp, vp]),
        "dmg_tectonics_get_map": (c_int, [vp, vp]),
        "dmg_tectonics_plate_count": (c_int, [vp]),
        "dmg_tectonics_plate_size": (c_int, [vp, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
        "dmg_tectonics_get_plate": (c_int, [vp, c_int, vp]),
        "dmg_climate_create_band": (c_int, [vp, PG, vp, vp, c_int, c_int, c_int, ctypes.POINTER(vp)]),
        "dmg_climate_halo": (c_int, [vp, c_int, c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(sz)]),
        "dmg_climate_set_terrain": (c_int, [vp, vp]),
        "dmg_world_params_default": (None, [ctypes.POINTER(WorldParams)]),
        "dmg_world_create": (c_int, [vp, PG, vp, vp, ctypes.POINTER(vp)]),
        "dmg_world_destroy": (None, [vp]),
        "dmg_world_step": (c_int, [vp, ctypes.POINTER(WorldParams)]),
        "dmg_world_get": (c_int, [vp, vp, vp, vp]),
        "dmg_multi_world_create": (c_int, [c_int, ctypes.POINTER(c_int), PG, vp, vp, c_int, c_int, ctypes.POINTER(vp)]),
        "dmg_multi_world_create_rank": (c_int, [c_int, c_int, vp, c_int, PG, vp, vp, c_int, c_int, ctypes.POINTER(vp)]),
        "dmg_multi_world_destroy": (None, [vp]),
        "dmg_multi_world_step": (c_int, [vp, ctypes.POINTER(WorldParams)]),
        "dmg_multi_world_get": (c_int, [vp, vp, vp, vp]),
        "dmg_field_create": (c_int, [vp, PG, c_int, c_int, c_int, ctypes.POINTER(vp)]),
        "dmg_field_destroy": (None, [vp]),
        "dmg_field_set": (c_int, [vp, vp]),
        "dmg_field_get": (c_int, [vp, vp]),
        "dmg_field_halo": (c_int, [vp, c_int, c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(sz)]),
        "dmg_fields_exchange": (c_int, [ctypes.POINTER(vp), c_int, c_int]),
        "dmg_blur_schedule": (c_int, [c_float, ctypes.POINTER(c_float), c_int]),
        "dmg_field_blur_halo": (c_int, [c_float, c_int]),
        "dmg_field_blur_pass": (c_int, [vp, c_float, c_int]),
        "dmg_field_erosion_pass": (c_int, [vp, vp, vp, c_float, c_float]),
        "dmg_field_thermal_round": (c_int, [vp, vp]),
        "dmg_multi_create": (c_int, [c_int, ctypes.POINTER(c_int), PG, PP, c_int, c_int, ctypes.POINTER(vp)]),
        "dmg_multi_unique_id": (c_int, [vp]),
        "dmg_multi_create_rank": (c_int, [c_int, c_int, vp, c_int, PG, PP, c_int, c_int, ctypes.POINTER(vp)]),
        "dmg_multi_destroy": (None, [vp]),
        "dmg_multi_store_blob": (c_int, [vp, vp]),
        "dmg_multi_store_connect": (c_int, [vp, vp, c_int]),
        "dmg_multi_climate_store_blob": (c_int, [vp, vp]),
        "dmg_multi_climate_store_connect": (c_int, [vp, vp, c_int]),
        "dmg_multi_set_order": (c_int, [vp, c_int]),
        "dmg_multi_set_terrain": (c_int, [vp, vp]),
        "dmg_multi_generate_terrain": (c_int, [vp, ctypes.POINTER(NoiseParams)]),
        "dmg_multi_set_divw": (c_int, [vp, vp]),
        "dmg_multi_set_pressure": (c_int, [vp, vp]),
        "dmg_multi_set_velocity": (c_int, [vp, vp, vp]),
        "dmg_multi_get": (c_int, [vp, vp, vp, vp, vp]),
        "dmg_multi_sync": (c_int, [vp]),
        "dmg_multi_jacobi": (c_int, [vp, c_int, c_int]),
        "dmg_multi_step": (c_int, [vp, c_int]),
        "dmg_multi_exchange": (c_int, [vp, c_int, c_int]),
        "dmg_multi_band_count": (c_int, [vp]),
        "dmg_multi_band": (c_int, [vp, c_int, ctypes.POINTER(vp)]),
        "dmg_multi_get_stats": (c_int, [vp, ctypes.POINTER(MultiStats)]),
        "dmg_multi_climate_create": (c_int, [c_int, ctypes.POINTER(c_int), PG, vp, vp, c_int, c_int, ctypes.POINTER(vp)]),
        "dmg_multi_climate_create_rank": (c_int, [c_int, c_int, vp, c_int, PG, vp, vp, c_int, c_int, ctypes.POINTER(vp)]),
        "dmg_multi_climate_destroy": (None, [vp]),
        "dmg_multi_climate_steps": (c_int, [vp, c_int, c_int]),
        "dmg_multi_climate_set_terrain": (c_int, [vp, vp]),
        "dmg_multi_climate_get": (c_int, [vp, vp]),
        "dmg_multi_climate_sync": (c_int, [vp]),
        "dmg_multi_comm_ranks": (c_int, [vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
        "dmg_multi_timer_start": (c_int, [vp]),
        "dmg_multi_timer_stop": (c_int, [vp, ctypes.POINTER(c_float)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    L._dmg_signatures = sig
    # The library itself reads two environment variables (DMG_MULTI_TIMEOUT_MS, DMG_ST_TWIN: demiurge_hip.h).  Its test hooks and
    # experiment knobs come in through dmg_debug_set: THIS binding -- the harness of the tests, the bench and the tools, not the
    # product -- hands over the DMG_* variables of its own environment, once, before any handle exists.
    _lib = L
    if L.dmg_version() != VERSION:
        raise RuntimeError("%s is version %d, this binding was written for %d" % (LIB_PATH, L.dmg_version(), VERSION))
    if _is_harness():
        load_hooks_from_env()
    return L


VERSION = 101          # DMG_VERSION of include/demiurge_hip.h (tests/test_abi.py holds the two together)


def _is_harness():
    """the hooks reach the library only from the harness -- the tests, bench.py, the tools, __graft_entry__ (or any
    process started with DMG_TEST_HOOKS=1) -- not from whatever else imports this binding with DMG_* variables in its environment"""
    if os.environ.get("DMG_TEST_HOOKS") == "1" or "pytest" in sys.modules:
        return True
    main = getattr(sys.modules.get("__main__"), "__file__", None)
    if not main:
        return False
    p = Path(main).resolve()
    root = Path(__file__).resolve().parents[1]
    return root in p.parents and (p.parent.name in ("tools", "tests") or p.name in ("bench.py", "__graft_entry__.py"))


def load_hooks_from_env():
    """hand the DMG_* variables of this process's environment to dmg_debug_set (before the first handle: most hooks are read once)"""
    for k, v in os.environ.items():
        if k.startswith("DMG_") and k not in HARNESS_ENV:
            lib().dmg_debug_set(k.encode(), v.encode())


# read by the Python side (this file, bands.py, bench.py), not hooks of the library
HARNESS_ENV = {"DMG_TEST_HOOKS", "DMG_LIB", "DMG_BENCH_REHEARSAL", "DMG_HALO_OVERLAP", "DMG_HALO_BOUNCE", "DMG_HALO_PIPELINE", "DMG_MULTI_TIMEOUT_MS", "DMG_ST_TWIN"}


def debug_set(name, value):
    """dmg_debug_set: a test hook / experiment knob by name (value None removes it); most are read once, before the first handle"""
    check(lib().dmg_debug_set(name.encode(), None if value is None else str(value).encode()))


def check(rc):
    if rc != 0:
        raise DmgError(rc, lib().dmg_last_error().decode(errors="replace"))


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


class _Ptr(ctypes.c_void_p):
    """the address of a NumPy array, holding the array: `_p(_f32(x))` may have made a contiguous copy that nothing else
    refers to, and a bare integer address would let it be collected before the C call reads it"""


def _p(a):
    if a is None:
        return None
    r = _Ptr(a.ctypes.data)
    r._keep = a
    return r


class Context:
    """dmg_ctx: one per GPU/process.  stream=None: a private non-blocking stream; stream=<int hipStream_t>
    (0 = the default stream, e.g. torch.cuda.current_stream().cuda_stream): run on the host program's stream."""

    def __init__(self, device=0, stream=None):
        h = ctypes.c_void_p()
        if stream is None:
            check(lib().dmg_create(int(device), None, PRIVATE_STREAM, ctypes.byref(h)))
        else:
            check(lib().dmg_create(int(device), ctypes.c_void_p(int(stream)), 0, ctypes.byref(h)))
        self.h = h
        self._children = weakref.WeakSet()     # handles that must be destroyed before the context

    def close(self):
        if self.h:
            for c in list(self._children):
                c.close()
            lib().dmg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trim(self):
        """hand the cached temporaries of finished calls back to the device"""
        check(lib().dmg_trim(self.h))

    def sync(self):
        check(lib().dmg_sync(self.h))

    def div2_split(self, c):
        """(h, l) with x / c == fma(x, h, x * l): the streaming Jacobi kernel's two-operation divisor of the row constant c"""
        h, l = ctypes.c_float(), ctypes.c_float()
        check(lib().dmg_div2_split(self.h, ctypes.c_float(float(c)), ctypes.byref(h), ctypes.byref(l)))
        return np.float32(h.value), np.float32(l.value)

    def name(self):
        b = ctypes.create_string_buffer(160)
        check(lib().dmg_device_name(self.h, b, 160))
        return b.value.decode()

    def copy_device(self, dst, src, nbytes):
        check(lib().dmg_copy_device(self.h, dst, src, nbytes))

    def timer_start(self):
        check(lib().dmg_timer_start(self.h))

    def timer_stop(self):
        ms = ctypes.c_float()
        check(lib().dmg_timer_stop(self.h, ctypes.byref(ms)))
        return ms.value

    # generic filters ------------------------------------------------------
    def blur(self, grid, field, radius):
        f = _f32(field).copy()
        check(lib().dmg_blur(self.h, ctypes.byref(grid), _p(f), radius))
        return f

    def temperature_steps(self, grid, T, terrain, first_step, n_steps, params=None):
        T = _f32(T).copy()
        terrain = _f32(terrain)
        check(lib().dmg_temperature_steps(self.h, ctypes.byref(grid), _p(T), _p(terrain), ctypes.byref(params) if params is not None else None,
                                          first_step, n_steps))
        return T

    def erosion_pass(self, grid, h, flow, updrift, factor=1.0, slope_exponent=1.0):
        h = _f32(h).copy()
        flow, updrift = _f32(flow), _f32(updrift)
        check(lib().dmg_erosion_pass(self.h, ctypes.byref(grid), _p(h), _p(flow), _p(updrift), factor, slope_exponent))
        return h


def _ctx_flow_accumulate(self, grid, h, preblur=0.5, exponent=1.0, sel=None, lakes=0):
    h = _f32(h)
    sel = None if sel is None else _f32(sel)
    out = np.empty_like(h)
    check(lib().dmg_flow_accumulate(self.h, ctypes.byref(grid), _p(h), _p(sel), preblur, exponent, lakes, _p(out)))
    return out


def _ctx_flow_passes(self, grid, h, preblur=0.5, sel=None):
    """-> (direction codes, receiver mask) of FlowFilter::findMagicNumbers' two GLSL passes"""
    h = _f32(h)
    sel = None if sel is None else _f32(sel)
    codes, mask = np.empty_like(h), np.empty_like(h)
    check(lib().dmg_flow_passes(self.h, ctypes.byref(grid), _p(h), _p(sel), preblur, _p(codes), _p(mask)))
    return codes, mask


def _ctx_erosion_run(self, grid, h, N=50, iters=150, exponent=0.5, slope_exponent=1.0, factor=1.0):
    h = _f32(h).copy()
    check(lib().dmg_erosion_run(self.h, ctypes.byref(grid), _p(h), N, iters, exponent, slope_exponent, factor))
    return h


def write_png(path, field, clamp=True):
    """dmg_field_write_png: Project::file_write's 8-bit grey PNG (host-side; needs no GPU)"""
    f = _f32(field)
    check(lib().dmg_field_write_png(str(path).encode(), f.shape[1], f.shape[0], _p(f), 1 if clamp else 0))


def _ctx_gradient_noise(self, grid, field, params, sel=None):
    f = _f32(field).copy()
    sel = None if sel is None else _f32(sel)
    check(lib().dmg_gradient_noise(self.h, ctypes.byref(grid), _p(f), _p(sel), ctypes.byref(params)))
    return f


def _ctx_thermal_erosion(self, grid, h):
    h = _f32(h).copy()
    check(lib().dmg_thermal_erosion(self.h, ctypes.byref(grid), _p(h)))
    return h


def _ctx_morphological(self, grid, field, radius, op):
    f = _f32(field).copy()
    check(lib().dmg_morphological(self.h, ctypes.byref(grid), _p(f), radius, 1 if op in (1, "max", "dilate") else 0))
    return f


def _ctx_morphological_gradient(self, grid, field, radius):
    f = _f32(field).copy()
    check(lib().dmg_morphological_gradient(self.h, ctypes.byref(grid), _p(f), radius))
    return f


Context.thermal_erosion = _ctx_thermal_erosion
Context.gradient_noise = _ctx_gradient_noise
Context.morphological = _ctx_morphological
Context.morphological_gradient = _ctx_morphological_gradient
Context.flow_accumulate = _ctx_flow_accumulate
Context.erosion_run = _ctx_erosion_run
Context.flow_passes = _ctx_flow_passes


def climate_params(**kw):
    """the reference's values (Temperature.cpp), with the given fields replaced"""
    p = ClimateParams()
    lib().dmg_climate_params_default(ctypes.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class Climate:
    """dmg_climate: the climate field (Temperature.cpp) resident on the device between step calls.  T None: the field starts
    at params.initial_temperature (the reference clears the layer)."""

    def __init__(self, ctx, grid, T, terrain, band_rank=0, band_count=1, ghost=0, params=None):
        terrain = _f32(terrain)
        T = _f32(T) if T is not None else None
        self.ctx, self.shape = ctx, terrain.shape
        h = ctypes.c_void_p()
        check(lib().dmg_climate_create_band(ctx.h, ctypes.byref(grid), _p(T), _p(terrain), band_rank, band_count, ghost, ctypes.byref(h)))
        self.h = h
        if params is not None:
            self.set_params(params)

    def set_params(self, params):
        check(lib().dmg_climate_set_params(self.h, ctypes.byref(params)))

    def steps(self, first_step, n_steps):
        check(lib().dmg_climate_steps(self.h, first_step, n_steps))

    def halo(self, side, rows):
        send, recv, nbytes = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t()
        check(lib().dmg_climate_halo(self.h, side, rows, ctypes.byref(send), ctypes.byref(recv), ctypes.byref(nbytes)))
        return send.value, recv.value, nbytes.value

    def get(self, out=None):
        if out is not None:                                    # a band: its owned rows into the caller's full-grid array
            check(lib().dmg_climate_get(self.h, _p(out)))
            return out
        out = np.empty(self.shape, np.float32)
        check(lib().dmg_climate_get(self.h, _p(out)))
        return out

    def close(self):
        if self.h:
            lib().dmg_climate_destroy(self.h)
            self.h = None


class MultiClimate:
    """dmg_multi_climate: the climate year cycle over latitude bands driven by the library (threads, streams and
    transports of dmg_multi).  devices=[...]: one process drives the bands; rank/world/unique_id: one process per GPU."""

    def __init__(self, grid, T, terrain, devices=None, ghost=8, flags=MULTI_AUTO, rank=None, world=None, unique_id=None, device=0, params=None):
        T, terrain = _f32(T), _f32(terrain)
        self.shape = T.shape
        h = ctypes.c_void_p()
        if devices is not None:
            ids = (ctypes.c_int * len(devices))(*devices)
            check(lib().dmg_multi_climate_create(len(devices), ids, ctypes.byref(grid), _p(T), _p(terrain), ghost, flags, ctypes.byref(h)))
        else:
            buf = ctypes.create_string_buffer(unique_id, 128) if unique_id is not None else None
            check(lib().dmg_multi_climate_create_rank(rank, world, buf, device, ctypes.byref(grid), _p(T), _p(terrain), ghost, flags, ctypes.byref(h)))
        self.h = h
        if params is not None:
            self.set_params(params)

    def set_params(self, params):
        check(lib().dmg_multi_climate_set_params(self.h, ctypes.byref(params)))

    def steps(self, first_step, n_steps):
        check(lib().dmg_multi_climate_steps(self.h, first_step, n_steps))

    def set_terrain(self, terrain):
        t = _f32(terrain)                                      # (held until the call returns: _p hands out a bare address)
        check(lib().dmg_multi_climate_set_terrain(self.h, _p(t)))

    def sync(self):
        check(lib().dmg_multi_climate_sync(self.h))

    def get(self, out=None):
        if out is None:
            out = np.zeros(self.shape, np.float32)
        check(lib().dmg_multi_climate_get(self.h, _p(out)))
        return out

    def close(self):
        if self.h:
            lib().dmg_multi_climate_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Erosion:
    """dmg_erosion: cpufilter::run round by round, heights resident on the device"""

    def __init__(self, ctx, grid, h, N=50, exponent=0.5, slope_exponent=1.0, factor=1.0, lakes=0):
        h = _f32(h)
        self.ctx, self.shape = ctx, h.shape
        e = ctypes.c_void_p()
        check(lib().dmg_erosion_create(ctx.h, ctypes.byref(grid), _p(h), N, exponent, slope_exponent, factor, lakes, ctypes.byref(e)))
        self.h = e

    def rounds(self, n):
        check(lib().dmg_erosion_rounds(self.h, n))

    def get(self):
        out = np.empty(self.shape, np.float32)
        check(lib().dmg_erosion_get(self.h, _p(out)))
        return out

    def close(self):
        if self.h:
            lib().dmg_erosion_destroy(self.h)
            self.h = None


def stream_units_probe(W, H, rowclass, lo, hi, y0=0, y1=None, gs=0, gn=0, part=0, zone_lo=0, zone_hi=0, num_cu=256):
    """Test hook (no GPU): the streaming kernel's unit list for local rows [lo, hi) of a band, from a row-class table
    rowclass[strip, local row] (uint8: 0 open water, 1 land everywhere, 2 mixed).  -> (int32 array [n, 4], twin flag)"""
    y1 = H if y1 is None else y1
    rc = np.ascontiguousarray(rowclass, dtype=np.uint8)
    L = gs + (y1 - y0) + gn
    assert rc.shape == ((W + 239) // 240, L), (rc.shape, L)
    n, twin = ctypes.c_size_t(0), ctypes.c_int(0)
    cap = 65536
    out = np.zeros((cap, 4), np.int32)
    check(lib().dmg_stream_units_probe(W, H, y0, y1, gs, gn, rc.ctypes.data_as(ctypes.c_void_p), num_cu, lo, hi, part, zone_lo, zone_hi,
                                       out.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(n), ctypes.byref(twin)))
    return out[:n.value].copy(), twin.value


def default_ocean_params():
    p = OceanParams()
    lib().dmg_ocean_params_default(ctypes.byref(p))
    return p


class Ocean:
    """dmg_ocean: the OceanCurrents filter state of one latitude band."""

    def __init__(self, ctx, grid, params=None, band_rank=0, band_count=1, ghost=1):
        self.ctx, self.grid = ctx, grid
        self.params = params if params is not None else default_ocean_params()
        self.rank, self.count, self.ghost = band_rank, band_count, ghost
        h = ctypes.c_void_p()
        check(lib().dmg_ocean_create(ctx.h, ctypes.byref(grid), ctypes.byref(self.params), band_rank, band_count,
                                     ghost, ctypes.byref(h)))
        self.h = h
        ctx._children.add(self)

    def close(self):
        if self.h:
            if not getattr(self, "borrowed", False):       # a band borrowed from a Multi belongs to it
                lib().dmg_ocean_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def speed_shade(self, land_value=0.5):
        out = np.zeros((self.grid.H, self.grid.W), np.float32)
        check(lib().dmg_ocean_speed_shade(self.h, land_value, _p(out)))
        return out

    def generate_terrain(self, params):
        """dmg_ocean_generate_terrain: gradient-noise terrain made on the device, band-local rows only"""
        check(lib().dmg_ocean_generate_terrain(self.h, ctypes.byref(params)))

    def set_terrain(self, img):
        img = _f32(img)
        assert img.shape == (self.grid.H, self.grid.W)
        check(lib().dmg_ocean_set_terrain(self.h, _p(img)))

    def set_divw(self, divw):
        divw = _f32(divw)
        assert divw.shape == (self.grid.H, self.grid.W)
        check(lib().dmg_ocean_set_divw(self.h, _p(divw)))

    def set_pressure(self, p):
        p = _f32(p)
        assert p.shape == (self.grid.H, self.grid.W)
        check(lib().dmg_ocean_set_pressure(self.h, _p(p)))

    def set_velocity(self, v):
        """v: (H, W, 2) interleaved like the reference's RG32F texture, or a (vx, vy) pair of planes."""
        if isinstance(v, (tuple, list)):
            vx, vy = _f32(v[0]), _f32(v[1])
        else:
            v = np.asarray(v, np.float32)
            vx, vy = _f32(v[..., 0]), _f32(v[..., 1])
        check(lib().dmg_ocean_set_velocity(self.h, _p(vx), _p(vy)))

    def zero_pressure(self):
        check(lib().dmg_ocean_zero_pressure(self.h))

    def jacobi(self, iters, zero_first=True):
        check(lib().dmg_ocean_jacobi(self.h, int(iters), 1 if zero_first else 0))

    def jacobi_sweeps(self, n):
        check(lib().dmg_ocean_jacobi_sweeps(self.h, int(n)))

    def jacobi_sweeps_begin(self, n):
        """all of a block of n sweeps but the interior tiles of its last launch; halo() then serves the new iterate"""
        check(lib().dmg_ocean_jacobi_sweeps_begin(self.h, int(n)))

    def jacobi_sweeps_end(self):
        check(lib().dmg_ocean_jacobi_sweeps_end(self.h))

    # ---- pieces for a self-scheduled block (bands.jacobi_solve_pipelined) ----
    def jacobi_geometry(self, k):
        """-> (lo, hi, tile_rows) of the 8-sweep launch ending k sweeps after the last halo refresh"""
        lo, hi, tr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().dmg_ocean_jacobi_geometry(self.h, int(k), ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(tr)))
        return lo.value, hi.value, tr.value

    def band_rows(self):
        """-> (stored rows L, ghost rows south, ghost rows north)"""
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().dmg_ocean_band_rows(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def jacobi_launch(self, lo, hi, n, part, zone_lo, zone_hi, src, stream=None):
        check(lib().dmg_ocean_jacobi_launch(self.h, lo, hi, n, part, zone_lo, zone_hi, src, ctypes.c_void_p(stream or 0)))

    def pressure_buffer(self):
        c = ctypes.c_int()
        check(lib().dmg_ocean_pressure_buffer(self.h, ctypes.byref(c)))
        return c.value

    def set_pressure_buffer(self, idx, launches=0, sweeps=0):
        check(lib().dmg_ocean_set_pressure_buffer(self.h, idx, launches, sweeps))

    def pressure_halo(self, buffer, side, rows):
        s, r, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t()
        check(lib().dmg_ocean_pressure_halo(self.h, buffer, side, rows, ctypes.byref(s), ctypes.byref(r), ctypes.byref(n)))
        return s.value, r.value, n.value

    def counters(self):
        """-> (jacobi kernel launches, jacobi sweeps) since creation"""
        a, b = ctypes.c_longlong(), ctypes.c_longlong()
        check(lib().dmg_ocean_counters(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def kernel_name(self):
        buf = ctypes.create_string_buffer(64)
        check(lib().dmg_ocean_jacobi_kernel_name(self.h, buf, 64))
        return buf.value.decode()

    def halo(self, field, side, rows):
        """-> (send_ptr, recv_ptr, nbytes) device addresses (ints); nbytes == 0 at a grid edge."""
        s, r, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t()
        check(lib().dmg_ocean_halo(self.h, field, side, rows, ctypes.byref(s), ctypes.byref(r), ctypes.byref(n)))
        return s.value, r.value, n.value

    def step(self, n_outer=1):
        check(lib().dmg_ocean_step(self.h, int(n_outer)))

    def advect(self):
        check(lib().dmg_ocean_advect(self.h))

    def diffusion(self, iters=None):
        check(lib().dmg_ocean_diffusion(self.h, int(self.params.diffusion_iters if iters is None else iters)))

    def divergence(self):
        check(lib().dmg_ocean_divergence(self.h))

    def subdiv(self):
        check(lib().dmg_ocean_subdiv(self.h))

    def advect_reach(self):
        r = ctypes.c_int()
        check(lib().dmg_ocean_advect_reach(self.h, ctypes.byref(r)))
        return r.value

    def advect_prepare(self, reach):
        check(lib().dmg_ocean_advect_prepare(self.h, reach))

    def checkpoint(self, prefix):
        check(lib().dmg_ocean_checkpoint(self.h, str(prefix).encode()))

    def restore(self, prefix):
        check(lib().dmg_ocean_restore(self.h, str(prefix).encode()))

    def get(self, want=("v", "pressure", "divw"), out=None):
        """Downloads owned rows into full-grid arrays (zeros elsewhere).  Returns dict of arrays."""
        H, W = self.grid.H, self.grid.W
        res = {} if out is None else out
        arr = {}
        for k in ("vx", "vy", "pressure", "divw"):
            need = (k in want) or (k in ("vx", "vy") and "v" in want)
            arr[k] = res.get(k, np.zeros((H, W), np.float32)) if need else None
        check(lib().dmg_ocean_get(self.h, _p(arr["vx"]), _p(arr["vy"]), _p(arr["pressure"]), _p(arr["divw"])))
        for k, a in arr.items():
            if a is not None:
                res[k] = a
        if "v" in want:
            res["v"] = np.stack([arr["vx"], arr["vy"]], axis=-1)
        return res


class World:
    """dmg_world: BASELINE config 5, tectonic uplift -> erosion -> climate on one terrain"""

    def __init__(self, ctx, grid, terrain, T0):
        self.ctx, self.grid = ctx, grid
        h = ctypes.c_void_p()
        check(lib().dmg_world_create(ctx.h, ctypes.byref(grid), _p(_f32(terrain)), _p(_f32(T0)), ctypes.byref(h)))
        self.h = h
        ctx._children.add(self)

    def close(self):
        if self.h:
            lib().dmg_world_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def default_params():
        p = WorldParams()
        lib().dmg_world_params_default(ctypes.byref(p))
        return p

    def step(self, params=None):
        check(lib().dmg_world_step(self.h, None if params is None else ctypes.byref(params)))

    def get(self):
        H, W = self.grid.H, self.grid.W
        t, e, T = (np.empty((H, W), np.float32) for _ in range(3))
        check(lib().dmg_world_get(self.h, _p(t), _p(e), _p(T)))
        return dict(terrain=t, eroded=e, T=T)


class MultiWorld:
    """dmg_multi_world: BASELINE config 5 on N GPUs -- tectonics and erosion on the first band's GPU (or as replicas, one process per
    GPU), the climate year cycle in latitude bands"""

    def __init__(self, grid, terrain, T0, devices=None, ghost=8, flags=MULTI_AUTO, rank=None, world=None, unique_id=None, device=0):
        self.grid = grid
        terrain, T0 = _f32(terrain), _f32(T0)
        h = ctypes.c_void_p()
        if devices is not None:
            ids = (ctypes.c_int * len(devices))(*devices)
            check(lib().dmg_multi_world_create(len(devices), ids, ctypes.byref(grid), _p(terrain), _p(T0), ghost, flags, ctypes.byref(h)))
        else:
            buf = ctypes.create_string_buffer(unique_id, 128) if unique_id is not None else None
            check(lib().dmg_multi_world_create_rank(rank, world, buf, device, ctypes.byref(grid), _p(terrain), _p(T0), ghost, flags, ctypes.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib().dmg_multi_world_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, params=None):
        check(lib().dmg_multi_world_step(self.h, ctypes.byref(params) if params is not None else None))

    def get(self):
        H, W = self.grid.H, self.grid.W
        out = {k: np.zeros((H, W), np.float32) for k in ("terrain", "eroded", "T")}
        check(lib().dmg_multi_world_get(self.h, _p(out["terrain"]), _p(out["eroded"]), _p(out["T"])))
        return out


class Field:
    """dmg_field: one scalar field of one latitude band on the device (ghost rows towards the neighbouring bands)."""

    def __init__(self, ctx, grid, band_rank=0, band_count=1, ghost=0, data=None):
        self.ctx, self.grid = ctx, grid
        h = ctypes.c_void_p()
        check(lib().dmg_field_create(ctx.h, ctypes.byref(grid), band_rank, band_count, ghost, ctypes.byref(h)))
        self.h = h
        ctx._children.add(self)
        if data is not None:
            self.set(data)

    def close(self):
        if self.h:
            lib().dmg_field_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set(self, full):
        check(lib().dmg_field_set(self.h, _p(_f32(full))))

    def get(self, out):
        check(lib().dmg_field_get(self.h, _p(out)))
        return out

    def blur_pass(self, r, vertical):
        check(lib().dmg_field_blur_pass(self.h, r, 1 if vertical else 0))

    def erosion_pass(self, flow, updrift, factor=1.0, slope_exponent=1.0):
        check(lib().dmg_field_erosion_pass(self.h, flow.h, updrift.h, factor, slope_exponent))

    def thermal_round(self, loss):
        check(lib().dmg_field_thermal_round(self.h, loss.h))


def fields_exchange(bands, rows):
    arr = (ctypes.c_void_p * len(bands))(*[b.h for b in bands])
    check(lib().dmg_fields_exchange(arr, len(bands), rows))


def blur_schedule(radius):
    buf = (ctypes.c_float * 64)()
    n = lib().dmg_blur_schedule(radius, buf, 64)
    if n < 0:
        check(n)
    return [buf[i] for i in range(n)]


def blur_halo(r, vertical):
    return lib().dmg_field_blur_halo(r, 1 if vertical else 0)


class Multi:
    """dmg_multi: one ocean over several bands / GPUs, driven by the library (RCCL or peer-copy halo exchange).
    Multi(grid, devices=[0, 1, ...]) = one process; Multi(grid, rank=r, world=n, unique_id=..., device=d) = one process per GPU."""

    def __init__(self, grid, devices=None, params=None, ghost=32, flags=MULTI_AUTO, rank=None, world=None, unique_id=None, device=0):
        self.grid, self.params = grid, params if params is not None else default_ocean_params()
        h = ctypes.c_void_p()
        if rank is None:
            devs = (ctypes.c_int * len(devices))(*devices)
            check(lib().dmg_multi_create(len(devices), devs, ctypes.byref(grid), ctypes.byref(self.params), ghost, flags, ctypes.byref(h)))
        else:
            idbuf = ctypes.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
            check(lib().dmg_multi_create_rank(rank, world, idbuf, device, ctypes.byref(grid), ctypes.byref(self.params), ghost, flags,
                                              ctypes.byref(h)))
        self.h = h

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        check(lib().dmg_multi_unique_id(buf))
        return buf.raw

    def store_blob(self):
        """MULTI_STORE, one process per GPU, no RCCL id: this rank's memory handles (all-gather them, then store_connect)"""
        buf = ctypes.create_string_buffer(MULTI_STORE_BLOB_BYTES)
        check(lib().dmg_multi_store_blob(self.h, buf))
        return buf.raw

    def store_connect(self, blobs):
        """blobs: the store_blob() bytes of every rank, in rank order"""
        raw = b"".join(blobs)
        check(lib().dmg_multi_store_connect(self.h, ctypes.create_string_buffer(raw, len(raw)), len(blobs)))

    def close(self):
        if self.h:
            lib().dmg_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_order(self, order):
        check(lib().dmg_multi_set_order(self.h, order))

    def set_terrain(self, img):
        check(lib().dmg_multi_set_terrain(self.h, _p(_f32(img))))

    def set_divw(self, b):
        check(lib().dmg_multi_set_divw(self.h, _p(_f32(b))))

    def set_pressure(self, p):
        check(lib().dmg_multi_set_pressure(self.h, _p(_f32(p))))

    def set_velocity(self, v):
        v = _f32(v)
        vx, vy = np.ascontiguousarray(v[..., 0]), np.ascontiguousarray(v[..., 1])
        check(lib().dmg_multi_set_velocity(self.h, _p(vx), _p(vy)))

    def jacobi(self, iters, zero_first=True):
        check(lib().dmg_multi_jacobi(self.h, iters, 1 if zero_first else 0))

    def step(self, n_outer=1):
        check(lib().dmg_multi_step(self.h, n_outer))

    def sync(self):
        check(lib().dmg_multi_sync(self.h))

    def get(self, want=("v", "pressure", "divw"), out=None):
        H, W = self.grid.H, self.grid.W
        out = {} if out is None else out
        arr = lambda k: out.setdefault(k, np.zeros((H, W), np.float32))
        vx = arr("vx") if "v" in want else None
        vy = arr("vy") if "v" in want else None
        p = arr("pressure") if "pressure" in want else None
        d = arr("divw") if "divw" in want else None
        check(lib().dmg_multi_get(self.h, _p(vx), _p(vy), _p(p), _p(d)))
        if "v" in want:
            out["v"] = np.stack([out["vx"], out["vy"]], -1)
        return out

    def band(self, i=0):
        """borrowed Ocean view of local band i (do not close it)"""
        h = ctypes.c_void_p()
        check(lib().dmg_multi_band(self.h, i, ctypes.byref(h)))
        o = Ocean.__new__(Ocean)
        o.h, o.ctx, o.grid, o.params, o.borrowed = h, None, self.grid, self.params, True
        return o

    def stats(self):
        s = MultiStats()
        check(lib().dmg_multi_get_stats(self.h, ctypes.byref(s)))
        return {k: getattr(s, k) for k, _ in MultiStats._fields_}

    def comm_ranks(self):
        """(ranks RCCL reports for the data communicator, distinct devices among this process's bands)"""
        r, d = ctypes.c_int(), ctypes.c_int()
        check(lib().dmg_multi_comm_ranks(self.h, ctypes.byref(r), ctypes.byref(d)))
        return r.value, d.value

    def timer_start(self):
        check(lib().dmg_multi_timer_start(self.h))

    def timer_stop(self):
        ms = ctypes.c_float()
        check(lib().dmg_multi_timer_stop(self.h, ctypes.byref(ms)))
        return ms.value


class Tectonics:
    """dmg_tectonics: the Tectonics filter state (plates + global map)."""

    def __init__(self, ctx, grid, terrain):
        self.ctx, self.grid = ctx, grid
        terrain = _f32(terrain)
        assert terrain.shape == (grid.H, grid.W)
        h = ctypes.c_void_p()
        check(lib().dmg_tectonics_create(ctx.h, ctypes.byref(grid), _p(terrain), ctypes.byref(h)))
        self.h = h
        ctx._children.add(self)

    def close(self):
        if self.h:
            lib().dmg_tectonics_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update_rotation(self, plate, theta, axis):
        ax = (ctypes.c_float * 3)(*[float(a) for a in axis])
        check(lib().dmg_tectonics_update_rotation(self.h, plate, theta, ax))

    def step(self, n=1):
        check(lib().dmg_tectonics_step(self.h, n))

    def fold(self):
        check(lib().dmg_tectonics_fold(self.h))

    def spread(self):
        check(lib().dmg_tectonics_spread(self.h))

    def render(self):
        check(lib().dmg_tectonics_render(self.h))

    def unfold(self):
        check(lib().dmg_tectonics_unfold(self.h))

    def terrain(self):
        out = np.empty((self.grid.H, self.grid.W), np.float32)
        check(lib().dmg_tectonics_get_terrain(self.h, _p(out)))
        return out

    def map(self):
        out = np.empty((self.grid.W, self.grid.W, 4), np.float32)
        check(lib().dmg_tectonics_get_map(self.h, _p(out)))
        return out

    def plates(self):
        res = []
        for k in range(lib().dmg_tectonics_plate_count(self.h)):
            w, h = ctypes.c_int(), ctypes.c_int()
            check(lib().dmg_tectonics_plate_size(self.h, k, ctypes.byref(w), ctypes.byref(h)))
            out = np.empty((h.value, w.value, 4), np.float32)
            check(lib().dmg_tectonics_get_plate(self.h, k, _p(out)))
            res.append(out)
        return res
