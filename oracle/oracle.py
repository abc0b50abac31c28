"""ctypes binding of oracle/liboracle.so (the CPU restatement).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from
the product package."""
import ctypes
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "liboracle.so"
FULL_GLOBE = (np.float32(-np.pi / 2), np.float32(np.pi / 2), np.float32(-np.pi), np.float32(np.pi))  # Project.h:97


class Grid(ctypes.Structure):
    _fields_ = [("W", ctypes.c_int), ("H", ctypes.c_int), ("coords", ctypes.c_float * 4),
                ("circumference", ctypes.c_float)]


def grid(W, H, coords=FULL_GLOBE, circumference=42000.0):
    g = Grid()
    g.W, g.H = int(W), int(H)
    for i in range(4):
        g.coords[i] = float(np.float32(coords[i]))
    g.circumference = float(circumference)
    return g


_lib = None
os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # shared hosts: do not spin at barriers


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(str(LIB_PATH))
        P = ctypes.c_void_p
        G = ctypes.POINTER(Grid)
        L.orc_set_linear_taps.argtypes = [ctypes.c_int]
        L.orc_set_threads.argtypes = [ctypes.c_int]
        L.orc_ocean_jacobi.argtypes = [G, P, P, P, P, ctypes.c_int]
        L.orc_ocean_jacobi_rows.argtypes = [G, P, P, P, P, ctypes.c_int, ctypes.c_int]
        L.orc_ocean_divergence.argtypes = [G, P, P, ctypes.c_float, P]
        L.orc_ocean_diffusion.argtypes = [G, P, P, P, ctypes.c_int, ctypes.c_int]
        L.orc_ocean_advect.argtypes = [G, P, P, P]
        L.orc_ocean_subdiv.argtypes = [G, P, P, P, ctypes.c_float, P]
        L.orc_jacobi_row_tables.argtypes = [G, P, P, P]
        L.orc_pole_columns.argtypes = [G, P]
        L.orc_blur_rlist.argtypes = [ctypes.c_float, P, ctypes.c_int]
        L.orc_blur_rlist.restype = ctypes.c_int
        L.orc_blur13_pass.argtypes = [G, P, ctypes.c_float, ctypes.c_float, P]
        L.orc_blur.argtypes = [G, P, P, ctypes.c_float]
        L.orc_temperature.argtypes = [G, P, P, P, ctypes.c_int, ctypes.c_int]
        L.orc_erosion_pass.argtypes = [G, P, P, P, ctypes.c_float, ctypes.c_float, P]
        L.orc_thermal_erosion.argtypes = [G, P, P, P]
        L.orc_morph_rlist.argtypes = [ctypes.c_float, P, ctypes.c_int]
        L.orc_morph_rlist.restype = ctypes.c_int
        L.orc_morphological.argtypes = [G, P, P, ctypes.c_float, ctypes.c_int]
        L.orc_flow_direction.argtypes = [G, P, P, P]
        L.orc_flow_mask.argtypes = [G, P, P, P, P]
        L.orc_lake_adjacency.argtypes = [G, P, P]
        L.orc_flow_accumulate.argtypes = [G, P, P, ctypes.c_float, ctypes.c_float, P, P, P, P]
        L.orc_flow_accumulate_lakes.argtypes = [G, P, P, ctypes.c_float, ctypes.c_float, P]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data


def set_linear_taps(bilinear):
    lib().orc_set_linear_taps(1 if bilinear else 0)


def set_threads(n):
    lib().orc_set_threads(int(n))


def ocean_jacobi(g, img, divw, iters, p0=None, rows=False, nthreads=0):
    img, divw = _f32(img), _f32(divw)
    p = np.zeros((g.H, g.W), np.float32) if p0 is None else _f32(p0).copy()
    s = np.empty_like(p)
    if rows:
        lib().orc_ocean_jacobi_rows(ctypes.byref(g), _ptr(img), _ptr(divw), _ptr(p), _ptr(s), iters, nthreads)
    else:
        lib().orc_ocean_jacobi(ctypes.byref(g), _ptr(img), _ptr(divw), _ptr(p), _ptr(s), iters)
    return p


def ocean_divergence(g, img, v, pressurefactor=100.0):
    img, v = _f32(img), _f32(v)
    out = np.empty((g.H, g.W), np.float32)
    lib().orc_ocean_divergence(ctypes.byref(g), _ptr(img), _ptr(v), pressurefactor, _ptr(out))
    return out


def ocean_diffusion(g, img, v, iters=50, scalar_rhs=True):
    img, v = _f32(img), _f32(v).copy()
    s = np.empty_like(v)
    lib().orc_ocean_diffusion(ctypes.byref(g), _ptr(img), _ptr(v), _ptr(s), iters, 1 if scalar_rhs else 0)
    return v


def ocean_advect(g, img, v):
    img, v = _f32(img), _f32(v)
    out = np.empty((g.H, g.W, 2), np.float32)
    lib().orc_ocean_advect(ctypes.byref(g), _ptr(img), _ptr(v), _ptr(out))
    return out


def ocean_subdiv(g, img, v, pressure, pressurefactor=100.0):
    img, v, pressure = _f32(img), _f32(v), _f32(pressure)
    out = np.empty((g.H, g.W, 2), np.float32)
    lib().orc_ocean_subdiv(ctypes.byref(g), _ptr(img), _ptr(v), _ptr(pressure), pressurefactor, _ptr(out))
    return out


def ocean_step(g, img, v, n_outer=1, jacobi_iters=5000, diffusion_iters=50, pressurefactor=100.0, rows=False):
    """OceanCurrents::run body (OceanCurrents.cpp:155-209) minus the WIP resize/while(true)."""
    p = divw = None
    for _ in range(n_outer):
        v = ocean_advect(g, img, v)
        v = ocean_diffusion(g, img, v, diffusion_iters)
        divw = ocean_divergence(g, img, v, pressurefactor)
        p = ocean_jacobi(g, img, divw, jacobi_iters, rows=rows)
        v = ocean_subdiv(g, img, v, p, pressurefactor)
    return v, p, divw


def jacobi_row_tables(g):
    cx, cy, be = (np.empty(g.H, np.float32) for _ in range(3))
    lib().orc_jacobi_row_tables(ctypes.byref(g), _ptr(cx), _ptr(cy), _ptr(be))
    return cx, cy, be


def pole_columns(g):
    pc = np.empty(g.W, np.int32)
    lib().orc_pole_columns(ctypes.byref(g), _ptr(pc))
    return pc


def blur_rlist(radius):
    buf = np.empty(4096, np.float32)
    n = lib().orc_blur_rlist(radius, _ptr(buf), 4096)
    return buf[:n].copy()


def blur13_pass(g, src, dirx, diry):
    src = _f32(src)
    out = np.empty_like(src)
    lib().orc_blur13_pass(ctypes.byref(g), _ptr(src), dirx, diry, _ptr(out))
    return out


def blur(g, field, radius):
    f = _f32(field).copy()
    s = np.empty_like(f)
    lib().orc_blur(ctypes.byref(g), _ptr(f), _ptr(s), radius)
    return f


def temperature(g, T, terrain, first_step, n_steps):
    T, terrain = _f32(T).copy(), _f32(terrain)
    s = np.empty_like(T)
    lib().orc_temperature(ctypes.byref(g), _ptr(T), _ptr(s), _ptr(terrain), first_step, n_steps)
    return T


def erosion_pass(g, h, flow, updrift, factor=1.0, slope_exponent=1.0):
    h, flow, updrift = _f32(h), _f32(flow), _f32(updrift)
    out = np.empty_like(h)
    lib().orc_erosion_pass(ctypes.byref(g), _ptr(h), _ptr(flow), _ptr(updrift), factor, slope_exponent, _ptr(out))
    return out


def flow_direction(g, img, sel=None):
    img = _f32(img)
    sel = np.ones_like(img) if sel is None else _f32(sel)
    out = np.empty_like(img)
    lib().orc_flow_direction(ctypes.byref(g), _ptr(img), _ptr(sel), _ptr(out))
    return out


def flow_mask(g, img, codes, sel=None):
    img, codes = _f32(img), _f32(codes)
    sel = np.ones_like(img) if sel is None else _f32(sel)
    out = np.empty_like(img)
    lib().orc_flow_mask(ctypes.byref(g), _ptr(img), _ptr(sel), _ptr(codes), _ptr(out))
    return out


def lake_adjacency(g, lakeid):
    lakeid = _f32(lakeid)
    out = np.empty_like(lakeid)
    lib().orc_lake_adjacency(ctypes.byref(g), _ptr(lakeid), _ptr(out))
    return out


def flow_accumulate(g, h, preblur=0.5, exponent=1.0, sel=None):
    """-> dict(flow, codes, mask, label)"""
    h = _f32(h)
    sel = None if sel is None else _f32(sel)
    flow, codes, mask = (np.empty_like(h) for _ in range(3))
    label = np.empty(h.shape, np.int32)
    lib().orc_flow_accumulate(ctypes.byref(g), _ptr(h), None if sel is None else _ptr(sel), preblur, exponent,
                              _ptr(flow), _ptr(codes), _ptr(mask), _ptr(label))
    return dict(flow=flow, codes=codes, mask=mask, label=label)


def flow_accumulate_lakes(g, h, preblur=0.5, exponent=1.0, sel=None):
    h = _f32(h)
    sel = None if sel is None else _f32(sel)
    flow = np.empty_like(h)
    lib().orc_flow_accumulate_lakes(ctypes.byref(g), _ptr(h), None if sel is None else _ptr(sel), preblur, exponent, _ptr(flow))
    return flow


def thermal_erosion(g, h):
    h = _f32(h).copy()
    a, b = np.empty_like(h), np.empty_like(h)
    lib().orc_thermal_erosion(ctypes.byref(g), _ptr(h), _ptr(a), _ptr(b))
    return h


def morph_rlist(radius):
    buf = np.empty(64, np.float32)
    n = lib().orc_morph_rlist(radius, _ptr(buf), 64)
    return buf[:n].copy()


def morphological(g, field, radius, op):
    f = _f32(field).copy()
    s = np.empty_like(f)
    lib().orc_morphological(ctypes.byref(g), _ptr(f), _ptr(s), radius, 1 if op == "max" else 0)
    return f


def morphological_gradient(g, field, radius):
    er = morphological(g, field, np.float32(radius) / np.float32(2), "min")
    di = morphological(g, field, (np.float32(radius) + np.float32(1)) / np.float32(2), "max")
    return (di - er).astype(np.float32)
