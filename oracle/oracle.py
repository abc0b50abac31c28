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
        L.orc_set_raster_st.argtypes = [ctypes.c_int]
        L.orc_set_glsl_math.argtypes = [ctypes.c_int]
        L.orc_ocean_jacobi.argtypes = [G, P, P, P, P, ctypes.c_int]
        L.orc_ocean_jacobi_rows.argtypes = [G, P, P, P, P, ctypes.c_int, ctypes.c_int]
        L.orc_ocean_jacobi_rows_form.argtypes = [G, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_jacobi_fused_tables.argtypes = [G, ctypes.c_int, P]
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
        L.orc_temperature_p.argtypes = [G, P, P, P, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.orc_climate_params_default.argtypes = [ctypes.c_void_p]
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
        L.orc_flow_accumulate_flags.argtypes = [G, P, P, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, P]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data


def set_linear_taps(bilinear):
    lib().orc_set_linear_taps(1 if bilinear else 0)


def set_raster_st(on):
    """literal Jacobi form only: take `st` as llvmpipe's rasteriser interpolates it (tools/st_probe.py) instead of
    (x+0.5)/W -- identical on power-of-two grids; with set_linear_taps(True) this reproduces llvmpipe on any grid"""
    lib().orc_set_raster_st(1 if on else 0)


def set_threads(n):
    lib().orc_set_threads(int(n))


def set_glsl_math(mode):
    """1 (default): GLSL sin/cos/tan/asin/acos/atan/pow as llvmpipe evaluates them; 0: libm"""
    lib().orc_set_glsl_math(int(mode))


def ocean_jacobi(g, img, divw, iters, p0=None, rows=False, nthreads=0):
    img, divw = _f32(img), _f32(divw)
    p = np.zeros((g.H, g.W), np.float32) if p0 is None else _f32(p0).copy()
    s = np.empty_like(p)
    if rows:
        lib().orc_ocean_jacobi_rows(ctypes.byref(g), _ptr(img), _ptr(divw), _ptr(p), _ptr(s), iters, nthreads)
    else:
        lib().orc_ocean_jacobi(ctypes.byref(g), _ptr(img), _ptr(divw), _ptr(p), _ptr(s), iters)
    return p


def ocean_jacobi_form(g, img, divw, iters, form, p0=None, nthreads=0):
    """a tolerance rendering of the sweep (dmg_oracle.c: orc_ocean_jacobi_rows_form); form 0 is the exact row-table form"""
    img, divw = _f32(img), _f32(divw)
    p = np.zeros((g.H, g.W), np.float32) if p0 is None else _f32(p0).copy()
    s = np.empty_like(p)
    lib().orc_ocean_jacobi_rows_form(ctypes.byref(g), _ptr(img), _ptr(divw), _ptr(p), _ptr(s), iters, nthreads, form)
    return p


def jacobi_fused_tables(g, form):
    tab = np.zeros((g.H, 8), np.float32)
    lib().orc_jacobi_fused_tables(ctypes.byref(g), form, _ptr(tab))
    return tab


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


class ClimateParams(ctypes.Structure):
    """orc_climate_params (= the layout of the product's dmg_climate_params)"""
    _fields_ = [("steps_per_year", ctypes.c_int)] + [(n, ctypes.c_float) for n in (
        "year_seconds", "diffusivity", "albedo", "olr_coeff", "olr_ref_kelvin", "emissivity", "celsius_offset", "heat_capacity_land",
        "heat_capacity_ocean", "eccentricity", "obliquity_deg", "perihelion_deg", "solar_constant", "initial_temperature")] + [("quirks", ctypes.c_int)]


def climate_params(**kw):
    p = ClimateParams()
    lib().orc_climate_params_default(ctypes.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def temperature(g, T, terrain, first_step, n_steps, params=None):
    T, terrain = _f32(T).copy(), _f32(terrain)
    s = np.empty_like(T)
    if params is None:
        lib().orc_temperature(ctypes.byref(g), _ptr(T), _ptr(s), _ptr(terrain), first_step, n_steps)
    else:
        cp = climate_params(**{n: getattr(params, n) for n, _ in ClimateParams._fields_})      # (a product-side struct is accepted: same fields)
        lib().orc_temperature_p(ctypes.byref(g), _ptr(T), _ptr(s), _ptr(terrain), ctypes.byref(cp), first_step, n_steps)
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


def flow_accumulate_flags(g, h, preblur=0.5, exponent=1.0, sel=None, lakes=0, fix_mouth_filter=0):
    h = _f32(h)
    sel = None if sel is None else _f32(sel)
    flow = np.empty_like(h)
    lib().orc_flow_accumulate_flags(ctypes.byref(g), _ptr(h), None if sel is None else _ptr(sel), preblur, exponent,
                                    lakes, fix_mouth_filter, _ptr(flow))
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


# ---------------------------------------------------------------------------------------------------------------
# Tectonics (src/filter/tectonics/Tectonics.cpp, Plate.cpp); pass arithmetic in tect_oracle.c
# ---------------------------------------------------------------------------------------------------------------
def glm_rotate(m, angle, axis):
    """glm::rotate(mat4 m, float angle, vec3 axis) as published in GLM's ext/matrix_transform.inl (the reference's
    lib/glm submodule is an empty directory in this checkout; .gitmodules:4-6 pins no version, the function has not
    changed since GLM 0.9.x).  m: 4x4 float32 indexed m[col][row]."""
    f = np.float32
    a = f(angle)
    c, s = f(_libm().cosf(float(a))), f(_libm().sinf(float(a)))       # std::cos(float) = libm cosf, not numpy's SIMD cos
    ax = np.asarray(axis, f)
    ax = ax * (f(1) / f(np.sqrt(f(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]))))   # glm::normalize: v * inversesqrt(dot)
    temp = (f(1) - c) * ax
    R = np.zeros((3, 3), f)
    R[0][0] = c + temp[0] * ax[0]; R[0][1] = temp[0] * ax[1] + s * ax[2]; R[0][2] = temp[0] * ax[2] - s * ax[1]
    R[1][0] = temp[1] * ax[0] - s * ax[2]; R[1][1] = c + temp[1] * ax[1]; R[1][2] = temp[1] * ax[2] + s * ax[0]
    R[2][0] = temp[2] * ax[0] + s * ax[1]; R[2][1] = temp[2] * ax[1] - s * ax[0]; R[2][2] = c + temp[2] * ax[2]
    m = np.asarray(m, f)
    out = np.array(m, f)
    for k in range(3):
        out[k] = m[0] * R[k][0] + m[1] * R[k][1] + m[2] * R[k][2]
    out[3] = m[3]
    return out.astype(f)


_libm_handle = None


def _libm():
    global _libm_handle
    if _libm_handle is None:
        import ctypes.util
        m = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        m.cosf.restype = m.sinf.restype = ctypes.c_float
        m.cosf.argtypes = m.sinf.argtypes = [ctypes.c_float]
        _libm_handle = m
    return _libm_handle


def tect_set_math(mode):
    lib().orc_tect_set_math(int(mode))


class TectPlate:
    """Plate (Plate.cpp:10-48): RGBA texture (height, age, 0, 0) + accumulated rotation"""

    def __init__(self, tex, angular_velocity):
        self.tex = tex
        self.rotation = np.eye(4, dtype=np.float32)
        self.av = np.asarray(angular_velocity, np.float32)

    def rotate(self):                                              # :26-28
        f = np.float32
        d = f(self.av[0] * self.av[0] + self.av[1] * self.av[1] + self.av[2] * self.av[2])
        ln = f(np.sqrt(d))                                         # glm::length
        unit = self.av * (f(1) / f(np.sqrt(d)))                    # glm::normalize (glm::rotate normalises once more)
        self.rotation = glm_rotate(self.rotation, ln, unit)

    def mat3(self):                                                # glm::mat3 a = rotation, value_ptr order
        return np.ascontiguousarray(self.rotation[:3, :3].reshape(-1), np.float32)

    def inv_mat3(self):                                            # glm::transpose(a)
        return np.ascontiguousarray(self.rotation[:3, :3].T.reshape(-1), np.float32)


def _tect_bind():
    L = lib()
    if not hasattr(L, "_tect_bound"):
        P, G, I, F = ctypes.c_void_p, ctypes.POINTER(Grid), ctypes.c_int, ctypes.c_float
        L.orc_tect_init_plate.argtypes = [G, P, I, P]
        L.orc_tect_fold_zero.argtypes = [P, I, I]
        L.orc_tect_fold.argtypes = [G, P, I, I, P, I, I, P, F, P, I, I]
        L.orc_tect_spread_flag.argtypes = [P, I, P, I, I]
        L.orc_tect_spread_ring.argtypes = [G, P, F, P, I, I]
        L.orc_tect_render.argtypes = [G, P, I, I, P]
        L.orc_tect_unfold.argtypes = [G, P, I, I, P, I, I, P, F, P, I, I]
        L._tect_bound = True
    return L


def tect_init_plates(g, img):
    L = _tect_bind()
    img = _f32(img)
    out = []
    for which in (0, 1):
        t = np.empty((g.H, g.W, 4), np.float32)
        L.orc_tect_init_plate(ctypes.byref(g), _ptr(img), which, _ptr(t))
        out.append(t)
    f = np.float32
    return [TectPlate(out[0], f(0.01) * np.array([-1, 0, 0], f)),      # Tectonics.cpp:18
            TectPlate(out[1], -f(0.01) * np.array([-1, 0, 0], f))]     # :20


def tect_fold(g, plates, S):
    """Tectonics::fold (:278-293) -> global map (S x S x 4): plate index, height, age, overlap marker"""
    L = _tect_bind()
    a = np.empty((S, S, 4), np.float32)
    L.orc_tect_fold_zero(_ptr(a), S, S)
    for idx, pl in enumerate(plates, 1):
        b = np.empty_like(a)
        m = pl.mat3()
        L.orc_tect_fold(ctypes.byref(g), _ptr(a), S, S, _ptr(pl.tex), pl.tex.shape[1], pl.tex.shape[0], _ptr(m),
                        float(idx), _ptr(b), S, S)
        a = b
    return a


def tect_spread(g, b):
    """Tectonics::oceanSpreading (:295-397)"""
    L = _tect_bind()
    S = b.shape[0]
    a = np.empty_like(b)
    L.orc_tect_spread_flag(_ptr(b), 0, _ptr(a), S, S)
    for i in list(range(5)) + list(range(5, 0, -1)):
        nxt = np.empty_like(a)
        L.orc_tect_spread_ring(ctypes.byref(g), _ptr(a), float(2.0 ** i), _ptr(nxt), S, S)
        a = nxt
    out = np.empty_like(a)
    L.orc_tect_spread_flag(_ptr(a), 1, _ptr(out), S, S)
    return out


def tect_unfold(g, b, pl, idx, shape):
    L = _tect_bind()
    out = np.empty(shape + (4,), np.float32)
    m = pl.inv_mat3()
    L.orc_tect_unfold(ctypes.byref(g), _ptr(b), b.shape[1], b.shape[0], _ptr(pl.tex), pl.tex.shape[1], pl.tex.shape[0],
                      _ptr(m), float(idx), _ptr(out), shape[1], shape[0])
    return out


def tectonics(g, img, steps, trace=None, extra_rotation=()):
    """Tectonics::Tectonics + Tectonics::run (:15-275) -> (terrain, [plate textures]).

    collision() (:399-618) is evaluated only as far as anything reads it afterwards: its result texture c is consumed
    by the render pass alone, which takes c.y = plateIndices.x (:590, :199); the other channels are overwritten before
    any use.  Its 14 a<->c swaps cancel.  The unfold loop's c<->plate swaps (:260) rotate three storage sizes among
    (c, plate 0, plate 1): W x W, W x H, W x H with period 3 -- reproduced through `c_shape`."""
    L = _tect_bind()
    plates = tect_init_plates(g, img)
    for k, theta, axis in extra_rotation:                                     # Plate::updateRotationBy
        plates[k].av = (plates[k].av + np.float32(theta) * np.asarray(axis, np.float32)).astype(np.float32)
    S = g.W
    c_shape = (S, S)
    terrain = None
    for it in range(steps):
        for pl in plates:
            pl.rotate()                                                       # :158-160
        b = tect_fold(g, plates, S)
        if trace is not None and it == 0:
            trace["fold"] = b.copy()
        b = tect_spread(g, b)
        if trace is not None and it == 0:
            trace["spread"] = b.copy()
        # collision -> c.y = b.x at c's resolution; render :186-210 fetches it at W x H
        cy = np.zeros(c_shape + (4,), np.float32)
        sy = ((np.arange(c_shape[0], dtype=np.float32) + np.float32(0.5)) / np.float32(c_shape[0]))
        sx = ((np.arange(c_shape[1], dtype=np.float32) + np.float32(0.5)) / np.float32(c_shape[1]))
        jj = np.clip(np.floor(sy * np.float32(S)).astype(int), 0, S - 1)
        ii = np.clip(np.floor(sx * np.float32(S)).astype(int), 0, S - 1)
        cy[..., 1] = b[jj][:, ii, 0]
        terrain = np.empty((g.H, g.W), np.float32)
        L.orc_tect_render(ctypes.byref(g), _ptr(cy), c_shape[1], c_shape[0], _ptr(terrain))
        for idx, pl in enumerate(plates, 1):                                  # :259-268
            new_shape, c_shape = c_shape, pl.tex.shape[:2]                    # c->swap(plate texture)
            pl.tex = tect_unfold(g, b, pl, idx, new_shape)
    return terrain, [pl.tex for pl in plates]


# ---------------------------------------------------------------------------------------------------------------
# GradientNoise (src/filter/GradientNoise.cpp); arithmetic in noise_oracle.c
# ---------------------------------------------------------------------------------------------------------------
NOISE_MODES = ["DEFAULT", "RIDGED", "BILLOWY", "IQNoise", "SWISS", "Jordan", "Plateaus"]          # GradientNoise.h:10
BLEND_MODES = ["replace", "add", "subtract", "multiply", "divide", "max", "min"]                  # Filter.cpp:210
NOISE_DEFAULTS = dict(scale=1.0, lacunarity=2.0, persistence=0.5, min=-10.0, max=10.0, octaves=8, seed=0, warp=0.0)  # .h:39


def noise_seed_offset(seed):
    out = np.zeros(3, np.float32)
    L = lib()
    L.orc_noise_seed_offset.argtypes = [ctypes.c_int, ctypes.c_void_p]
    L.orc_noise_seed_offset(int(seed), _ptr(out))
    return out


def gradient_noise(g, img, mode="DEFAULT", blend="replace", sel=None, **params):
    """GradientNoiseFilter::step (GradientNoise.cpp:61-500) -> new terrain"""
    prm = dict(NOISE_DEFAULTS)
    prm.update(params)
    img = _f32(img)
    sel = None if sel is None else _f32(sel)
    p6 = np.array([prm["scale"], prm["lacunarity"], prm["persistence"], prm["min"], prm["max"], prm["warp"]], np.float32)
    out = np.empty_like(img)
    L = lib()
    L.orc_gradient_noise.argtypes = [ctypes.POINTER(Grid), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.orc_gradient_noise(ctypes.byref(g), _ptr(img), None if sel is None else _ptr(sel), _ptr(p6), int(prm["octaves"]),
                         int(prm["seed"]), NOISE_MODES.index(mode), BLEND_MODES.index(blend), _ptr(out))
    return out
