"""ctypes driver for oracle/_ref/librefgl.so: the reference's GLSL run verbatim on
Mesa llvmpipe.  TEST INFRASTRUCTURE ONLY (golden generation + "reference" CPU
baseline); the product path never imports this.

The host loops below restate the reference's per-filter pass loops; every function
cites the reference lines it follows.  Shader keys are "<file>:<line>" of the
`Shader::builder()` token in the reference at v1.
"""
import ctypes
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE.parent / "_ref" / "librefgl.so"

K_JACOBI = "OceanCurrents.cpp:442"
K_DIVERGENCE = "OceanCurrents.cpp:367"
K_DIFFUSION = "OceanCurrents.cpp:639"
K_ADVECT = "OceanCurrents.cpp:271"
K_SUBDIV = "OceanCurrents.cpp:522"
K_SETZERO = "OceanCurrents.cpp:64"
K_BLUR = "BlurMenu.cpp:65"
K_TEMPERATURE = "Temperature.cpp:52"
K_EROSION = "cpufilter.cpp:110"
K_UPDRIFT = "cpufilter.cpp:48"
K_SCALEDH = "cpufilter.cpp:66"
K_FLOWDIR = "FlowFilter.cpp:109"
K_FLOWMASK = "FlowFilter.cpp:268"
K_COPY = "Shader.h:copy_img"


def available():
    return LIB_PATH.exists()


class RefGL:
    _inst = None

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def __init__(self):
        L = ctypes.CDLL(str(LIB_PATH))
        L.refgl_last_error.restype = ctypes.c_char_p
        L.refgl_renderer.restype = ctypes.c_char_p
        L.refgl_shader_key.restype = ctypes.c_char_p
        L.refgl_shader_code.restype = ctypes.c_char_p
        L.refgl_shader_code.argtypes = [ctypes.c_char_p]
        L.refgl_set_canvas.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_float]
        L.refgl_tex_create.argtypes = [ctypes.c_int] * 4
        L.refgl_tex_upload.argtypes = [ctypes.c_int, ctypes.c_void_p]
        L.refgl_tex_download.argtypes = [ctypes.c_int, ctypes.c_void_p]
        L.refgl_program.argtypes = [ctypes.c_char_p]
        L.refgl_program_source.argtypes = [ctypes.c_char_p]
        L.refgl_uniform1i.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.refgl_noise_seed_offset.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        L.refgl_uniform1f.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_float]
        L.refgl_uniform2f.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_float, ctypes.c_float]
        L.refgl_uniform3f.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_float, ctypes.c_float, ctypes.c_float]
        L.refgl_uniform_mat3.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_float)]
        L.refgl_apply.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int)]
        L.refgl_pingpong.restype = ctypes.c_double
        L.refgl_pingpong.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        self.L = L
        if L.refgl_init() != 0:
            raise RuntimeError("refgl_init: " + L.refgl_last_error().decode())
        self._progs = {}
        self._ncomp = {}
        self._shape = {}

    # -- thin wrappers -----------------------------------------------------
    def renderer(self):
        return self.L.refgl_renderer().decode()

    def shader_keys(self):
        return [self.L.refgl_shader_key(i).decode() for i in range(self.L.refgl_num_shaders())]

    def shader_code(self, key):
        return self.L.refgl_shader_code(key.encode()).decode()

    def set_canvas(self, coords, circumference):
        c = (ctypes.c_float * 4)(*[float(x) for x in coords])
        self.L.refgl_set_canvas(c, float(circumference))

    def tex(self, W, H, ncomp=1, linear=False, data=None):
        t = self.L.refgl_tex_create(W, H, ncomp, 1 if linear else 0)
        self._ncomp[t] = ncomp
        self._shape[t] = (H, W)
        if data is not None:
            self.upload(t, data)
        return t

    def upload(self, t, data):
        H, W = self._shape[t]
        a = np.ascontiguousarray(data, dtype=np.float32)
        assert a.size == H * W * self._ncomp[t], (a.shape, H, W, self._ncomp[t])
        if self.L.refgl_tex_upload(t, a.ctypes.data) != 0:
            raise RuntimeError("upload failed")

    def download(self, t):
        H, W = self._shape[t]
        n = self._ncomp[t]
        out = np.empty((H, W, n) if n > 1 else (H, W), dtype=np.float32)
        if self.L.refgl_tex_download(t, out.ctypes.data) != 0:
            raise RuntimeError("download failed")
        return out

    def swap(self, a, b):
        self.L.refgl_tex_swap(a, b)
        self._shape[a], self._shape[b] = self._shape[b], self._shape[a]

    def delete(self, *ts):
        for t in ts:
            self.L.refgl_tex_delete(t)

    def program(self, key):
        if key not in self._progs:
            p = self.L.refgl_program(key.encode())
            if p < 0:
                raise RuntimeError("program %s: %s" % (key, self.L.refgl_last_error().decode()))
            self._progs[key] = p
        return self._progs[key]

    def program_source(self, frag_src):
        """probe program from GLSL written by the tooling (vertex2D + frag_src)"""
        p = self.L.refgl_program_source(frag_src.encode())
        if p < 0:
            raise RuntimeError(self.L.refgl_last_error().decode())
        return p

    def uniform1f(self, prog, name, v):
        return self.L.refgl_uniform1f(prog, name.encode(), float(v))

    def uniform1i(self, prog, name, v):
        return self.L.refgl_uniform1i(prog, name.encode(), int(v))

    def uniform2f(self, prog, name, a, b):
        return self.L.refgl_uniform2f(prog, name.encode(), float(a), float(b))

    def uniform3f(self, prog, name, a, b, c):
        return self.L.refgl_uniform3f(prog, name.encode(), float(a), float(b), float(c))

    def uniform_mat3(self, prog, name, m9):
        arr = (ctypes.c_float * 9)(*[float(x) for x in m9])
        return self.L.refgl_uniform_mat3(prog, name.encode(), arr)

    @staticmethod
    def _pairs(samplers):
        names = (ctypes.c_char_p * len(samplers))(*[k.encode() for k in samplers])
        ids = (ctypes.c_int * len(samplers))(*[int(v) for v in samplers.values()])
        return names, ids

    def apply(self, prog, target, samplers):
        names, ids = self._pairs(samplers)
        self.L.refgl_apply(prog, target, len(samplers), names, ids)

    def pingpong(self, prog, field_name, scratch, iters, samplers):
        """for i<iters: apply(prog, scratch); samplers[field_name].swap(scratch).  Returns seconds."""
        names, ids = self._pairs(samplers)
        slot = list(samplers).index(field_name)
        return self.L.refgl_pingpong(prog, scratch, iters, len(samplers), names, ids, slot)


# ---------------------------------------------------------------------------
# Host loops restating the reference's filters on top of the verbatim GLSL
# ---------------------------------------------------------------------------
FULL_GLOBE = (np.float32(-np.pi / 2), np.float32(np.pi / 2), np.float32(-np.pi), np.float32(np.pi))  # Project.h:97


def ocean_jacobi(img, divw, iters, coords=FULL_GLOBE, circ=42000.0, p0=None, timing=None):
    """OceanCurrents::jacobi (OceanCurrents.cpp:437-509): pressure:=0 (:495-497) then `iters` x
    { apply(jacobiProgram, scratch); pressure.swap(scratch) } (:500-508).
    img: terrain R32F NEAREST (Layer texture); pressure/scratch/divw R32F LINEAR (:44-45,54)."""
    g = RefGL.get()
    g.set_canvas(coords, circ)
    H, W = img.shape
    t_img = g.tex(W, H, 1, False, img)
    t_b = g.tex(W, H, 1, True, divw)
    t_p = g.tex(W, H, 1, True, np.zeros((H, W), np.float32) if p0 is None else p0)
    t_s = g.tex(W, H, 1, True, np.zeros((H, W), np.float32))
    prog = g.program(K_JACOBI)
    secs = g.pingpong(prog, "pressure", t_s, iters, {"img": t_img, "divw": t_b, "pressure": t_p})
    if timing is not None:
        timing.append(secs)
    out = g.download(t_p)
    g.delete(t_img, t_b, t_p, t_s)
    return out


def ocean_divergence(img, v, pressurefactor=100.0, coords=FULL_GLOBE, circ=42000.0):
    """OceanCurrents::divergence (:363-435). v RG32F LINEAR; `pressure` is bound only for textureSize."""
    g = RefGL.get()
    g.set_canvas(coords, circ)
    H, W = img.shape
    t_img = g.tex(W, H, 1, False, img)
    t_v = g.tex(W, H, 2, True, v)
    t_p = g.tex(W, H, 1, True, np.zeros((H, W), np.float32))
    t_o = g.tex(W, H, 1, True)
    prog = g.program(K_DIVERGENCE)
    g.uniform1f(prog, "pressurefactor", pressurefactor)      # :430-431
    g.apply(prog, t_o, {"img": t_img, "v": t_v, "pressure": t_p})
    out = g.download(t_o)
    g.delete(t_img, t_v, t_p, t_o)
    return out


def ocean_diffusion(img, v, iters=50, coords=FULL_GLOBE, circ=42000.0):
    """OceanCurrents::diffusion (:625-708): `iters` x { apply(jacobi, v_scratch); v_scratch.swap(v) }."""
    g = RefGL.get()
    g.set_canvas(coords, circ)
    H, W = img.shape
    t_img = g.tex(W, H, 1, False, img)
    t_v = g.tex(W, H, 2, True, v)
    t_s = g.tex(W, H, 2, True, np.zeros((H, W, 2), np.float32))
    prog = g.program(K_DIFFUSION)
    g.pingpong(prog, "v", t_s, iters, {"img": t_img, "v": t_v})
    out = g.download(t_v)
    g.delete(t_img, t_v, t_s)
    return out


def ocean_advect(img, v, coords=FULL_GLOBE, circ=42000.0):
    """OceanCurrents::advect (:235-360): apply(advectProgram, v_scratch); v.swap(v_scratch)."""
    g = RefGL.get()
    g.set_canvas(coords, circ)
    H, W = img.shape
    t_img = g.tex(W, H, 1, False, img)
    t_v = g.tex(W, H, 2, True, v)
    t_s = g.tex(W, H, 2, True)
    g.apply(g.program(K_ADVECT), t_s, {"img": t_img, "v": t_v})
    out = g.download(t_s)
    g.delete(t_img, t_v, t_s)
    return out


def ocean_subdiv(img, v, pressure, pressurefactor=100.0, coords=FULL_GLOBE, circ=42000.0):
    """OceanCurrents::subDiv (:511-622)."""
    g = RefGL.get()
    g.set_canvas(coords, circ)
    H, W = img.shape
    t_img = g.tex(W, H, 1, False, img)
    t_v = g.tex(W, H, 2, True, v)
    t_p = g.tex(W, H, 1, True, pressure)
    t_s = g.tex(W, H, 2, True)
    prog = g.program(K_SUBDIV)
    g.uniform1f(prog, "pressurefactor", pressurefactor)      # :618-619
    g.apply(prog, t_s, {"img": t_img, "v": t_v, "pressure": t_p})
    out = g.download(t_s)
    g.delete(t_img, t_v, t_p, t_s)
    return out


def ocean_step(img, v, n_outer=1, jacobi_iters=5000, diffusion_iters=50, pressurefactor=100.0,
               coords=FULL_GLOBE, circ=42000.0):
    """OceanCurrents::run body (:155-209) without the WIP resize(:207)/while(true)(:212):
    advect -> diffusion -> divergence -> jacobi -> subDiv.  Returns (v, pressure, divw)."""
    p = None
    divw = None
    for _ in range(n_outer):
        v = ocean_advect(img, v, coords, circ)
        v = ocean_diffusion(img, v, diffusion_iters, coords, circ)
        divw = ocean_divergence(img, v, pressurefactor, coords, circ)
        p = ocean_jacobi(img, divw, jacobi_iters, coords, circ)
        v = ocean_subdiv(img, v, p, pressurefactor, coords, circ)
    return v, p, divw


def blur_rlist(radius):
    """Blur::Blur radius schedule (BlurMenu.cpp:27,79-94) in fp32, same operation order."""
    f = np.float32
    radius = f(radius) / f(2)
    R = f(f(radius * radius) / f(2))
    rl = []
    i = f(1.0)
    inc = f(0.5)
    if R < 3:
        k = f(1.0 / np.sqrt(np.float64(f(55) / R)))   # `1/sqrt(55/R)`: float/float -> double sqrt -> float k
        inc = k
        i = k
    while R >= f(i * i):
        R = f(R - f(i * i))
        rl.append(i)
        i = f(i + inc)
    if R > 0:
        rl.append(f(np.sqrt(np.float64(R))))
    return sorted(rl)


def blur(field, radius, coords=FULL_GLOBE, circ=42000.0, rlist=None):
    """Blur::Blur + Blur::step until finished (BlurMenu.cpp:24-117). tex1/tex2 R32F LINEAR (:29-30)."""
    g = RefGL.get()
    g.set_canvas(coords, circ)
    H, W = field.shape
    t1 = g.tex(W, H, 1, True, field)
    t2 = g.tex(W, H, 1, True)
    prog = g.program(K_BLUR)
    rl = blur_rlist(radius) if rlist is None else rlist
    i = 0
    while True:
        g.uniform2f(prog, "direction", 0.0, rl[i])           # :100
        g.apply(prog, t2, {"img": t1})                        # :102
        g.uniform2f(prog, "direction", rl[i], 0.0)           # :103
        g.apply(prog, t1, {"img": t2})                        # :104
        i += 1
        if i > len(rl) - 1:                                   # :111
            break
    out = g.download(t1)
    g.delete(t1, t2)
    return out


def temperature(T, terrain, first_step, n_steps, coords=FULL_GLOBE, circ=42000.0):
    """Temperature::run inner pass (Temperature.cpp:140-150): uniform M = 2*pi/15000*i (float(double)),
    apply(mainfilter, scratch2); terrain.swap(scratch2).  `img` = T is the Layer texture (NEAREST),
    scratch1 = terrain height (NEAREST)."""
    g = RefGL.get()
    g.set_canvas(coords, circ)
    H, W = T.shape
    t_T = g.tex(W, H, 1, False, T)
    t_s = g.tex(W, H, 1, False)
    t_h = g.tex(W, H, 1, False, terrain)
    prog = g.program(K_TEMPERATURE)
    for i in range(first_step, first_step + n_steps):
        g.uniform1f(prog, "M", np.float32(np.pi * 2 / 15000 * i))   # :146
        g.apply(prog, t_s, {"img": t_T, "scratch1": t_h})
        g.swap(t_T, t_s)
    out = g.download(t_T)
    g.delete(t_T, t_s, t_h)
    return out


def erosion_shader(h, flow, updrift, factor=1.0, slope_exponent=1.0, coords=FULL_GLOBE, circ=42000.0):
    """The erosion/uplift pass of cpufilter::run (cpufilter.cpp:106-214).  The reference renders INTO the
    texture it samples (:213, undefined in GL); pinned here to read = old heights, write = new buffer."""
    g = RefGL.get()
    g.set_canvas(coords, circ)
    H, W = h.shape
    t_h = g.tex(W, H, 1, False, h)
    t_f = g.tex(W, H, 1, False, flow)
    t_u = g.tex(W, H, 1, False, updrift)
    t_o = g.tex(W, H, 1, False)
    prog = g.program(K_EROSION)
    g.uniform1f(prog, "slope_exponent", slope_exponent)      # :207-208
    g.uniform1f(prog, "factor", factor)                      # :209-210
    g.apply(prog, t_o, {"img": t_h, "scratch2": t_f, "updrift": t_u})
    out = g.download(t_o)
    g.delete(t_h, t_f, t_u, t_o)
    return out


def flow_direction_and_mask(img, sel=None, coords=FULL_GLOBE, circ=42000.0):
    """The two GLSL passes of FlowFilter::findMagicNumbers (FlowFilter.cpp:109-315) on an (already blurred) terrain:
    direction codes 0..9 -> scratch1 with cornerCoords moved 1e-3 off the poles (:253-256), then the receiver
    bitmask -> scratch2 with the project's own coords (:304-310).  Returns (codes, mask)."""
    g = RefGL.get()
    H, W = img.shape
    sel = np.ones((H, W), np.float32) if sel is None else sel
    t_img = g.tex(W, H, 1, False, img)
    t_sel = g.tex(W, H, 1, False, sel)
    t_s1 = g.tex(W, H, 1, False)
    t_s2 = g.tex(W, H, 1, False)
    cm = [np.float32(c) for c in coords]
    cm[0] = np.float32(cm[0] + np.float32(1e-3))          # coordsMod[0]+=1e-3 (float vector, double literal)
    cm[1] = np.float32(cm[1] - np.float32(1e-3))
    g.set_canvas(cm, circ)
    g.apply(g.program(K_FLOWDIR), t_s1, {"img": t_img, "sel": t_sel})
    g.set_canvas(coords, circ)
    g.apply(g.program(K_FLOWMASK), t_s2, {"img": t_img, "sel": t_sel, "scratch1": t_s1})
    codes, mask = g.download(t_s1), g.download(t_s2)
    g.delete(t_img, t_sel, t_s1, t_s2)
    return codes, mask


def lake_adjacency(lake_id_bits, coords=FULL_GLOBE, circ=42000.0):
    """FlowFilter::findAllConnections GLSL pass (FlowFilter.cpp:405-447): scratch1 = lake ids (int bits + 2^30 read
    as float, -1 elsewhere) -> scratch2 bitmask of neighbours that belong to another lake."""
    g = RefGL.get()
    g.set_canvas(coords, circ)
    H, W = lake_id_bits.shape
    t_s1 = g.tex(W, H, 1, False, lake_id_bits)
    t_img = g.tex(W, H, 1, False, np.zeros((H, W), np.float32))
    t_s2 = g.tex(W, H, 1, False)
    g.apply(g.program("FlowFilter.cpp:405"), t_s2, {"img": t_img, "scratch1": t_s1})
    out = g.download(t_s2)
    g.delete(t_s1, t_img, t_s2)
    return out


def thermal_erosion(img, coords=FULL_GLOBE, circ=42000.0, include="#noproj"):
    """ThermalErosion::step (ThermalErosion.cpp:25-142): 10 x { talus loss -> scratch1 ; gain pass -> scratch2 ;
    scratch2.swap(terrain) }.  All textures NEAREST.
    include: what stands at `.include(p->canvas->projection_shader())` (:30) -- "" the canvas's real shader text (Canvas.cpp:188-277
    + Equiretangular.cpp:16-23), "#inita" the same with its uninitialised `bool a` set false, "#noproj" (default) nothing.  Under
    llvmpipe the real text discards EVERY fragment: `bool a; projection(st, a); if(a) discard;` reads a variable nothing has
    written (and filter programs never get the view uniforms, Project.cpp:456-461), so the tool would do nothing; "#inita" and
    "#noproj" agree bit for bit, and are what the committed goldens hold."""
    g = RefGL.get()
    g.set_canvas(coords, circ)
    H, W = img.shape
    t_img = g.tex(W, H, 1, False, img)
    t_s1 = g.tex(W, H, 1, False)
    t_s2 = g.tex(W, H, 1, False)
    p1, p2 = g.program("ThermalErosion.cpp:27" + include), g.program("ThermalErosion.cpp:84" + include)
    for _ in range(10):
        g.apply(p1, t_s1, {"img": t_img})
        g.apply(p2, t_s2, {"img": t_img, "scratch1": t_s1})
        g.swap(t_s2, t_img)
    out = g.download(t_img)
    g.delete(t_img, t_s1, t_s2)
    return out


def morph_rlist(radius):
    """Morphological::Morphological radius schedule (Morphological.cpp:31-43), float arithmetic"""
    f = np.float32
    radius = f(radius)
    r, x = [], 1
    while radius >= 0:
        if x < radius:
            radius = f(radius - f(x))
            r.append(f(x))
            x *= 2
        else:
            r.append(radius)
            break
    return sorted(r)


def morphological(field, radius, op, coords=FULL_GLOBE, circ=42000.0):
    """Morphological::step until finished (Morphological.cpp:75-85); op = "min" (Erode) or "max" (Dilate)."""
    g = RefGL.get()
    g.set_canvas(coords, circ)
    H, W = field.shape
    t_t = g.tex(W, H, 1, False, field)
    t_s = g.tex(W, H, 1, False)
    prog = g.program("Morphological.cpp:63#" + op)
    for r in morph_rlist(radius):
        g.uniform1f(prog, "radius", r)
        g.apply(prog, t_s, {"target": t_t})
        g.swap(t_s, t_t)
    out = g.download(t_t)
    g.delete(t_t, t_s)
    return out


def morphological_gradient(field, radius, coords=FULL_GLOBE, circ=42000.0):
    """MorphologicalGradient (Morphological.cpp:87-131): dilate((radius+1)/2) - erode(radius/2) of the same input"""
    er = morphological(field, np.float32(radius) / np.float32(2), "min", coords, circ)
    di = morphological(field, (np.float32(radius) + np.float32(1)) / np.float32(2), "max", coords, circ)
    return (di - er).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------
# Tectonics (src/filter/tectonics/Tectonics.cpp, Plate.cpp)
# ---------------------------------------------------------------------------------------------------------------
def _oracle_host():
    """host-side arithmetic of Plate (glm::rotate, mat3 packing) lives in oracle.py; the GLSL runs here"""
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import oracle as orc
    return orc


def _plate_uniforms(g, prog, pl, index):                 # Plate::setPlateUniforms (Plate.cpp:30-44)
    g.uniform1f(prog, "plateIndex", index)
    g.uniform_mat3(prog, "rotationmatrix", pl.mat3())
    g.uniform_mat3(prog, "inverserotationmatrix", pl.inv_mat3())
    g.uniform3f(prog, "plateVelocity", *pl.av)


def tectonics(img, steps, coords=FULL_GLOBE, circ=42000.0, trace=None):
    """Tectonics::Tectonics + Tectonics::run (Tectonics.cpp:15-275) for `steps` (reference: 70) iterations.
    Returns (terrain, [plate textures]).  Texture ids AND sizes move with Texture::swap exactly as in the reference:
    the scratch maps a, b, c are W x W (:22-24) and trade places with the W x H plate textures in the unfold loop."""
    g = RefGL.get()
    g.set_canvas(coords, circ)
    H, W = img.shape
    t_img = g.tex(W, H, 1, False, img)
    orc = _oracle_host()
    f = np.float32
    plates = [orc.TectPlate(g.tex(W, H, 4, False), f(0.01) * np.array([-1, 0, 0], f)),       # :16-18
              orc.TectPlate(g.tex(W, H, 4, False), -f(0.01) * np.array([-1, 0, 0], f))]      # :17,:20
    a, b, c = g.tex(W, W, 4, False), g.tex(W, W, 4, False), g.tex(W, W, 4, False)
    P = g.program
    g.apply(P("Tectonics.cpp:28"), plates[0].tex, {"img": t_img})             # :41-43
    g.apply(P("Tectonics.cpp:44"), plates[1].tex, {"img": t_img})             # :57-59
    zero, fold = P("Tectonics.cpp:99"), P("Tectonics.cpp:112")
    for it in range(steps):
        for pl in plates:
            pl.rotate()                                                       # :158-160
        # fold(setzero, foldShader, a, b) :278-293
        g.apply(zero, a, {})
        g.swap(a, b)
        for idx, pl in enumerate(plates, 1):
            _plate_uniforms(g, fold, pl, idx)
            g.apply(fold, a, {"foldtex": b, "plate": pl.tex})
            g.swap(a, b)
        if trace is not None and it == 0:
            trace["fold"] = g.download(b)
        # oceanSpreading :295-397
        g.apply(P("Tectonics.cpp:303"), a, {"foldtex": b}); g.swap(a, b)
        spread = P("Tectonics.cpp:325")
        for i in list(range(5)) + list(range(5, 0, -1)):
            g.uniform1f(spread, "radius", np.float32(2.0 ** i))
            g.apply(spread, a, {"foldtex": b}); g.swap(a, b)
        g.apply(P("Tectonics.cpp:376"), a, {"foldtex": b}); g.swap(a, b)
        if trace is not None and it == 0:
            trace["spread"] = g.download(b)
        # collision :399-618
        g.apply(P("Tectonics.cpp:400"), a, {}); g.swap(a, c)
        setrot = P("Tectonics.cpp:410")
        for idx, pl in enumerate(plates, 1):
            _plate_uniforms(g, setrot, pl, idx)
            g.apply(setrot, a, {"foldtex": c, "plateIndices": b}); g.swap(a, c)
        g.apply(P("Tectonics.cpp:445"), a, {"foldtex": c, "plateIndices": b}); g.swap(a, c)
        coll = P("Tectonics.cpp:534")
        for i in range(10):
            g.uniform1f(coll, "radius", float(i))
            g.apply(coll, a, {"foldtex": c, "plateIndices": b}); g.swap(a, c)
        # render :186-209
        g.apply(P("Tectonics.cpp:186"), t_img, {"foldtex": c})
        # back to plates :216-267
        unfold = P("Tectonics.cpp:216")
        for idx, pl in enumerate(plates, 1):
            g.swap(c, pl.tex)
            _plate_uniforms(g, unfold, pl, idx)
            g.apply(unfold, pl.tex, {"foldtex": b, "oldPlate": c})
    out = g.download(t_img)
    pts = [g.download(pl.tex) for pl in plates]
    if trace is not None:
        trace["sizes"] = [g._shape[pl.tex] for pl in plates] + [g._shape[x] for x in (a, b, c)]
    g.delete(t_img, a, b, c, *[pl.tex for pl in plates])
    return out, pts


# ---------------------------------------------------------------------------------------------------------------
# GradientNoise (src/filter/GradientNoise.cpp)
# ---------------------------------------------------------------------------------------------------------------
NOISE_DEFAULTS = dict(scale=1.0, lacunarity=2.0, persistence=0.5, min=-10.0, max=10.0, octaves=8, seed=0, warp=0.0)  # .h:39


def noise_seed_offset(seed):
    """GradientNoise.cpp:489-491: mt19937{seed} + uniform_real_distribution<float>(0, 10000), three draws"""
    out = (ctypes.c_float * 3)()
    RefGL.get().L.refgl_noise_seed_offset(int(seed), out)
    return np.array(out[:], np.float32)


def gradient_noise(img, mode="DEFAULT", blend="replace", sel=None, coords=FULL_GLOBE, circ=42000.0, **params):
    """GradientNoiseFilter::step (GradientNoise.cpp:61-500): one pass; returns the new terrain"""
    g = RefGL.get()
    g.set_canvas(coords, circ)
    prm = dict(NOISE_DEFAULTS); prm.update(params)
    H, W = img.shape
    sel = np.ones_like(img) if sel is None else sel
    t_img, t_sel, t_out = g.tex(W, H, 1, False, img), g.tex(W, H, 1, False, sel), g.tex(W, H, 1, False)
    p = g.program("GradientNoise.cpp:440#%s_%s" % (mode, blend))
    g.uniform1f(p, "scale", prm["scale"]); g.uniform1i(p, "octaves", prm["octaves"])           # :462-466
    g.uniform1f(p, "lacunarity", prm["lacunarity"]); g.uniform1f(p, "persistence", prm["persistence"])
    g.uniform1f(p, "warp_factor", prm["warp"])
    g.uniform1f(p, "lower_limit", prm["min"]); g.uniform1f(p, "higher_limit", prm["max"])
    g.uniform3f(p, "seed_offset", *noise_seed_offset(prm["seed"]))
    g.apply(p, t_out, {"img": t_img, "sel": t_sel})
    out = g.download(t_out)
    g.delete(t_img, t_sel, t_out)
    return out
