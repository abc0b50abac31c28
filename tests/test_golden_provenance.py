"""The committed fixtures are outputs of the reference's own GLSL: where the llvmpipe harness is available (it is
built by `make -C oracle` whenever /root/reference and Mesa exist; the built library travels to the GPU box) a sample
of every fixture family is regenerated and must equal the committed arrays bit for bit.  Runs in a child process so a
missing or fragile Mesa cannot take the test session down."""
import subprocess
import sys

import pytest

from conftest import ROOT

CHILD = r"""
import sys
sys.path.insert(0, r'%(root)s'); sys.path.insert(0, r'%(root)s/oracle'); sys.path.insert(0, r'%(root)s/oracle/refgl'); sys.path.insert(0, r'%(root)s/tests')
import numpy as np
import refgl
from conftest import load_golden
def same(a, b, what):
    assert a.shape == b.shape and np.array_equal(a, b), what
G = load_golden('jacobi_128x64_full')
same(refgl.ocean_jacobi(G['img'], G['divw'], 60, coords=tuple(G['coords']), p0=G['p0']), G['p60'], 'jacobi')
G = load_golden('ocean_passes_128x64_full')
same(refgl.ocean_divergence(G['img'], G['v'], 100.0, coords=tuple(G['coords'])), G['divergence'], 'divergence')
same(refgl.ocean_advect(G['img'], G['v'], coords=tuple(G['coords'])), G['advect'], 'advect')
G = load_golden('blur_128x64_r10')
same(refgl.blur(G['field'], 10.0), G['out'], 'blur')
# ThermalErosion.cpp:30 includes the canvas's projection shader (Canvas.cpp:188-277), lifted as text like the rest.  Its main-body
# snippet reads an uninitialised bool (`bool a; projection(st, a); if(a) discard;`) and llvmpipe discards every fragment: with the
# reference's text verbatim the tool changes nothing.  With that one variable initialised -- or without the include -- the passes
# run, and agree bit for bit: that is what the fixture holds (VERDICT r4 item 8).
G = load_golden('thermal_128x64_full')
same(refgl.thermal_erosion(G['h'], include='#noproj'), G['out'], 'thermal, no include')
same(refgl.thermal_erosion(G['h'], include='#inita'), G['out'], 'thermal, real include with `bool a = false`')
same(refgl.thermal_erosion(G['h'], include=''), G['h'], 'thermal, real include verbatim: every fragment discarded')
G = load_golden('tectonics_globe_64x32_s70')
t, pl = refgl.tectonics(G['img'], int(G['steps']))
same(t, G['terrain'], 'tectonics terrain'); same(pl[0], G['plate0'], 'plate0'); same(pl[1], G['plate1'], 'plate1')
G = load_golden('noise_128x64_modes')
for m in ('DEFAULT', 'SWISS', 'Plateaus'):
    same(refgl.gradient_noise(G['img'], m, 'replace', seed=int(G['seed']), warp=float(G['warp'])), G['out_' + m], 'noise ' + m)
print('PROVENANCE_OK', refgl.RefGL.get().renderer())
"""


def test_fixtures_regenerate_from_the_reference_glsl():
    if not (ROOT / "oracle" / "_ref" / "librefgl.so").exists():
        pytest.skip("oracle/_ref/librefgl.so not built here (needs /root/reference + Mesa at build time)")
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=str(ROOT))], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and ("swrast" in r.stderr or "refgl_init" in r.stderr or "libglapi" in r.stderr):
        pytest.skip("Mesa llvmpipe not usable on this host: " + r.stderr.strip().splitlines()[-1][:120])
    assert r.returncode == 0 and "PROVENANCE_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    assert "llvmpipe" in r.stdout
