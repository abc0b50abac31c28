"""Where does the advect pass leave the reference's pixels?  Cuts the reference's advect shader (lifted by
oracle/refgl at build time) after each intermediate, runs the cut shader on llvmpipe, and compares with the same
intermediate of the oracle (orc_set_advect_stage).  This is how the merged wind constant ((st.y-0.5) * 9.42477) and
the (0.0001*a)*a association were found.  Needs /root/reference (oracle/_ref/librefgl.so); CPU only.

    python tools/advect_probe.py [golden-name]
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
for p in ("oracle", "oracle/refgl", "tests"):
    sys.path.insert(0, str(ROOT / p))
import oracle  # noqa: E402
import refgl  # noqa: E402
from conftest import load_golden  # noqa: E402

G = load_golden(sys.argv[1] if len(sys.argv) > 1 else "ocean_passes_128x64_full")
img, v = G["img"], G["v"]
H, W = img.shape
coords, circ = tuple(G["coords"]), float(G["circ"])
g = refgl.RefGL.get()
src = g.shader_code(refgl.K_ADVECT)


def upto(marker, tail):
    return src[:src.index(marker)] + tail + " }\n"


ROT = "\tcartesian_coord = rotation_matrix(-arclength"
WIND = "\tfc = fc + wind*(1-pow(stress"
stages = {
    4: upto(ROT, "fc = vec2(arclength, rotation_direction.x);"),
    5: upto(ROT, "fc = rotation_direction.yz;"),
    6: upto(ROT, "fc = v_cart.xy;"),
    8: upto("\tvec2 spheric_coord_new = cartesian_to_spheric", "fc = cartesian_coord.xy;"),
    9: upto("\tvec2 coord = spheric_to_tex", "fc = spheric_coord_new;"),
    2: upto("\tvec2 newV = texture(v, coord).xy;", "fc = coord;"),
    3: upto("\tvec3 testv = v_to_cartesian", "fc = newV;"),
    1: upto("\t//if (abs(st.y-0.5)<0.05", ""),
    12: upto(WIND, "fc = wind;"),
    10: upto(WIND, "fc = stress;"),
    11: upto(WIND, "fc = pow(stress,-vec2(1.0)/24.0*2);"),
    13: upto(WIND, "fc = wind*(1-pow(stress,-vec2(1.0)/24.0*2));"),
    14: upto(WIND, "fc = fc*(1-pow(vec2(0.4),vec2(1.0/24)));"),
    0: src,
}
NAMES = {4: "arclength, rotation_direction.x", 5: "rotation_direction.yz", 6: "v_cart.xy", 8: "rotated cartesian_coord.xy",
         9: "spheric_coord_new", 2: "coord", 3: "texture(v, coord)", 1: "newV (before the wind term)", 12: "wind",
         10: "stress", 11: "pow(stress, -1/12)", 13: "wind*(1-pow)", 14: "fc*(1-pow(0.4,1/24))", 0: "fc (whole shader)"}
og = oracle.grid(W, H, coords, circ)
ocean = img <= 0
oracle.set_linear_taps(True)
for st, fs in stages.items():
    g.set_canvas(coords, circ)
    t_img, t_v, t_s = g.tex(W, H, 1, False, img), g.tex(W, H, 2, True, v), g.tex(W, H, 2, True)
    g.apply(g.program_source(fs), t_s, {"img": t_img, "v": t_v})
    ref = g.download(t_s)
    g.delete(t_img, t_v, t_s)
    oracle.lib().orc_set_advect_stage(st)
    got = oracle.ocean_advect(og, img, v)
    oracle.lib().orc_set_advect_stage(0)
    bad = (got != ref) & ocean[..., None]
    print("%-32s differing components %5d %5d of %d ocean cells" % (NAMES[st], bad[..., 0].sum(), bad[..., 1].sum(), ocean.sum()))
