#!/usr/bin/env python3
"""Lift the reference's `Shader::builder()...create(...)` expressions, as text,
out of /root/reference/src/filter/*.cpp into a generated C++ translation unit
(oracle/_ref/gen_shaders.cpp) that rebuilds each shader with the reference's OWN
Shader::builder (src/Shader.cpp, compiled in place) and registers it under the
key "<file>:<line of the Shader::builder() token>".

TEST INFRASTRUCTURE ONLY.  The generated file contains reference text and is
therefore written to oracle/_ref/ (git-ignored), never committed.

Usage: extract_shaders.py <reference_root> <out.cpp>
"""
import re
import sys
from pathlib import Path

# file -> C++ statements that must be in scope for the lifted expressions
# (host-side literals the shader text is concatenated with).
# (file, key suffix, prelude): a file may be lifted more than once with different host-side splices
FILES = {
    "src/filter/OceanCurrents.cpp": "",
    "src/filter/BlurMenu.cpp": "",
    "src/filter/Temperature.cpp": "",
    # cpufilter.cpp:42  `int N = 50;` is spliced into the GLSL via std::to_string(N) (:52,:72)
    "src/filter/cpufilter.cpp": "int N = 50;",
    "src/filter/FlowFilter.cpp": "",
    # ThermalErosion.cpp:30 includes p->canvas->projection_shader(): the display-projection helpers of the project's canvas
    # (AbstractCanvas::projection_shader, src/projections/Canvas.cpp:188-277, with the canvas's inverseShader() -- the project's
    # canvas is an Equiretangular, src/Project.cpp:109,185; src/projections/Equiretangular.cpp:16-23).  The canvas CLASSES need
    # imgui/glm; their two shader expressions are lifted as text like every other one and handed out by a two-method struct
    # that stands where `p->canvas` stands (thermal_prelude below).
    "src/filter/ThermalErosion.cpp": ("thermal",),
    # ... and the two variants that show what that include does to the tool (tests/test_golden_provenance.py):
    #   #noproj  an empty shader in the include's place: the erosion passes by themselves -- the committed goldens
    #   #inita   the real include with its one uninitialised variable given the value a zero-initialising driver gives it
    #            (`bool a;` read by `if(a) discard;`, Canvas.cpp:271-273): equals #noproj
    "src/filter/ThermalErosion.cpp#noproj": "struct _Cv { Shader* projection_shader() { return Shader::builder().create(\"\",\"\"); } }; "
                                            "struct _Pj { _Cv* canvas; }; static _Cv _cv; static _Pj _pj{&_cv}; _Pj* p = &_pj;",
    "src/filter/ThermalErosion.cpp#inita": ("thermal", "inita"),
    # Morphological.cpp:57 splices the GLSL builtin name: "min" (Erode) or "max" (Dilate) (:25)
    "src/filter/Morphological.cpp#min": "std::string operation = \"min\";",
    "src/filter/Morphological.cpp#max": "std::string operation = \"max\";",
    "src/filter/tectonics/Tectonics.cpp": "",
}

# GradientNoise.cpp:441-455 builds its shader from two host-side values: `code` (one raw string per noise mode, chosen
# by the switch at :181-437) and `blendmode` (one of the static shaders of filter::blendMode(), Filter.cpp:170-207).
# One variant per (mode, blend) is lifted, with both values spliced in as text by gradient_noise_prelude().
NOISE_MODES = ["DEFAULT", "RIDGED", "BILLOWY", "IQNoise", "SWISS", "Jordan", "Plateaus"]
NOISE_BLENDS = ["replace", "add"]
for _m in NOISE_MODES:
    for _b in NOISE_BLENDS:
        if _b == "replace" or _m == "DEFAULT":
            FILES["src/filter/GradientNoise.cpp#%s_%s" % (_m, _b)] = ("noise", _m, _b)


def gradient_noise_prelude(ref, mode, blend):
    src = (ref / "src/filter/GradientNoise.cpp").read_text()
    m = re.search(r"case\s+%s\s*:\s*\{?\s*code\s*=\s*(R\"\(.*?\)\")\s*;" % mode, src, re.S)
    code = m.group(1)
    fsrc = (ref / "src/filter/Filter.cpp").read_text()
    bl = {name: expr for _, name, expr in extract(fsrc)}[blend]
    return "std::string code = %s; Shader* blendmode = %s;" % (code, bl)


def extract_returned(src, signature):
    """the `Shader::builder()...` expression a function of the reference returns: `signature` is the text of its head up to the
    opening parenthesis, e.g. "AbstractCanvas::projection_shader("; raw strings and comments are skipped over like in extract()"""
    at = src.index(signature)
    spans = scan(src)
    skip = [False] * (len(src) + 1)
    for _, a, b in spans:
        for k in range(a, b):
            skip[k] = True
    m = re.compile(r"return\s+(Shader::builder\(\))").search(src, at)
    a = m.start(1)
    depth, j = 0, m.end(1)
    while True:
        if not skip[j]:
            ch = src[j]
            if ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            elif ch == ";" and depth == 0:
                break
        j += 1
    return src[a:j]


def thermal_prelude(ref, init_a=False):
    proj = extract_returned((ref / "src/projections/Canvas.cpp").read_text(), "AbstractCanvas::projection_shader(")
    if init_a:
        assert proj.count("bool a;") == 1
        proj = proj.replace("bool a;", "bool a = false;")
    inv = extract_returned((ref / "src/projections/Equiretangular.cpp").read_text(), "Equiretangular::inverseShader(")
    return ("struct _Cv { Shader* inverseShader() { return %s; } Shader* projection_shader() { return %s; } }; "
            "struct _Pj { _Cv* canvas; }; static _Cv _cv; static _Pj _pj{&_cv}; _Pj* p = &_pj;" % (inv, proj))


def scan(src):
    """Yield (kind, start, end) for comments/strings so callers can skip them."""
    i, n = 0, len(src)
    spans = []
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            spans.append(("comment", i, j)); i = j
        elif src.startswith("/*", i):
            j = src.find("*/", i) + 2
            spans.append(("comment", i, j)); i = j
        elif src.startswith('R"', i):
            m = re.match(r'R"([^()\\ ]{0,16})\(', src[i:])
            delim = m.group(1)
            end = src.find(")" + delim + '"', i) + len(delim) + 2
            spans.append(("string", i, end)); i = end
        elif c == '"':
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            spans.append(("string", i, j + 1)); i = j + 1
        elif c == "'":
            j = i + 1
            while src[j] != "'":
                j += 2 if src[j] == "\\" else 1
            spans.append(("string", i, j + 1)); i = j + 1
        else:
            i += 1
    return spans


def extract(src):
    spans = scan(src)
    skip = [False] * (len(src) + 1)
    for _, a, b in spans:
        for k in range(a, b):
            skip[k] = True
    out = []
    for m in re.finditer(r"Shader::builder\(\)", src):
        a = m.start()
        if skip[a]:
            continue
        # forward to the terminating ';' at paren depth 0
        depth, j = 0, m.end()
        while True:
            if skip[j]:
                j += 1
                continue
            ch = src[j]
            if ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            elif ch == ";" and depth == 0:
                break
            j += 1
        expr = src[a:j]
        # backward to find "name =" on the same statement
        k = a - 1
        while k >= 0 and (skip[k] or src[k] not in ";{}"):
            k -= 1
        head = "".join(ch for idx, ch in enumerate(src[k + 1:a], k + 1) if not skip[idx])
        mm = re.search(r"(\w+)\s*=\s*$", head)
        if not mm or ".create(" not in expr:
            continue  # e.g. `return Shader::builder()...` helpers: not needed
        line = src.count("\n", 0, a) + 1
        out.append((line, mm.group(1), expr))
    return out


def main():
    ref = Path(sys.argv[1])
    dst = Path(sys.argv[2])
    parts = [
        "// GENERATED by oracle/refgl/extract_shaders.py from the reference tree -- DO NOT COMMIT.\n"
        "#include <map>\n#include <string>\n#include \"Shader.h\"\n"
    ]
    calls = []
    for rel, prelude in FILES.items():
        rel, _, variant = rel.partition("#")
        p = ref / rel
        if not p.exists():
            continue
        if isinstance(prelude, tuple):
            prelude = thermal_prelude(ref, len(prelude) > 1) if prelude[0] == "thermal" else gradient_noise_prelude(ref, prelude[1], prelude[2])
        src = p.read_text()
        items = extract(src)
        tag = re.sub(r"\W", "_", p.name + ("_" + variant if variant else ""))
        names = sorted({n for _, n, _ in items})
        body = [f"static void reg_{tag}(std::map<std::string, Shader*>& reg) {{", f"  {prelude}"]
        for n in names:
            body.append(f"  Shader* {n} = nullptr; (void){n};")
        for line, name, expr in items:
            body.append(f"  {name} = {expr};")
            body.append(f'  reg["{p.name}:{line}{("#" + variant) if variant else ""}"] = {name};')
        body.append("}\n")
        parts.append("\n".join(body))
        calls.append(f"  reg_{tag}(reg);")
    parts.append("void refgl_register_shaders(std::map<std::string, Shader*>& reg) {\n" + "\n".join(calls) + "\n}\n")
    dst.parent.mkdir(parents=True, exist_ok=True)
    dst.write_text("\n".join(parts))


if __name__ == "__main__":
    main()
