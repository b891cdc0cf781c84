"""Reads the numeric literals of the hot path OUT OF THE REFERENCE'S OWN TEXT and writes them to tests/golden/ref_constants.json.

Run in the build container (the only place /root/reference exists):   python tests/golden/pin_constants.py

Why: the reference is Rust and cannot be compiled here, and its own tests pin no pixel; the oracle (oracle/b32_oracle.c), the numpy
restatement (oracle/np_model.py) and the device code (bonnie-32_amd/csrc) each carry their own copy of the constants of the
algorithm.  This script makes the reference text itself -- not a retyped copy -- the pin for every one of them: each literal is
located by an anchored regular expression inside the function / constant that owns it, the file:line it was found at is recorded,
and tests assert equality on all three sides:
    tests/test_oracle_kats.py::test_oracle_constants_are_the_reference_text   (C oracle tap b32o_constants)
    tests/test_oracle_kats.py::test_np_model_constants_are_the_reference_text (oracle/np_model.py loads the fixture)
    tests/test_gpu_parity.py::test_device_constants_are_the_reference_text    (device tap b32_device_constants, runs on the GPU)
    tests/test_oracle_kats.py::test_fixture_is_current                        (re-derives the fixture when /root/reference is present)
The fixture holds numbers and where they came from -- no reference source text.
"""
import json
import os
import re
import struct
import sys

REF = os.environ.get("B32_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "src", "rasterizer")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_constants.json")


class Text:
    def __init__(self, rel):
        self.rel = rel
        self.lines = open(os.path.join(SRC, rel), encoding="utf-8").read().split("\n")

    def region(self, start_pat, end_pat=None, max_lines=400):
        """(first line index, last line index exclusive) of the item that starts at the first line matching start_pat: up to the first
        later line matching end_pat, or to the line where the brace depth opened on the start line returns to zero."""
        for i, ln in enumerate(self.lines):
            if re.search(start_pat, ln):
                break
        else:
            raise SystemExit(f"{self.rel}: anchor {start_pat!r} not found")
        if end_pat is not None:
            for j in range(i + 1, min(len(self.lines), i + max_lines)):
                if re.search(end_pat, self.lines[j]):
                    return i, j + 1
            raise SystemExit(f"{self.rel}: end {end_pat!r} not found after line {i + 1}")
        depth, seen = 0, False
        for j in range(i, min(len(self.lines), i + max_lines)):
            code = self.lines[j].split("//")[0]
            depth += code.count("{") - code.count("}")
            seen = seen or "{" in code
            if seen and depth == 0:
                return i, j + 1
        raise SystemExit(f"{self.rel}: unbalanced item at line {i + 1}")

    def find(self, rng, pat, group=1, nth=0):
        """nth match of pat inside the line range; returns (text of the group, 'file:line')."""
        k = 0
        for j in range(rng[0], rng[1]):
            code = self.lines[j].split("//")[0]
            for m in re.finditer(pat, code):
                if k == nth:
                    return m.group(group), f"src/rasterizer/{self.rel}:{j + 1}"
                k += 1
        raise SystemExit(f"{self.rel}:{rng[0] + 1}-{rng[1]}: pattern {pat!r} (match {nth}) not found")


def num(s):
    s = s.replace("_", "")
    s = re.sub(r"(u8|u16|u32|u64|i8|i16|i32|i64|usize|f32)$", "", s)
    if s.lower().startswith("0x") or s.lower().startswith("-0x"):
        return int(s, 16)
    return float(s) if ("." in s or "e" in s.lower()) else int(s)


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


FLOAT = r"(-?\d+\.\d+(?:e-?\d+)?)"
INT = r"(-?(?:0x[0-9A-Fa-f_]+|\d[\d_]*))"


def derive():
    out = {}

    def put(name, text_loc, kind=None):
        text, loc = text_loc
        v = num(text)
        e = {"value": v, "source": loc}
        if isinstance(v, float):
            e["f32_bits"] = f32_bits(v)          # what `const X: f32 = <literal>` holds
        out[name] = e
        return v

    fx = Text("fixed.rs")
    # ---- UNR table generator (fixed.rs: const UNR_TABLE)
    r = fx.region(r"const\s+UNR_TABLE\s*:")
    n_entries = put("unr.entries", fx.find(r, r"\[u8;\s*" + INT + r"\]"))
    add = put("unr.index_offset", fx.find(r, r"let\s+div\s*=\s*i\s*\+\s*" + INT))
    numer = put("unr.numerator", fx.find(r, r"let\s+quotient\s*=\s*" + INT + r"\s*/\s*div"))
    rnd = put("unr.round_add", fx.find(r, r"\(\s*quotient\s*\+\s*" + INT + r"\s*\)"))
    half = put("unr.round_div", fx.find(r, r"\(\s*quotient\s*\+\s*\d+\s*\)\s*/\s*" + INT))
    sub = put("unr.subtract", fx.find(r, r"as\s+i32\s*-\s*" + INT))
    loop_n = num(fx.find(r, r"while\s+i\s*<\s*" + INT)[0])
    assert loop_n == n_entries
    table = [max(0, (numer // (i + add) + rnd) // half - sub) for i in range(n_entries)]
    assert all(0 <= t <= 255 for t in table)
    out["unr.table"] = {"value": table, "source": out["unr.entries"]["source"], "note": "the generator loop above evaluated with its own literals"}
    # ---- Fixed32
    put("fixed.frac_bits", fx.find(fx.region(r"const\s+FRAC_BITS\s*:", r";"), r"=\s*" + INT))
    # ---- div_unr
    r = fx.region(r"pub\s+fn\s+div_unr\s*\(")
    put("div_unr.d16_shift", fx.find(r, r"d_norm\s*>>\s*" + INT))
    put("div_unr.index_bias", fx.find(r, r"d16\.wrapping_sub\(\s*" + INT + r"\s*\)"))
    put("div_unr.index_shift", fx.find(r, r"d16\.wrapping_sub\([^)]*\)\s*\)\s*>>\s*" + INT))
    put("div_unr.index_max", fx.find(r, r">>\s*\d+\s*\)\s*\.min\(\s*" + INT + r"\s*\)"))
    put("div_unr.u_add", fx.find(r, r"UNR_TABLE\[table_idx\]\s*as\s*u64\s*\+\s*" + INT))
    put("div_unr.nr1_const", fx.find(r, r"let\s+nr1\s*=\s*\(\s*" + INT + r"u64"))
    put("div_unr.nr1_shift", fx.find(r, r"let\s+nr1\s*=.*>>\s*" + INT))
    put("div_unr.nr2_const", fx.find(r, r"let\s+nr2\s*=\s*\(\s*" + INT + r"u64"))
    put("div_unr.nr2_shift", fx.find(r, r"let\s+nr2\s*=.*>>\s*" + INT))
    put("div_unr.shift_base", fx.find(r, r"let\s+shift\s*=\s*" + INT + r"u32\.wrapping_sub\(z\)"))
    # ---- project_to_screen
    r = fx.region(r"pub\s+fn\s+project_to_screen\s*\(")
    put("project_fixed.distance", fx.find(r, r"let\s+distance\s*=\s*Fixed32::from_f32\(\s*" + FLOAT))
    put("project_fixed.scale", fx.find(r, r"let\s+scale\s*=\s*Fixed32::from_f32\(\s*" + FLOAT))
    put("project_fixed.viewport_div", fx.find(r, r"as\s+f32\s*/\s*" + FLOAT + r"\s*\)"))
    put("project_fixed.viewport_frac", fx.find(r, r"/\s*\d+\.\d+\s*\)\s*\*\s*" + FLOAT))
    put("project_fixed.denom_guard", fx.find(r, r"denom\.0\.abs\(\)\s*<\s*" + INT))

    mt = Text("math.rs")
    put("near_plane", mt.find(mt.region(r"pub\s+const\s+NEAR_PLANE\s*:", r";"), r"=\s*" + FLOAT))
    r = mt.region(r"pub\s+fn\s+project\s*\(")
    put("project.distance", mt.find(r, r"const\s+DISTANCE\s*:\s*f32\s*=\s*" + FLOAT))
    put("project.viewport_frac", mt.find(r, r"const\s+SCALE\s*:\s*f32\s*=\s*" + FLOAT))
    put("project.us_sub", mt.find(r, r"let\s+us\s*=\s*ud\s*-\s*" + FLOAT))
    put("project.viewport_div", mt.find(r, r"as\s+f32\s*/\s*" + FLOAT + r"\s*\)\s*\*\s*SCALE"))
    put("project.denom_guard", mt.find(r, r"denom\.abs\(\)\s*<\s*" + FLOAT))

    rd = Text("render.rs")
    # ---- dither matrix
    r = rd.region(r"const\s+PS1_DITHER_MATRIX\s*:", r"^\];")
    rows = []
    for j in range(r[0] + 1, r[1]):
        vals = re.findall(r"-?\d+", rd.lines[j].split("//")[0])
        if vals:
            rows.append([int(v) for v in vals])
    assert len(rows) == 4 and all(len(x) == 4 for x in rows), rows
    out["dither.matrix"] = {"value": rows, "source": f"src/rasterizer/render.rs:{r[0] + 2}-{r[1] - 1}"}
    r = rd.region(r"fn\s+dither_and_quantize\s*\(")
    put("dither.shift", rd.find(r, r"\+\s*offset\s*\)\s*>>\s*" + INT))
    put("dither.clamp_lo", rd.find(r, r"\.clamp\(\s*" + INT + r"\s*,"))
    put("dither.clamp_hi", rd.find(r, r"\.clamp\(\s*\d+\s*,\s*" + INT + r"\s*\)"))
    r = rd.region(r"fn\s+apply_dither\s*\(")
    put("dither8.shift", rd.find(r, r"\+\s*offset\s*\)\s*>>\s*" + INT))
    put("dither8.clamp_hi", rd.find(r, r"\.clamp\(\s*\d+\s*,\s*" + INT + r"\s*\)"))
    put("dither8.expand_shift", rd.find(r, r"r5\s*<<\s*" + INT))
    # ---- expand_5_to_8
    r = rd.region(r"fn\s+expand_5_to_8\s*\(")
    put("expand5.shl", rd.find(r, r"v5\s*<<\s*" + INT))
    put("expand5.shr", rd.find(r, r"v5\s*>>\s*" + INT))
    # ---- blend_rgb555
    r = rd.region(r"fn\s+blend_rgb555\s*\(")
    put("blend555.in_shift", rd.find(r, r"front_r\s*>>\s*" + INT))
    put("blend555.average_div", rd.find(r, r"f_r5\s+as\s+u16\s*\)\s*/\s*" + INT))
    put("blend555.clamp_hi", rd.find(r, r"f_r5\s+as\s+u16\s*\)\s*\.min\(\s*" + INT))
    put("blend555.clamp_lo", rd.find(r, r"f_r5\s+as\s+i16\s*\)\s*\.max\(\s*" + INT))
    put("blend555.quarter_div", rd.find(r, r"f_r5\s+as\s+u16\s*/\s*" + INT))
    put("blend555.out_shift", rd.find(r, r"r5\s*<<\s*" + INT))
    # ---- rasterize_triangle_15
    r = rd.region(r"^fn\s+rasterize_triangle_15\s*\(", max_lines=600)
    put("fill.area_eps", rd.find(r, r"area\.abs\(\)\s*<\s*" + FLOAT))
    put("fill.err", rd.find(r, r"const\s+ERR\s*:\s*f32\s*=\s*" + FLOAT))
    put("fill.modulate_div", rd.find(r, r"vertex_r\s+as\s+u32\s*\)\s*/\s*" + INT))
    put("fill.modulate_max", rd.find(r, r"vertex_r\s+as\s+u32\s*\)\s*/\s*\d+\s*\)\s*\.min\(\s*" + INT))
    put("fill.shade_clamp_lo", rd.find(r, r"shade_r\.clamp\(\s*" + FLOAT))
    put("fill.shade_clamp_hi", rd.find(r, r"shade_r\.clamp\(\s*\d+\.\d+\s*,\s*" + FLOAT))
    put("fill.shade_max", rd.find(r, r"shade_r\.clamp\([^)]*\)\s*\)\s*\.min\(\s*" + FLOAT))
    put("fill.nodither_shift", rd.find(r, r"shaded_r8\s*>>\s*" + INT))
    # ---- 8-bit fill (render_mesh path): same tolerances, its own text
    r = rd.region(r"^fn\s+rasterize_triangle\s*\(", max_lines=600)
    put("fill8.area_eps", rd.find(r, r"area\.abs\(\)\s*<\s*" + FLOAT))
    put("fill8.err", rd.find(r, r"const\s+ERR\s*:\s*f32\s*=\s*" + FLOAT))
    # ---- render_mesh_15: DISTANCE added to the float camera depth
    r = rd.region(r"pub\s+fn\s+render_mesh_15\s*\(", max_lines=400)
    put("mesh.distance", rd.find(r, r"const\s+DISTANCE\s*:\s*f32\s*=\s*" + FLOAT))
    # ---- lighting
    r = rd.region(r"fn\s+shade_multi_light_color\s*\(")
    put("light.min_dist", rd.find(r, r"dist\s*<\s*" + FLOAT))
    put("light.color_div", rd.find(r, r"as\s+f32\s*/\s*" + FLOAT))
    put("light.total_max", rd.find(r, r"\.min\(\s*" + FLOAT + r"\s*\)"))

    ty = Text("types.rs")
    r = ty.region(r"impl\s+Color15\s*\{", max_lines=260)
    put("color15.transparent", ty.find(r, r"const\s+TRANSPARENT\s*:\s*Color15\s*=\s*Color15\(\s*" + INT))
    put("color15.black_drawable", ty.find(r, r"const\s+BLACK_DRAWABLE\s*:\s*Color15\s*=\s*Color15\(\s*" + INT))
    put("color15.white", ty.find(r, r"const\s+WHITE\s*:\s*Color15\s*=\s*Color15\(\s*" + INT))
    put("color15.semi_bit", ty.find(r, r"c\.0\s*\|=\s*" + INT))
    put("color15.r_shift", ty.find(r, r"r\.min\(\d+\)\s+as\s+u16\s*\)\s*<<\s*" + INT))
    put("color15.g_shift", ty.find(r, r"g\.min\(\d+\)\s+as\s+u16\s*\)\s*<<\s*" + INT))
    put("color15.channel_max", ty.find(r, r"r\.min\(\s*" + INT + r"\s*\)"))
    return out


def main():
    if not os.path.isdir(SRC):
        raise SystemExit(f"{SRC} not found: this script runs in the build container only")
    out = derive()
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    for k in sorted(out):
        if k != "unr.table":
            print(f"{k:28s} {out[k]['value']!r:>24}   {out[k]['source']}")
    print(f"unr.table: {len(out['unr.table']['value'])} entries, sum {sum(out['unr.table']['value'])}")


if __name__ == "__main__":
    sys.exit(main())
