"""`.b32scene` (bonnie-32_amd/scenefile.py): the file format that carries one render_mesh_15 / render_mesh call to the CPU oracle, to the
GPU library (tests/cpp/mesh_harness.cpp) and to the Rust harness that would pin the oracle against the reference (tests/rust/pin_oracle).
CPU side: every golden scene survives write -> read unchanged, the oracle rendering FROM THE FILE reproduces tests/golden/hashes.json,
and the committed sample files are current."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from bonnie32_amd import scenefile
from tests.golden.make_golden import SCENES, SCENES8

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HASHES = json.load(open(os.path.join(GOLD, "hashes.json")))
SMALL = [n for n in list(SCENES) + list(SCENES8) if n not in ("C3", "C5", "C3:100k", "C2", "C2:blend", "8:C2", "C5:20k")]


def _render_from(sc, oracle):
    fb = oracle.Framebuffer(sc.width, sc.height); fb.clear(sc.clear_color)
    if sc.fmt8:
        rc, tm = oracle.render_mesh(fb, sc.vertices, sc.faces, sc.textures8, sc.camera, sc.settings)
    else:
        rc, tm = oracle.render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, sc.fog)
    assert rc == 0
    return fb, tm


@pytest.mark.parametrize("name", SMALL)
def test_scene_file_round_trip_and_oracle_from_file(oracle, tmp_path, name):
    sc = (SCENES.get(name) or SCENES8[name])()
    p = str(tmp_path / "s.b32scene")
    scenefile.write_scene(p, sc, expect=HASHES[name])
    got = scenefile.read_scene(p)
    assert (got.width, got.height) == (sc.width, sc.height) and got.clear_color == sc.clear_color
    assert np.array_equal(got.vertices, sc.vertices) and np.array_equal(got.faces, sc.faces)
    assert got.camera == sc.camera or all(np.array_equal(np.float32(getattr(got.camera, k)), np.float32(getattr(sc.camera, k)))
                                          for k in ("position", "basis_x", "basis_y", "basis_z"))
    a, b = got.settings, sc.settings
    for k in ("affine_textures", "use_zbuffer", "shading", "backface_cull", "backface_wireframe", "dithering", "wireframe_overlay",
              "use_rgb555", "use_fixed_point", "xray_mode"):
        assert getattr(a, k) == getattr(b, k), k
    assert np.float32(a.ambient) == np.float32(b.ambient) and len(a.lights) == len(b.lights)
    assert (a.ortho_projection is None) == (b.ortho_projection is None)
    src_tex = sc.textures8 if got.fmt8 else sc.textures
    dst_tex = got.textures8 if got.fmt8 else got.textures
    assert len(src_tex) == len(dst_tex) and all(np.array_equal(x.pixels, y.pixels) and x.blend_mode == y.blend_mode for x, y in zip(src_tex, dst_tex))
    assert got.expect["sha256"] == HASHES[name]["sha256"]
    fb, tm = _render_from(got, oracle)                         # lights, fog, ortho, every setting came through the file
    assert hashlib.sha256(fb.pixels).hexdigest() == HASHES[name]["sha256"]
    assert hashlib.sha256(fb.zbuffer.tobytes()).hexdigest() == HASHES[name]["zbuffer_sha256"]
    assert (tm.triangles_drawn, tm.fragments) == (HASHES[name]["triangles_drawn"], HASHES[name]["fragments"])


def test_committed_scene_files_are_current(oracle):
    man = json.load(open(os.path.join(GOLD, "scenes", "manifest.json")))
    assert set(man) >= {"cube", "C1"}
    for name, e in man.items():
        path = os.path.join(GOLD, "scenes", e["file"])
        assert hashlib.sha256(open(path, "rb").read()).hexdigest() == e["file_sha256"], name
        sc = scenefile.read_scene(path)
        assert sc.expect["sha256"] == HASHES[name]["sha256"] == e["sha256"]
        fb, tm = _render_from(sc, oracle)
        assert hashlib.sha256(fb.pixels).hexdigest() == sc.expect["sha256"] and tm.triangles_drawn == sc.expect["triangles_drawn"]


def test_cpp_scene_reader_compiles():
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "bonnie-32_amd", "host"), "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "mesh_harness.cpp")], check=True)


def test_rust_pin_harness_is_present_and_cites_the_reference():
    """tests/rust/pin_oracle: Cargo crate SOURCE (no Rust toolchain in this image: not compiled here) that #[path]-includes the
    reference's rasterizer modules, reads .b32scene files and prints what tests/golden/hashes.json must say."""
    d = os.path.join(ROOT, "tests", "rust", "pin_oracle")
    main_rs = open(os.path.join(d, "src", "main.rs")).read()
    for needle in ("render_mesh_15", "B32SCENE", "#[path", "src/rasterizer"):
        assert needle in main_rs
    assert "macroquad" in open(os.path.join(d, "Cargo.toml")).read() or "get_time" in main_rs
