"""The pin against the reference's own compiled code (tests/golden/wasm_pin/README.md).

The fixtures under tests/golden/wasm_pin/ are frames that `render_mesh` of /root/reference/docs/bonnie-engine.wasm -- the crate as
its authors compiled it, an older build than the source tree -- produced under node in the build container, and the outputs of that
module's `acosf`.  Here the CPU oracle (the restatement of TODAY's source) and the numpy restatement must reproduce them bit for bit;
the `-m gpu` tests do the same through the C ABI.  Nothing here reads /root/reference except `test_fixtures_are_current`, which is
skipped where the reference (or node) is absent."""
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from bonnie32_amd import scenefile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = os.path.join(ROOT, "tests", "golden", "wasm_pin")
MANIFEST = json.load(open(os.path.join(PIN, "manifest.json")))
NAMES = sorted(MANIFEST["scenes"])
WASM = "/root/reference/docs/bonnie-engine.wasm"


def load(name):
    sc = scenefile.read_scene(os.path.join(PIN, name + ".b32scene"))
    assert sc.fmt8 and sc.expect["sha256"] == MANIFEST["scenes"][name]["sha256"]
    return sc


def test_every_fixture_is_listed():
    assert sorted(os.path.basename(p)[:-9] for p in glob.glob(os.path.join(PIN, "*.b32scene"))) == NAMES and len(NAMES) >= 14


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_the_reference_binarys_frame(oracle, name):
    sc = load(name)
    fb = oracle.Framebuffer(sc.width, sc.height); fb.clear(sc.clear_color)
    rc, tm = oracle.render_mesh(fb, sc.vertices, sc.faces, sc.textures8, sc.camera, sc.settings)
    assert rc == 0 and tm.triangles_drawn > 0
    assert hashlib.sha256(np.asarray(fb.pixels).tobytes()).hexdigest() == sc.expect["sha256"]
    full = os.path.join(PIN, name + ".frame.npy")
    if os.path.exists(full):                              # the small scene's whole frame is committed, not only its hash
        assert np.array_equal(np.asarray(fb.pixels).reshape(sc.height, sc.width, 4), np.load(full))


@pytest.mark.parametrize("name", ["plain_64x48", "wire_overlay_160x120", "medium_160x120"])
def test_numpy_restatement_reproduces_the_reference_binarys_frame(name):
    from oracle import np_model as M
    sc = load(name)
    px = np.zeros(sc.width * sc.height * 4, np.uint8)
    px.reshape(-1, 4)[:] = [sc.clear_color.r, sc.clear_color.g, sc.clear_color.b, 255]
    M.render_mesh(px, sc.width, sc.height, sc.vertices, sc.faces, sc.textures8, sc.camera, sc.settings)
    assert hashlib.sha256(px.tobytes()).hexdigest() == sc.expect["sha256"]


def test_the_frames_are_not_trivial():
    """Every scene changes thousands of pixels, and the lit ones really depend on their lights (a frame that ignored them would
    hash the same with the lights removed)."""
    for name in NAMES:
        m = MANIFEST["scenes"][name]
        assert m["changed_pixels"] > (400 if m["width"] * m["height"] <= 64 * 48 else 1000), name


@pytest.mark.parametrize("name", ["flat_lights_320x240", "gouraud_spot_acos_320x240"])
def test_lit_frames_depend_on_their_lights(oracle, name):
    sc = load(name)
    sc.settings.lights = []
    fb = oracle.Framebuffer(sc.width, sc.height); fb.clear(sc.clear_color)
    oracle.render_mesh(fb, sc.vertices, sc.faces, sc.textures8, sc.camera, sc.settings)
    assert hashlib.sha256(np.asarray(fb.pixels).tobytes()).hexdigest() != sc.expect["sha256"]


def test_acosf_is_bit_identical_to_the_wasm32_targets(oracle):
    """f32::acos on the reference's shipping target = the module's own acosf; 9.5 k arguments incl. every branch point and the NaN
    domain.  Both restatements; the device's copy is checked in test_device_acos_is_the_wasm32_targets."""
    from oracle import np_model as M
    k = np.load(os.path.join(PIN, "acosf_kat.npz"))
    x, want = k["x_bits"].view(np.float32), k["acos_bits"]
    assert len(x) == MANIFEST["acosf_kat"]["n"] and hashlib.sha256(want.tobytes()).hexdigest() == MANIFEST["acosf_kat"]["sha256"]
    L = oracle.lib()
    got = np.array([L.b32o_acosf(float(v)) for v in x], np.float32)
    nan = np.isnan(want.view(np.float32))
    assert nan.sum() >= 5 and np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want[~nan])
    got2 = np.array([M.acosf(np.float32(v)) for v in x[::4]], np.float32)
    assert np.array_equal(np.isnan(got2), nan[::4]) and np.array_equal(got2.view(np.uint32)[~nan[::4]], want[::4][~nan[::4]])


@pytest.mark.skipif(not (os.path.exists(WASM) and shutil.which("node")), reason="needs the reference tree and node (build container only)")
def test_fixtures_are_current():
    """Re-runs the reference's module on every scene (and the acosf table) and compares with what is committed."""
    assert hashlib.sha256(open(WASM, "rb").read()).hexdigest() == MANIFEST["module_sha256"]
    r = subprocess.run([sys.executable, os.path.join(PIN, "make_vectors.py"), "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-1500:] + r.stderr[-1500:]


# ------------------------------------------------------------------------------------------------------------------ on the GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_reproduces_the_reference_binarys_frame(gpu_ctx, name):
    """b32_render_mesh (drop-in call, and a resident scene drawn twice) == the frame the reference's compiled render_mesh produced."""
    from bonnie32_amd import rasterizer as R
    sc = load(name)
    fb = R.Framebuffer(sc.width, sc.height, gpu_ctx)
    fb.clear(sc.clear_color)
    tm = R.render_mesh(fb, sc.vertices, sc.faces, sc.textures8, sc.camera, sc.settings)
    assert tm.triangles_drawn > 0
    assert hashlib.sha256(np.asarray(fb.pixels).tobytes()).hexdigest() == sc.expect["sha256"]
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, textures8=sc.textures8)
    for _ in range(2):
        fb.clear(sc.clear_color)
        rs.render(sc.camera, sc.settings)
        assert hashlib.sha256(np.asarray(fb.pixels).tobytes()).hexdigest() == sc.expect["sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["plain_320x240", "gouraud_spot_acos_320x240", "rotated_camera_320x240", "hostile_320x240"])
def test_gpu_keyed_routes_reproduce_the_reference_binarys_frame(keyed_ctx, name):
    from bonnie32_amd import rasterizer as R
    sc = load(name)
    fb = R.Framebuffer(sc.width, sc.height, keyed_ctx)
    fb.clear(sc.clear_color)
    R.render_mesh(fb, sc.vertices, sc.faces, sc.textures8, sc.camera, sc.settings)
    assert hashlib.sha256(np.asarray(fb.pixels).tobytes()).hexdigest() == sc.expect["sha256"]


@pytest.mark.gpu
def test_device_acos_is_the_wasm32_targets(gpu_ctx):
    k = np.load(os.path.join(PIN, "acosf_kat.npz"))
    x, want = k["x_bits"].view(np.float32).copy(), k["acos_bits"]
    got = gpu_ctx.selftest_f32(4, x, x, x)
    nan = np.isnan(want.view(np.float32))
    assert np.array_equal(np.isnan(got), nan) and np.array_equal(got.view(np.uint32)[~nan], want[~nan])
