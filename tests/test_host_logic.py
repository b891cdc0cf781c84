"""Host-side logic that needs no GPU: scene generator determinism, type mirrors, band partition."""
import hashlib

import numpy as np
import pytest

import bonnie32_amd as b32
from bonnie32_amd import parallel, scenegen


def test_splitmix64_reference_values():
    # splitmix64 with seed 0: the published first outputs of the reference implementation (Vigna)
    z = scenegen.splitmix64(0, 3)
    assert [int(x) for x in z] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    u = scenegen.uniform01(123, 1000)
    assert u.dtype == np.float32 and (u >= 0).all() and (u < 1).all()


def test_scene_shapes_and_defaults():
    sc = scenegen.make_scene("C1")
    assert (sc.width, sc.height, sc.n_tris) == (320, 240, 2000) and len(sc.vertices) == 6000
    assert sc.textures[0].width == 64 and sc.indexed_textures[0].clut.size == 16
    assert sc.settings.use_zbuffer is False and sc.settings.shading == b32.abi.SHADE_NONE and sc.settings.dithering
    assert (sc.vertices["uv"] >= -1).all() and (sc.vertices["uv"] < 2).all()
    c3 = scenegen.CONFIGS["C3"]
    assert (c3["width"], c3["height"], c3["n_tris"], c3["atlas"], c3["clut"]) == (2560, 1920, 1_000_000, 256, 256)


def test_reference_defaults_mirrored():
    s = b32.RasterSettings()                      # types.rs:1475-1495
    assert (s.affine_textures, s.use_zbuffer, s.shading, s.backface_cull, s.backface_wireframe) == (True, True, 2, True, True)
    assert (s.ambient, s.dithering, s.use_rgb555, s.use_fixed_point, s.xray_mode) == (0.3, True, True, True, False)
    d = s.lights[0].direction                     # Light::directional normalizes (types.rs:1318-1326)
    assert abs(d[0] + 0.57735026) < 1e-7 and s.lights[0].intensity == 0.7
    assert b32.RasterSettings.game().backface_wireframe is False      # types.rs:1455-1460
    from tests.golden.ref_fixtures import create_test_cube
    v, f = create_test_cube()                 # draw.rs:138-214
    assert len(v) == 24 and len(f) == 12 and (v["r"] == 128).all() and (f["texture_id"] == 0).all()
    assert tuple(f["v"][1]) == (0, 2, 3)
    t = b32.Texture15.checkerboard(8, 8, 1, 2)    # types.rs:702-711
    assert t.pixels[0] == 1 and t.pixels[4] == 2 and t.pixels[4 * 8] == 2


@pytest.mark.parametrize("h,n", [(1920, 8), (1920, 1), (240, 7), (5, 8), (241, 2)])
def test_band_rows_partition(h, n):
    bands = [parallel.band_rows(h, n, r) for r in range(n)]
    assert bands[0][0] == 0 and bands[-1][1] == h
    for a, b_ in zip(bands, bands[1:]):
        assert a[1] == b_[0]
    sizes = [b_[1] - b_[0] for b_ in bands]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == h
