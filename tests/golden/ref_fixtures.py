"""Reference-authored fixtures (test data, not product code): the mesh of rasterizer/draw.rs:138-214 `create_test_cube` as a vertex /
face table, and the scene the golden cube frame is rendered from."""
import numpy as np

import bonnie32_amd as b32
from bonnie32_amd import scenegen
from bonnie32_amd.rtypes import make_faces, make_vertices


def create_test_cube():
    """rasterizer/draw.rs:138-214: the reference's own fixture mesh (24 vertices, 12 faces, texture 0)."""
    positions = np.array([
        [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1],
        [-1, -1, -1], [-1, 1, -1], [1, 1, -1], [1, -1, -1],
        [-1, 1, -1], [-1, 1, 1], [1, 1, 1], [1, 1, -1],
        [-1, -1, -1], [1, -1, -1], [1, -1, 1], [-1, -1, 1],
        [1, -1, -1], [1, 1, -1], [1, 1, 1], [1, -1, 1],
        [-1, -1, -1], [-1, -1, 1], [-1, 1, 1], [-1, 1, -1]], dtype=np.float32)
    normals = np.array([[0, 0, 1], [0, 0, -1], [0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0]], dtype=np.float32)
    uvs = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32)
    v = make_vertices(24)
    f = make_faces(12, texture_id=0)
    for face_idx in range(6):
        for i in range(4):
            k = face_idx * 4 + i
            v["pos"][k] = positions[k]
            v["uv"][k] = uvs[i]
            v["normal"][k] = normals[face_idx]
        b = face_idx * 4
        f["v"][face_idx * 2] = (b, b + 1, b + 2)
        f["v"][face_idx * 2 + 1] = (b, b + 2, b + 3)
    return v, f


def cube_scene(width=320, height=240):
    """The reference-authored fixture: create_test_cube (draw.rs:138-214) + Texture15::checkerboard
    (types.rs:702-711), camera pulled back along -z, Gouraud default light, painter's."""
    v, f = create_test_cube()
    tex = b32.Texture15.checkerboard(32, 32, 0x7FFF, 0x3DEF)
    s = b32.RasterSettings.benchmark()
    s.shading = b32.abi.SHADE_GOURAUD
    s.lights = [b32.Light.directional((-1.0, -1.0, -1.0), 0.7)]
    cam = b32.Camera(position=(0.7, -0.9, -4.5))
    return scenegen.Scene("cube", width, height, v, f, [tex], [], cam, s)
