"""Writes every golden scene of tests/golden/make_golden.py (SCENES, SCENES8 -- BASELINE's C1, C2, C3, C5 and all parity variants) as
`.b32scene` files (bonnie-32_amd/scenefile.py) with the expectation record from tests/golden/hashes.json (frame and depth-buffer
SHA-256, triangles_drawn, fragment stores), plus manifest.json (file SHA-256 per scene).  These files are the inputs of
tests/rust/pin_oracle (the reference itself, once a Rust toolchain exists) and of tests/cpp/mesh_harness.cpp (the GPU library).
usage: python tools/export_scenes.py [out_dir = gpurun_out/scenes] [--small] [name ...]        (--small: skip the 1 M-triangle C3 / C5)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bonnie32_amd import scenefile  # noqa: E402
from tests.golden.make_golden import SCENES, SCENES8  # noqa: E402


def file_name(scene_name):
    return scene_name.replace(":", "_") + ".b32scene"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    small = "--small" in sys.argv
    out = args[0] if args else os.path.join(ROOT, "gpurun_out", "scenes")
    want = set(args[1:])
    os.makedirs(out, exist_ok=True)
    H = json.load(open(os.path.join(ROOT, "tests", "golden", "hashes.json")))
    manifest = {}
    for name, mk in {**SCENES, **SCENES8}.items():
        if (want and name not in want) or (small and name in ("C3", "C5")):
            continue
        sc = mk()
        path = os.path.join(out, file_name(name))
        digest = scenefile.write_scene(path, sc, expect=H[name])
        manifest[name] = {"file": file_name(name), "file_sha256": digest, "bytes": os.path.getsize(path), **{k: H[name][k] for k in ("sha256", "zbuffer_sha256", "triangles_drawn", "fragments", "width", "height")}}
        print(f"{name:28s} {os.path.getsize(path):>11,d} B  frame {H[name]['sha256'][:16]}")
    json.dump(manifest, open(os.path.join(out, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
