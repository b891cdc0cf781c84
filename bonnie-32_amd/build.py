"""hipcc recipe for the gfx950 rasterizer library (bonnie-32_amd/csrc -> libb32raster.so).

Flags that matter for bit-exactness against the reference's Rust f32 semantics:
  -ffp-contract=off                 Rust never contracts a*b+c into an FMA
  (default) correctly rounded f32 divide and sqrt; no -ffast-math; f32 denormals are not flushed
"""
import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["b32_api.hip", "b32_scene.hip", "b32_frame.hip", "b32_batch.hip", "b32_setup.hip", "b32_sort.hip", "b32_bin.hip", "b32_fill.hip", "b32_shade.hip", "b32_blend.hip", "b32_wire.hip", "b32_sky.hip", "b32_gather.hip"]
OUT = os.path.join(CSRC, "libb32raster.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fno-gpu-flush-denormals-to-zero", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-Wall", "-Wno-unused-function"] + [f for f in os.environ.get("B32_EXTRA_FLAGS", "").split() if f]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def csrc_digest():
    """SHA-256 (16 hex digits) over everything the library is compiled from: the kernel / host sources and headers in csrc/, the public
    header and the compiler flags.  It is compiled INTO the library (-DB32_SRC_DIGEST, b32_build_digest()), so a built .so says which
    sources it came from: needs_build() compares the two instead of file times (the .so is git-ignored but travels to the GPU box),
    bench.py / smoke() refuse a library whose digest is not the tree's, and profiles/pmc_traffic.json is keyed by it."""
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(_HERE, "..", "include", "b32raster.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


DIGEST_MARK = b"B32-SRC-DIGEST:"


def built_digest(path=None):
    """The source digest a built library carries (read from the file, no dlopen); None when there is no library or no marker."""
    path = path or OUT
    if not os.path.exists(path):
        return None
    blob = open(path, "rb").read()
    i = blob.find(DIGEST_MARK)
    return blob[i + len(DIGEST_MARK):i + len(DIGEST_MARK) + 16].decode("ascii", "replace") if i >= 0 else None


def needs_build():
    return built_digest() != csrc_digest()


def build(force=False, verbose=False, out=None, extra=()):
    """out / extra: an experiment build with more -D flags into another file (tools/exp_variants.py)"""
    if out is None and not force and not needs_build():
        return OUT
    cmd = [hipcc()] + FLAGS + [f'-DB32_SRC_DIGEST="{csrc_digest()}"'] + list(extra) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out or OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode:
        print(" ".join(cmd))
        print(r.stdout)
        print(r.stderr)
    if r.returncode:
        raise RuntimeError("hipcc failed")
    return out or OUT


if __name__ == "__main__":
    build(force=True, verbose=True)
