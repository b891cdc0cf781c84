"""hipcc recipe for the gfx950 rasterizer library (bonnie-32_amd/csrc -> libb32raster.so).

Flags that matter for bit-exactness against the reference's Rust f32 semantics:
  -ffp-contract=off                 Rust never contracts a*b+c into an FMA
  (default) correctly rounded f32 divide and sqrt; no -ffast-math; f32 denormals are not flushed
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["b32_api.hip", "b32_scene.hip", "b32_frame.hip", "b32_batch.hip", "b32_setup.hip", "b32_sort.hip", "b32_bin.hip", "b32_fill.hip", "b32_wire.hip", "b32_sky.hip"]
OUT = os.path.join(CSRC, "libb32raster.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fno-gpu-flush-denormals-to-zero", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-Wall", "-Wno-unused-function"] + [f for f in os.environ.get("B32_EXTRA_FLAGS", "").split() if f]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "b32_device.h"), os.path.join(CSRC, "b32_host.h"),
                                                       os.path.join(_HERE, "..", "include", "b32raster.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, out=None, extra=()):
    """out / extra: an experiment build with more -D flags into another file (tools/exp_variants.py)"""
    if out is None and not force and not needs_build():
        return OUT
    cmd = [hipcc()] + FLAGS + list(extra) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out or OUT]
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
