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


def _tu_digest(src, extra):
    """What one translation unit is compiled from: its own text, every header of csrc/, the public header, the flags."""
    h = hashlib.sha256()
    h.update(open(os.path.join(CSRC, src), "rb").read())
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".h"):
            h.update(f.encode()); h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(_HERE, "..", "include", "b32raster.h"), "rb").read())
    h.update(" ".join(FLAGS + list(extra)).encode())
    return h.hexdigest()[:16]


def build(force=False, verbose=False, out=None, extra=()):
    """Every source is compiled to an object of its own, in parallel, and only when its translation-unit digest changed (objects under
    csrc/_obj/, git-ignored); the link always runs.  out / extra: an experiment build with more -D flags into another file
    (tools/exp_variants.py; its objects live in a directory of their own)."""
    if out is None and not force and not needs_build():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    cflags = [f for f in FLAGS if f != "-shared"]
    objdir = os.path.join(CSRC, "_obj" if out is None else "_obj_" + hashlib.sha256((out + " ".join(extra)).encode()).hexdigest()[:8])
    os.makedirs(objdir, exist_ok=True)
    digest = csrc_digest()

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        tag = obj + ".digest"
        want = _tu_digest(src, extra)
        # b32_api.hip carries the library digest (-DB32_SRC_DIGEST): it is recompiled whenever any source changes
        if src == "b32_api.hip":
            want += ":" + digest
        if not force and os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == want:
            return src, 0, "", ""
        cmd = [hipcc()] + cflags + list(extra) + (["-DB32_SRC_DIGEST=\"%s\"" % digest] if src == "b32_api.hip" else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            open(tag, "w").write(want)
        return src, r.returncode, " ".join(cmd), r.stdout + r.stderr

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    bad = False
    for src, rc, cmd, log in results:
        if verbose or rc or log.strip():
            print(cmd); print(log)
        bad = bad or rc != 0
    if bad:
        raise RuntimeError("hipcc failed")
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES] + ["-o", out or OUT]
    r = subprocess.run(link, capture_output=True, text=True)
    if verbose or r.returncode:
        print(" ".join(link)); print(r.stdout); print(r.stderr)
    if r.returncode:
        raise RuntimeError("hipcc link failed")
    return out or OUT


if __name__ == "__main__":
    build(force=True, verbose=True)
