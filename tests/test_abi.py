"""The C-ABI library: builds for gfx950, loads, exports every symbol include/b32raster.h declares, and its PODs have
the layout the ctypes / numpy mirrors assume.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "b32raster.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from bonnie32_amd import abi
    return abi.load_library()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(b32_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported(lib):
    from bonnie32_amd import abi
    names = declared_functions()
    assert len(names) >= 20
    bound = {n for n, _, _ in abi.SYMBOLS}
    assert set(names) == bound, "abi.SYMBOLS must bind exactly what the header declares"
    for n in names:
        assert hasattr(lib, n), n


def test_pod_layouts_match_c(lib):
    from bonnie32_amd import abi
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "b32raster.h"
int main(void){
 printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(B32Vertex), sizeof(B32Face), sizeof(B32Texture15), sizeof(B32IndexedTexture),
   sizeof(B32Camera), sizeof(B32Light), sizeof(B32Settings), sizeof(B32Fog), sizeof(B32Timings));
 printf("%zu %zu %zu %zu\n", offsetof(B32Settings, ambient), offsetof(B32Settings, lights), offsetof(B32Timings, fragments), offsetof(B32Vertex, r));
 printf("%zu %zu %zu\n", sizeof(B32MeshParams), offsetof(B32MeshParams, has_fog), offsetof(B32MeshParams, fog));
 return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")], check=True)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()
    sizes = [int(x) for x in out]
    assert sizes[:9] == [abi.VERTEX_DTYPE.itemsize, abi.FACE_DTYPE.itemsize, C.sizeof(abi.B32Texture15), C.sizeof(abi.B32IndexedTexture),
                         C.sizeof(abi.B32Camera), C.sizeof(abi.B32Light), C.sizeof(abi.B32Settings), C.sizeof(abi.B32Fog), C.sizeof(abi.B32Timings)]
    assert sizes[9:13] == [abi.B32Settings.ambient.offset, abi.B32Settings.lights.offset, abi.B32Timings.fragments.offset,
                           abi.VERTEX_DTYPE.fields["r"][1]]
    assert sizes[13:] == [C.sizeof(abi.B32MeshParams), abi.B32MeshParams.has_fog.offset, abi.B32MeshParams.fog.offset]


def test_no_cpu_fallback(lib):
    """Without a HIP device b32_create must fail loudly (B32_E_NO_DEVICE); with one it must succeed."""
    import torch
    from bonnie32_amd import abi
    h = C.c_void_p()
    rc = lib.b32_create(0, C.byref(h))
    if torch.cuda.is_available():
        assert rc == abi.B32_OK
        lib.b32_destroy(h)
    else:
        assert rc == abi.B32_E_NO_DEVICE and not h.value
    assert lib.b32_strerror(abi.B32_E_NO_DEVICE).decode().startswith("no HIP device")


def test_product_never_imports_oracle():
    """The product package must not reference the oracle (tests/bench/smoke are the only allowed users)."""
    pkg = os.path.join(ROOT, "bonnie-32_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "b32o_" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_cpp_host_mirror_compiles():
    """The C++ mirror of the reference interface (host/rasterizer.hpp) is header-only over the C ABI."""
    hpp = os.path.join(ROOT, "bonnie-32_amd", "host", "rasterizer.hpp")
    if not os.path.exists(hpp):
        pytest.skip("host mirror not present")
    src = ('#include "rasterizer.hpp"\nint main(){ b32::RasterSettings s = b32::RasterSettings::game(); '
           'std::vector<std::pair<const b32::ResidentMesh*, b32::MeshParams>> m; (void)&b32::render_frame; return s.use_zbuffer && m.empty() ? 0 : 1; }\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(src)
        subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.dirname(hpp), "-I", os.path.join(ROOT, "include"),
                        os.path.join(d, "t.cpp")], check=True)
