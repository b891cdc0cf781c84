import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    """A live b32 context. Fails (does not skip) when the HIP library or device is missing: GPU tests must
    never pass on a fallback."""
    import __graft_entry__ as g
    g.build()
    from bonnie32_amd import rasterizer as R
    ctx = R.Context(0)
    ctx.set_fragment_counting(1)      # instrumented path: exact coverage + exact fragment-store counts (tests compare them)
    return ctx


@pytest.fixture(scope="session")
def keyed_ctx():
    """A context with the sort-free path switched off (b32_set_routes): frames take the keyed
    pipelines -- global painter's sort + EXACT coverage with fragment counting on, per-tile LDS sort + visibility buffer with it
    off, the keyed z-buffer kernel in z-buffer mode -- so those stay covered although the default path no longer needs them."""
    import __graft_entry__ as g
    g.build()
    from bonnie32_amd import rasterizer as R
    ctx = R.Context(0)
    ctx.set_routes(R.Context.ROUTE_SORT_FREE)
    ctx.set_fragment_counting(1)
    return ctx
