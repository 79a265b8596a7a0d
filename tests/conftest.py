import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _library_is_built():
    """In-tree incremental build before anything loads the library (a no-op when the .so is newer than every source;
    hipcc cross-compiles gfx950 without a GPU).  The product path itself never builds: it fails loudly without the .so."""
    from neuspeech1_amd import build as b
    if os.path.exists(b.HIPCC):
        b.build(verbose=False)          # compile / link errors in modified sources must fail the run, never hide behind a stale .so
        return
    # no compiler on the box: the prebuilt library that travelled with the tree serves, but only if it is current
    if not os.path.exists(b.LIB):
        raise RuntimeError(f"{b.LIB} is missing and {b.HIPCC} does not exist")
    stale = b.stale_sources()
    if stale:
        raise RuntimeError(f"{b.LIB} is older than {stale} and {b.HIPCC} does not exist to rebuild it")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _ring_mode_from_env():
    """NS_RING=0/2/3 forces a GEMM kernel variant for the whole test session (default: auto dispatch)."""
    mode = os.environ.get("NS_RING")
    if mode is not None:
        from neuspeech1_amd import lib
        lib.load().ns_debug_set_ring(int(mode))
    yield
