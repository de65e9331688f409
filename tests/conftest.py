import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def lib_built():
    """The in-tree HIP library, built from THIS tree: build it if it is missing, and rebuild it when its stamped source hash
    (iwae_build_id) is not the hash of the sources next to it -- the .so is git-ignored and ships prebuilt, nothing else ties the
    binary under test to the tree (round-4 verdict).  hipcc cross-compiles without a GPU."""
    from iwae_amd import _capi
    want = _capi.source_build_id()
    have = _capi.file_build_id()
    if have != want:
        import __graft_entry__
        __graft_entry__.build()
        have = _capi.file_build_id()
        assert have == want, "libiwae_amd.so does not match its sources after a rebuild: %r vs %r" % (have, want)
    return _capi.LIB_PATH


@pytest.fixture(scope="session")
def gpu(lib_built):
    if not _have_gpu():
        pytest.skip("no GPU in this container")
    return True
