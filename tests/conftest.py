"""Shared pytest configuration.

``-m "not gpu"`` runs in the CPU-only build container (oracle vs goldens, host logic, C-ABI
symbol check); ``-m gpu`` runs on the MI355X box and calls the HIP path through the C-ABI.
"""
import sys
from pathlib import Path

import pyarrow  # noqa: F401  (imported first: pandas' lazy import of it failed once on a GPU box after torch)
import pyarrow.parquet  # noqa: F401
import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """The built C-ABI library (builds it in-tree when missing; hipcc cross-compiles on CPU)."""
    import __graft_entry__ as entry
    from semantic_search_kd_amd import _native

    if not _native.lib_path().exists():
        entry.build()
    return _native.load()


@pytest.fixture(scope="session")
def gpu(native_lib):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("test is marked gpu but no HIP device is visible")
    return torch.device("cuda:0")
