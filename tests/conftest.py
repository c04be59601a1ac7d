import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(ROOT, "eventful-transformer_amd")
for p in (PKG_ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Every test session needs libevt_hip.so (CPU tests check its exports; GPU tests call it)."""
    sys.path.insert(0, PKG_ROOT)
    import build as evt_build  # eventful-transformer_amd/build.py

    import shutil

    lib = os.path.join(PKG_ROOT, "eventful_transformer", "libevt_hip.so")
    if evt_build.needs_build():
        # stale or missing: rebuild (a no-op when up to date).  Without hipcc a stale library is an error, not
        # something to test against silently.
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            evt_build.build()
        elif not os.path.exists(lib):
            raise RuntimeError("libevt_hip.so is missing and hipcc is not available")
        else:
            raise RuntimeError("libevt_hip.so is older than csrc/ and hipcc is not available to rebuild it")
    yield
