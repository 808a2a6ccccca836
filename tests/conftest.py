import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Builds whatever is missing (HIP libs cross-compile without a GPU)."""
    from madrona_amd import build
    from madrona_amd.simlib import HIP_BUILD_DIR, REF_BUILD_DIR

    if not os.path.exists(os.path.join(HIP_BUILD_DIR, "libmadrona_hip.so")):
        build.build_hip()
    if not os.path.exists(os.path.join(REF_BUILD_DIR, "liboracle_restate.so")):
        build.build_oracle()
    return True


def ref_available(sim: str) -> bool:
    from madrona_amd.simlib import ref_lib_path
    return os.path.exists(ref_lib_path(sim))
