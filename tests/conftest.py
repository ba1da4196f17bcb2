import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
for p in (ROOT / "pnmol-experiments_amd", ROOT / "oracle", ROOT):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_ctx():
    """Loads libpnmol_hip.so and opens device 0; fails loudly (no fallback) if either is missing."""
    from pnmol import _hip

    return _hip.Context.default(0)
