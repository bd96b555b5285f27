import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "first_hardware_run(reason): a GPU test written after the round's GPU budget was spent -- the code path it "
                            "covers has compiled but never executed on a device; reported as xfail / XPASS (not strict) until a run on "
                            "hardware has been seen, then the marker is removed")


def pytest_collection_modifyitems(config, items):
    for item in items:
        m = item.get_closest_marker("first_hardware_run")
        if m is not None:
            reason = m.args[0] if m.args else m.kwargs.get("reason", "")
            item.add_marker(pytest.mark.xfail(strict=False, reason="not yet run on hardware: " + reason))


def _gpu_available():
    try:
        from recsys2019_deeplearning_evaluation_amd import _native
        return _native.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must FAIL (not skip) on a GPU box whose native library is broken; without any device they skip."""
    from recsys2019_deeplearning_evaluation_amd import _native
    _native.load()          # raises if libmi355rec.so is missing: no silent fallback
    if not _gpu_available():
        pytest.skip("no HIP device visible")
    return _native.device_name()
