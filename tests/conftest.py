import os
import sys

import pytest

# torch FIRST: its wheel bundles its own ROCm runtime; if libmi_speech.so (linked against /opt/rocm) were loaded
# before it, torch would bind to the wrong libamdhip64 and report "no GPU" (every -m gpu test silently skipped).
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False,
                     help="also run the tests marked slow (the >= 130 s full-depth parity tests; the final-tree GPU run uses this)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-depth parity variants (minutes of CPU oracle); run with --runslow or MIS_RUN_SLOW=1 - "
                                       "the default -m gpu set holds their 2- and 8-layer variants with the same gates")


def _has_gpu():
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        if torch is not None and torch.cuda.is_available():
            return True
    except Exception:
        pass
    try:                                   # ask the HIP library itself
        import mlx_audio_swift_amd as mas
        return mas._lib.lib().mis_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if not (config.getoption("--runslow") or os.environ.get("MIS_RUN_SLOW") == "1"):
        skip_slow = pytest.mark.skip(reason="slow full-depth variant: --runslow / MIS_RUN_SLOW=1 (its 2- and 8-layer variant runs by default)")
        for item in items:
            if "slow" in item.keywords:
                item.add_marker(skip_slow)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
