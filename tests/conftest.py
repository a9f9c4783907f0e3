import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)          # repo root first: a HuggingFace `datasets` wheel is installed in this image

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `-m gpu`)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def synth_sd():
    from lip2speech_amd import synth
    return synth.synth_state_dict()


@pytest.fixture(scope="session", autouse=True)
def _launch_path_by_default():
    """The suite pins the launch-per-phase decode loop unless a test asks otherwise: models created here start with "persist_decode" = 0 (the library's
    default is 4: up to four clips take the persistent loop of pdecode.hip, which has its own tests, test_persistent_decode_*, on models that switch it on)."""
    try:
        from lip2speech_amd import native
        native.set_option("persist_decode", 0)
    except Exception:
        pass          # library not built: the tests that need it say so themselves
    yield
