import importlib.util
import json
import os
import sys

import pytest

try:
    # Some GPU tests use torch (synthetic data on the device, process groups) next to the engine.  torch ships its own
    # copy of the HIP runtime; whichever libamdhip64 is loaded first serves the whole process, and torch does not find
    # its device when that is the system copy pulled in by libcusk_hip.so.  Loading torch's first works for both.
    import torch  # noqa: F401
except Exception:  # the engine itself does not need torch
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLDEN, "ref_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(GOLDEN, "files")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def synth():
    import cigwas_amd.synth as S

    return S
