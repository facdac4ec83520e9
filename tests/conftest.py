import glob
import json
import os
import sys

import numpy as np
import pytest

# Every buffer the kernels are expected to fill (outputs, workspace) starts as NaN bit patterns in the tests
# (disenlink_amd/ops.py, DL_POISON): a read of memory nobody wrote fails a test instead of passing or failing with
# whatever the allocator happened to hand out.  Must be set before disenlink_amd.ops is imported.
os.environ.setdefault("DL_POISON", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def lib_env(monkeypatch):
    """Set (value) or unset (None) one of the LIBRARY's environment switches inside a test: the library reads its
    switches once (csrc/dl_config.h), so every change is followed by dl_config_reload(); the environment and the
    library's view of it are restored when the test ends."""
    from disenlink_amd import _lib

    def change(name, value=None):
        if value is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, str(value))
        _lib.config_reload()
    yield change
    monkeypatch.undo()
    _lib.config_reload()


def golden_case_names():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "case_*.npz")))


def trajectory_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "traj_*.npz")))


def load_trajectory(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"), allow_pickle=False))
    g["meta"] = json.loads(str(g["meta"]))
    return g


def load_golden(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, f"case_{name}.npz"), allow_pickle=False))
    g["meta"] = json.loads(str(g["meta"]))
    return g


@pytest.fixture(params=golden_case_names())
def golden(request):
    return load_golden(request.param)
