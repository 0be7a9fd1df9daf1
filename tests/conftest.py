import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C-ABI library is a build product (git-ignored): build it once if a fresh checkout lacks it
    lib = os.path.join(REPO, "texpose_amd", "libtexpose_amd.so")
    if not os.path.exists(lib):
        import shutil
        import subprocess
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            subprocess.run(["make", "-C", os.path.join(REPO, "texpose_amd", "csrc"), "-j", "8"], check=False)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        v = z[k]
        if v.dtype == np.float32 or v.dtype == np.int64 or v.dtype == np.uint8:
            out[k] = torch.from_numpy(v) if v.ndim > 0 else v.item()
        elif v.ndim == 0:
            out[k] = v.item()
        else:
            out[k] = torch.from_numpy(v)
    return out


@pytest.fixture
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()


@pytest.fixture(autouse=True)
def _knobs_follow_the_environment():
    """texpose_amd.knobs parses the TP_* switches once; a test that flips one (monkeypatch.setenv + knobs.reload()) must not leak it:
    this fixture is set up before `monkeypatch` and so torn down after it has restored the environment."""
    yield
    from texpose_amd import knobs
    knobs.reload()
