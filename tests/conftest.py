"""pytest configuration.

Markers
  gpu : needs an MI355X; run with ``-m gpu`` on the GPU box.  Everything else runs on CPU.

The oracle (oracle/) is the checker; the product is libpdt.so (HIP).  GPU tests call the
product only through the C ABI (ctypes binding in project-desert-tortoise_amd/__init__.py).
"""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) GPU")


@pytest.fixture(scope="session")
def pdt():
    mod = importlib.import_module("project-desert-tortoise_amd")
    if not os.path.exists(mod.LIBSYNTH_PATH):
        subprocess.run(["make", "-C", ROOT, mod.LIBSYNTH_PATH[len(ROOT) + 1:]], check=True, capture_output=True)
    return mod


@pytest.fixture(scope="session")
def orc():
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        meta = json.load(f)
    meta["dir"] = GOLDEN
    return meta


@pytest.fixture(scope="session")
def clip(pdt):
    rate, iq = pdt.read_wav(os.path.join(GOLDEN, "5sec_clip.wav"))
    assert rate == 50000 and iq.shape == (250195, 2)
    return rate, iq


def golden_text(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def gpu_available(pdt):
    if not os.path.exists(pdt.LIBPDT_PATH):
        return False
    try:
        return pdt.lib().pdt_device_count() > 0
    except Exception:
        return False


def bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()
