import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """npz -> dict; uint8 blobs that were json-encoded by make_goldens.py are decoded."""
    import torch
    raw = np.load(os.path.join(GOLDEN, name))
    out = {}
    for k in raw.files:
        v = raw[k]
        if v.dtype == np.uint8 and k in ("cfg", "scene"):
            out[k] = json.loads(v.tobytes().decode())
        elif v.shape == () and k.startswith("seed"):
            out[k] = int(v)
        else:
            out[k] = torch.from_numpy(v)
    return out


def rel_max_err(y, ref):
    """max |y - ref| / max |ref| : the 'rel' of north_star's 1e-3 tolerance."""
    return float((y.double() - ref.double()).abs().max() / ref.double().abs().max())


@pytest.fixture(scope="session")
def golden():
    return load_golden
