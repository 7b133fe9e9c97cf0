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


# Order of the GPU suite: whatever budget the driver's run has, every SURVEY 8 row's oracle / golden parity
# test runs BEFORE the stress, sweep, soak and reproducibility tests, so a timeout can only cost duplicated evidence.
# Files in row order: fusion (a1-a13), operators, encoders (a14-a16), model (a18, f1), post-processing (f1), pillariser (f2),
# gradients (configs[4]), trainer (f3), camera lift (a17), FAX (f4), AP replay; then camera training, range stress, multi-GPU.
_FILE_ORDER = ["test_hip_fusion.py", "test_hip_ops.py", "test_hip_encoder.py", "test_hip_model.py", "test_hip_post.py",
               "test_hip_vox.py", "test_hip_train.py", "test_hip_trainer.py", "test_hip_cvt.py", "test_hip_camera.py",
               "test_hip_fax.py", "test_hip_ap.py", "test_hip_camera_train.py", "test_hip_range.py", "test_hip_multigpu.py"]
_LATE = ("random_sweep", "reproducible", "dynamic_range", "under_input_and_weight_scaling", "sharpening", "soak",
         "independent_of_the_scale", "finite_differences", "edge_cases", "graph_capturable")


def pytest_collection_modifyitems(config, items):
    def key(pair):
        i, item = pair
        fname = os.path.basename(str(item.fspath))
        late = any(t in item.name for t in _LATE)
        rank = _FILE_ORDER.index(fname) if fname in _FILE_ORDER else -1      # CPU-side files keep their place in front
        return (1 if late else 0, rank, i)
    items[:] = [it for _, it in sorted(enumerate(items), key=key)]


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


def p999_err(y, ref):
    """99.9-percentile of the ELEMENT-WISE relative error |y - ref| / max(|ref|, 1e-3 rms(ref)) - the measure of
    tests/test_hip_fusion.py::full_size_report: what the max-normalised figure cannot see, a regression confined to
    small-magnitude outputs."""
    y, ref = y.double().reshape(-1), ref.double().reshape(-1)
    floor = 1e-3 * float(ref.pow(2).mean().sqrt())
    e = (y - ref).abs() / ref.abs().clamp_min(floor)
    k = max(1, int(round(0.999 * e.numel())))
    return float(e.kthvalue(k).values)


@pytest.fixture(scope="session")
def golden():
    return load_golden
