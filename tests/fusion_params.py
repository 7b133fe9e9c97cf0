"""The parameter containers of hm-vit_amd/fusion.py without the HIP library: fusion.py's classes are imported from source with
`_lib` stubbed, for CPU tests that only need the module's parameter names and shapes (state_dict contract, DDP)."""
import importlib.util
import os
import sys
import types


def parameter_skeleton(cfg):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg_dir = os.path.join(root, "hm-vit_amd")
    pkg = types.ModuleType("_hmvit_cpu")
    pkg.__path__ = [pkg_dir]
    sys.modules["_hmvit_cpu"] = pkg
    lib = types.ModuleType("_hmvit_cpu._lib")
    lib.NUM_TYPES, lib.PREC_F32, lib.PREC_F16, lib.PREC_SPLIT, lib.PREC_MIXED = 2, 0, 1, 2, 3
    sys.modules["_hmvit_cpu._lib"] = lib
    for name in ("weights", "fusion"):
        spec = importlib.util.spec_from_file_location(f"_hmvit_cpu.{name}", os.path.join(pkg_dir, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"_hmvit_cpu.{name}"] = mod
        spec.loader.exec_module(mod)
    return sys.modules["_hmvit_cpu.fusion"].HeteroFusion(cfg)
