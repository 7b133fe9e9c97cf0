"""Import alias: ``import hmvit_amd`` loads the package that lives in ``hm-vit_amd/`` (a hyphen
cannot appear in a Python module name)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hm-vit_amd")
_spec = importlib.util.spec_from_file_location(
    "hmvit_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hmvit_amd"] = _mod
_spec.loader.exec_module(_mod)
