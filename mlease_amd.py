"""Import shim: exposes the package directory ``ml-ease_amd/`` (hyphenated per the repo layout
contract) under the importable module name ``mlease_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ml-ease_amd")
_spec = importlib.util.spec_from_file_location(
    "mlease_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mlease_amd"] = _mod
_spec.loader.exec_module(_mod)
