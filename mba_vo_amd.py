"""Import shim: makes the hyphenated package directory `mba-vo_amd/` importable as `mba_vo_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mba-vo_amd")
_spec = importlib.util.spec_from_file_location("mba_vo_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mba_vo_amd"] = _mod
_spec.loader.exec_module(_mod)
