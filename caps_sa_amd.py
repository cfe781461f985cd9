"""Import shim: the package directory is named ``caps-sa_amd`` (not an identifier), so
``import caps_sa_amd`` loads it from that directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "caps-sa_amd")
_spec = importlib.util.spec_from_file_location("caps_sa_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["caps_sa_amd"] = _mod
_spec.loader.exec_module(_mod)
