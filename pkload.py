"""Import helper: the package directory is named `parakeet.cpp_amd` (not a valid
Python identifier), so it is loaded under the module name `parakeet_cpp_amd`."""
import importlib.util
import os
import sys

_NAME = "parakeet_cpp_amd"
ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "parakeet.cpp_amd")


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
