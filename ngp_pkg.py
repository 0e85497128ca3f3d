"""Path-based loader for the `nextgp.jl_amd/` package (its directory name is not a valid module name)."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(_ROOT, "nextgp.jl_amd")
MODNAME = "nextgp_jl_amd"


def load_pkg():
    if MODNAME in sys.modules:
        return sys.modules[MODNAME]
    spec = importlib.util.spec_from_file_location(MODNAME, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[MODNAME] = mod
    spec.loader.exec_module(mod)
    return mod
