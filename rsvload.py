"""Import helper: the package directory is named ``recursive-stwo_amd`` (not an importable identifier),
so it is loaded by path under the module name ``recursive_stwo_amd``."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "recursive-stwo_amd")


def load_package(lib_path=None):
    """lib_path: another build of the library than csrc/librsv_hip.so (the diagnostic build with permutation counters,
    tests/perm_census.py) — an explicit argument of the first load, not an environment variable."""
    name = "recursive_stwo_amd"
    if name in sys.modules:
        if lib_path and os.path.abspath(lib_path) != os.path.abspath(sys.modules[name].LIB_PATH):
            raise RuntimeError("recursive_stwo_amd is already loaded with another library")
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    if lib_path:
        mod._LIB_PATH_OVERRIDE = lib_path
    try:
        spec.loader.exec_module(mod)
    except Exception:
        del sys.modules[name]
        raise
    return mod
