"""Import shim: `import hwmc` loads the package whose directory name mirrors the reference
repository (hyphens make it unimportable by name)."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "monte-carlo-simulation-of-hull-white-model-and-sensitivities-computation_amd")
_NAME = "hull_white_mc_amd"

if _NAME not in sys.modules:
    _spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_NAME] = _mod
    _spec.loader.exec_module(_mod)
_pkg = sys.modules[_NAME]
globals().update({k: getattr(_pkg, k) for k in _pkg.__all__})
PACKAGE_DIR = _PKG_DIR
LIB_PATH = _pkg._lib.LIB_PATH
SIGNATURES = _pkg._lib.SIGNATURES            # include/hw_mc.h
DIAG_SIGNATURES = _pkg._lib.DIAG_SIGNATURES  # include/hw_mc_diag.h


def _lib_abi_version():
    return _pkg._lib.ABI_VERSION
