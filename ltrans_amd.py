"""Import alias for the package directory ``ltransv.2b_amd/`` (whose name, taken
from the reference repository, is not a Python identifier)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
_pkg_dir = _os.path.join(_here, "ltransv.2b_amd")
_spec = _ilu.spec_from_file_location(
    "ltrans_amd", _os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["ltrans_amd"] = _mod
_spec.loader.exec_module(_mod)
