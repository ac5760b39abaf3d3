"""Import shim: `import rpgp_amd` loads the package that lives in `randomly-projected-additive-gps_amd/`
(the directory name required by the build contract is not a valid Python identifier)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "randomly-projected-additive-gps_amd")
_spec = _ilu.spec_from_file_location("rpgp_amd", _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["rpgp_amd"] = _mod
_spec.loader.exec_module(_mod)
