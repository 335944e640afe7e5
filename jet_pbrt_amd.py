"""Import shim: the package directory is `jet-pbrt_amd/` (hyphen, as the build contract names it), which is
not a valid Python identifier.  `import jet_pbrt_amd` loads that directory as a regular package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jet-pbrt_amd")
_spec = importlib.util.spec_from_file_location("jet_pbrt_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["jet_pbrt_amd"] = _mod
_spec.loader.exec_module(_mod)
