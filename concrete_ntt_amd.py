"""Import alias: the package directory is `concrete-ntt_amd/` (not a valid Python identifier), so
`import concrete_ntt_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "concrete-ntt_amd")
_spec = importlib.util.spec_from_file_location(
    "concrete_ntt_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["concrete_ntt_amd"] = _mod
_spec.loader.exec_module(_mod)
