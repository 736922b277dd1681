"""Import alias: the package directory is `me-trpo_amd/` (not a valid Python identifier), so
`import metrpo_amd` loads that directory as the package `metrpo_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "me-trpo_amd")
_spec = importlib.util.spec_from_file_location("metrpo_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["metrpo_amd"] = _mod
_spec.loader.exec_module(_mod)
