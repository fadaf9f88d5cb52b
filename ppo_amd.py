"""Import alias: `import ppo_amd` loads the package directory `proximalpolicyoptimization.jl_amd/`
(whose name, mandated by the repo layout, is not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "proximalpolicyoptimization.jl_amd")
_spec = importlib.util.spec_from_file_location("ppo_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ppo_amd"] = _mod
_spec.loader.exec_module(_mod)
