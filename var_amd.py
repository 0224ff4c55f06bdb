"""Import alias: the package directory is `voicecontrolledrobot-var_amd/` (not a valid Python
identifier), so `import var_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "voicecontrolledrobot-var_amd")
_spec = importlib.util.spec_from_file_location("var_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["var_amd"] = _mod
_spec.loader.exec_module(_mod)
