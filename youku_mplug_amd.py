"""Import alias: the package directory is `youku-mplug_amd/` (not a valid Python identifier);
`import youku_mplug_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "youku-mplug_amd")
_spec = importlib.util.spec_from_file_location(
    "youku_mplug_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["youku_mplug_amd"] = _mod
_spec.loader.exec_module(_mod)
