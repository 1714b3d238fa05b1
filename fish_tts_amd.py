"""Import alias: the package directory is `fish-tts_amd/` (not a valid identifier), so
`import fish_tts_amd` loads it from there under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fish-tts_amd")
_spec = importlib.util.spec_from_file_location(
    "fish_tts_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fish_tts_amd"] = _mod
_spec.loader.exec_module(_mod)
