"""Import shim: the package directory is `mlx-audio-swift_amd/` (a hyphen is not importable),
so `import mlx_audio_swift_amd` loads that directory as a package under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mlx-audio-swift_amd")
_spec = importlib.util.spec_from_file_location(
    "mlx_audio_swift_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mlx_audio_swift_amd"] = _mod
_spec.loader.exec_module(_mod)
