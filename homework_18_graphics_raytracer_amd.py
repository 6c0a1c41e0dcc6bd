"""Import shim: the package directory is named ``homework-18-graphics-raytracer_amd`` (not a
valid Python identifier), so ``import homework_18_graphics_raytracer_amd`` loads it from there."""
import importlib.util
import sys
from pathlib import Path

_dir = Path(__file__).resolve().parent / "homework-18-graphics-raytracer_amd"
_spec = importlib.util.spec_from_file_location(
    __name__, _dir / "__init__.py", submodule_search_locations=[str(_dir)]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
