"""Import shim: makes the package directory `rust-raytracer_amd/` importable as `rust_raytracer_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rust-raytracer_amd")
_spec = importlib.util.spec_from_file_location("rust_raytracer_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rust_raytracer_amd"] = _mod
_spec.loader.exec_module(_mod)
