"""Import shim: makes the on-disk package directory `sketchlib.rust_amd/` importable
as the Python module `sketchlib.rust_amd` (a directory name with a dot cannot be found
by the default path finder)."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "sketchlib.rust_amd")
if "sketchlib.rust_amd" not in sys.modules:
    _spec = importlib.util.spec_from_file_location(
        "sketchlib.rust_amd", os.path.join(_pkg_dir, "__init__.py"),
        submodule_search_locations=[_pkg_dir])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["sketchlib.rust_amd"] = _mod
    _spec.loader.exec_module(_mod)
rust_amd = sys.modules["sketchlib.rust_amd"]
