"""Import alias: the package directory is ``rust-lbfgs_amd/`` (a hyphen is not importable by name).

``import rust_lbfgs_amd`` loads that directory as the package ``rust_lbfgs_amd``.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rust-lbfgs_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
