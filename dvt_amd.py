"""Import alias: ``import dvt_amd`` loads the package that lives in the
(non-identifier) directory ``data-efficient-video-transformers_amd/``."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data-efficient-video-transformers_amd")
_spec = importlib.util.spec_from_file_location(
    "dvt_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dvt_amd"] = _mod
_spec.loader.exec_module(_mod)
