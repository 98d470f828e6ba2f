"""Import shim: exposes the package in `vk-renderer_amd/` (not a valid identifier) as `vk_renderer_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vk-renderer_amd")
_spec = importlib.util.spec_from_file_location(
    "vk_renderer_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vk_renderer_amd"] = _mod
_spec.loader.exec_module(_mod)
