"""MI355X-native post-process hot path of FptrP/vk-renderer (Hi-Z / SSR / GTAO / TAA).

The product is the HIP C-ABI library in csrc/ (include/vkr_postfx.h) and the C++ host
mirror of the reference's pass structs in host/.  This Python package is plumbing for
tests, bench.py and multi-GPU launch: ctypes bindings (abi), image layout (images),
camera conventions (camera) and a flat driver of the chain (chain).
The directory name contains a hyphen; import it through the `vk_renderer_amd` shim at
the repository root.
"""
from . import abi, camera, images  # noqa: F401
