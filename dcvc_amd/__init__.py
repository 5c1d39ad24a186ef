"""dcvc_amd - MI355X-native DCVC-UF inference codec (hot path only).

  dcvc_amd/csrc      hand-written HIP kernels (gfx950), host codec, rANS coder, C ABI
  dcvc_amd/plugin    Python modules with the reference's plugin names
                     (``inference_extensions_cuda``, ``MLCodec_extensions_cpp``)
  dcvc_amd/_lib.py   ctypes loader for libdcvc_amd.so (no fallback)
"""
import os
import sys

__version__ = "0.1.0"

PLUGIN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plugin")


def install_plugin():
    """Make ``import inference_extensions_cuda`` / ``import MLCodec_extensions_cpp`` (the names
    the reference imports lazily, image_model.py:197, entropy_models.py:34) resolve to the
    MI355X implementation."""
    if PLUGIN_DIR not in sys.path:
        sys.path.insert(0, PLUGIN_DIR)
