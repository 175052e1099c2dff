"""MI355X-native neural-audio-codec forward path (encoder -> RVQ -> decoder).

Hand-written HIP kernels for gfx950 behind the flat C ABI of ``include/agx.h``;
the Python modules mirror the reference's ``networks/vae.py`` module surface so
they drop into its callers.  Importing the package does not touch the GPU and
does not load the shared library; the first op call does, and raises if
``lib/libagx.so`` has not been built (``python -m audio_generation_amd.build``).
"""
from ._lib import AgxError, LIB_PATH  # noqa: F401

__all__ = ["AgxError", "LIB_PATH", "vae", "quantizer", "ops"]
