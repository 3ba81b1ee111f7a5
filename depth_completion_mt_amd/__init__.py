"""MI355X-native (gfx950) morphological depth completion: the `img_completion` cascade of
PatrizioPerugini/depth_completion_MT as hand-written HIP kernels behind a C ABI.

    from depth_completion_mt_amd import img_completion, interpolate_with_superpixels, Context
"""
from .api import Context, DcmtError, img_completion, interpolate_with_superpixels, make_params  # noqa: F401
from . import synth  # noqa: F401

__all__ = ["Context", "DcmtError", "img_completion", "interpolate_with_superpixels", "make_params", "synth"]
