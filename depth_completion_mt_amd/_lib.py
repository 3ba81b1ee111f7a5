"""ctypes loader for libdcmt_hip.so (the C ABI of include/dcmt.h).

There is no fallback: if the shared library is missing it is built with hipcc, and if that
fails the import raises.  Nothing under oracle/ is ever imported from here.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "libdcmt_hip.so")

OK, E_INVALID, E_UNSUPPORTED, E_NOMEM, E_HIP, E_NOT_CONVERGED, E_NO_DEVICE = 0, -1, -2, -3, -4, -5, -6
BLUR_NONE, BLUR_GAUSSIAN, BLUR_BILATERAL = 0, 1, 2
STAGE_NORMALIZE = 1
STAGE_INVERT, STAGE_DILATE_K, STAGE_CLOSE5, STAGE_FILL7, STAGE_EXTEND = 2, 3, 4, 5, 6
FLAG_FORCE_STAGED = 1
FLAG_FORCE_FUSED = 2
FLAG_NORMALIZE = 4
STAGE_FILL31, STAGE_FILLLOOP, STAGE_MEDIAN5, STAGE_BLUR, STAGE_FINAL = 7, 8, 9, 10, 11

# every symbol include/dcmt.h declares (tests check the library exports exactly these)
EXPORTS = (
    "dcmt_device_count", "dcmt_create", "dcmt_destroy", "dcmt_default_params", "dcmt_k0_as_compiled",
    "dcmt_k0_diamond", "dcmt_complete_f32", "dcmt_complete_f32_dev", "dcmt_complete_labeled_f32",
    "dcmt_complete_labeled_f32_dev", "dcmt_complete_u16_dev", "dcmt_last_fill_iters", "dcmt_last_holes_after_extend",
    "dcmt_strerror", "dcmt_last_hip_error", "dcmt_version", "dcmt_project_points_dev", "dcmt_set_kernel_timing", "dcmt_last_kernel_times",
    "dcmt_slic_num_centers", "dcmt_slic_labels_dev", "dcmt_default_stereo_params", "dcmt_stereo_refine_dev",
    "dcmt_project_points", "dcmt_slic_labels", "dcmt_stereo_refine", "dcmt_last_path",
)


class Params(ctypes.Structure):
    """Mirror of dcmt_params (include/dcmt.h)."""
    _fields_ = [
        ("max_depth", ctypes.c_float),
        ("valid_thresh", ctypes.c_float),
        ("k0", ctypes.c_uint8 * 25),
        ("_pad", ctypes.c_uint8 * 3),
        ("blur", ctypes.c_int32),
        ("max_fill_iters", ctypes.c_int32),
        ("spec_fill_iters", ctypes.c_int32),
        ("stop_after", ctypes.c_int32),
        ("verbose", ctypes.c_int32),
        ("flags", ctypes.c_int32),
        ("norm_lo", ctypes.c_float),
        ("norm_hi", ctypes.c_float),
    ]


class StereoParams(ctypes.Structure):
    """Mirror of dcmt_stereo_params (include/dcmt.h)."""
    _fields_ = [("baseline", ctypes.c_float), ("focal", ctypes.c_float), ("damp", ctypes.c_float),
                ("max_depth", ctypes.c_float), ("iterations", ctypes.c_int32)]


def build(force: bool = False) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))]
    root = os.path.dirname(os.path.dirname(_CSRC))
    srcs += [os.path.join(root, "include", "dcmt.h")]
    stale = force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs)
    if stale:
        r = subprocess.run(["make", "-C", _CSRC, "-B", "libdcmt_hip.so"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building libdcmt_hip.so failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


_lib = None


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7, found through its RPATH); libdcmt_hip.so needs libamdhip64.so.7 too.  If
    the system copy were loaded for us and the bundled one for torch, two HIP/HSA runtimes
    would fight over the device ("No HIP GPUs are available").  Loading torch's copy first
    makes the dynamic linker resolve our NEEDED entry to it by SONAME; a later `import torch`
    finds the same file.  Without torch installed the system runtime is used."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _share_hip_runtime_with_torch()
        L = ctypes.CDLL(LIB_PATH)
        vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        pp = ctypes.POINTER(Params)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        ip = ctypes.POINTER(ctypes.c_int)
        L.dcmt_device_count.argtypes = []
        L.dcmt_create.argtypes = [i, i, i, i, ctypes.POINTER(vp)]
        L.dcmt_destroy.argtypes = [vp]
        L.dcmt_destroy.restype = None
        L.dcmt_default_params.argtypes = [pp]
        L.dcmt_default_params.restype = None
        L.dcmt_k0_as_compiled.argtypes = [u8p]
        L.dcmt_k0_as_compiled.restype = None
        L.dcmt_k0_diamond.argtypes = [u8p]
        L.dcmt_k0_diamond.restype = None
        L.dcmt_complete_f32.argtypes = [vp, vp, sz, sz, vp, sz, sz, i, i, i, pp]
        L.dcmt_complete_f32_dev.argtypes = [vp, vp, vp, i, i, i, pp, vp]
        L.dcmt_complete_labeled_f32.argtypes = [vp, vp, sz, sz, vp, sz, sz, i, vp, sz, sz, i, i, i, pp, i]
        L.dcmt_complete_labeled_f32_dev.argtypes = [vp, vp, vp, i, vp, i, i, i, pp, i, vp]
        L.dcmt_complete_u16_dev.argtypes = [vp, vp, ctypes.c_float, vp, i, i, i, pp, vp]
        L.dcmt_project_points_dev.argtypes = [vp, vp, vp, i, i, vp, vp, vp, i, i, vp]
        L.dcmt_default_stereo_params.argtypes = [vp]
        L.dcmt_default_stereo_params.restype = None
        L.dcmt_stereo_refine_dev.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp, vp]
        L.dcmt_project_points.argtypes = [vp, vp, i, vp, vp, vp, sz, i, i]
        L.dcmt_slic_labels.argtypes = [vp, vp, sz, i, i, i, i, vp, vp]
        L.dcmt_stereo_refine.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz, i, i, vp]
        L.dcmt_slic_num_centers.argtypes = [i, i, i]
        L.dcmt_slic_labels_dev.argtypes = [vp, vp, i, i, i, i, i, vp, vp, vp]
        L.dcmt_last_fill_iters.argtypes = [vp, ip, i]
        L.dcmt_last_holes_after_extend.argtypes = [vp, ip, i]
        L.dcmt_strerror.argtypes = [i]
        L.dcmt_strerror.restype = ctypes.c_char_p
        L.dcmt_last_hip_error.argtypes = [vp]
        L.dcmt_last_path.argtypes = [vp]
        L.dcmt_last_path.restype = ctypes.c_char_p
        L.dcmt_set_kernel_timing.argtypes = [vp, i]
        L.dcmt_last_kernel_times.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        L.dcmt_version.argtypes = []
        _lib = L
    return _lib


def strerror(status: int) -> str:
    return lib().dcmt_strerror(status).decode()
