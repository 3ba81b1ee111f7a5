"""ctypes binding of the C oracle (oracle/dcmt_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.  PARITY UNPINNED (see
dcmt_oracle.h): OpenCV is absent and the reference ships no fixtures for this path.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

STAGE_INVERT, STAGE_DILATE_K, STAGE_CLOSE5, STAGE_FILL7, STAGE_EXTEND = 2, 3, 4, 5, 6
STAGE_FILL31, STAGE_FILLLOOP, STAGE_MEDIAN5, STAGE_BLUR, STAGE_FINAL = 7, 8, 9, 10, 11
BLUR_NONE, BLUR_GAUSSIAN = 0, 1


class Params(ctypes.Structure):
    _fields_ = [
        ("max_depth", ctypes.c_float),
        ("k0", ctypes.c_uint8 * 25),
        ("blur", ctypes.c_int),
        ("max_fill_iters", ctypes.c_int),
        ("stop_after", ctypes.c_int),
    ]


def build(native: bool = False) -> str:
    """Compile the oracle with gcc if the .so is missing or stale; returns its path."""
    target = "libdcmt_oracle_native.so" if native else "libdcmt_oracle.so"
    # DCMT_ORACLE_TARGET=libdcmt_oracle_asan.so: the sanitizer build (tests/test_oracle.py runs the golden set through it in a
    # child process that preloads the sanitizer runtimes)
    target = os.environ.get("DCMT_ORACLE_TARGET", target)
    so = os.path.join(_HERE, target)
    deps = [os.path.join(_HERE, f) for f in ("dcmt_oracle.c", "dcmt_oracle.h", "median_nets.h")]
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps)
    if stale:
        subprocess.run(["make", "-C", _HERE, target], check=True, capture_output=True)
    return so


_libs: dict = {}


def lib(native: bool = False) -> ctypes.CDLL:
    if native not in _libs:
        L = ctypes.CDLL(build(native))
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        i32p = ctypes.POINTER(ctypes.c_int32)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        pp = ctypes.POINTER(Params)
        L.dcmt_oracle_default_params.argtypes = [pp]
        L.dcmt_oracle_default_params.restype = None
        for n in ("dcmt_oracle_k0_as_compiled", "dcmt_oracle_k0_diamond"):
            getattr(L, n).argtypes = [u8p]
            getattr(L, n).restype = None
        L.dcmt_oracle_img_completion.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, pp, ip, ip]
        L.dcmt_oracle_img_completion.restype = ctypes.c_int
        for n in ("dcmt_oracle_interpolate_with_superpixels",
                  "dcmt_oracle_interpolate_with_superpixels_bruteforce"):
            getattr(L, n).argtypes = [fp, i32p, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, pp,
                                      ctypes.c_int, ip]
            getattr(L, n).restype = ctypes.c_int
        L.dcmt_oracle_img_completion_batch.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_int, pp, ctypes.c_int]
        L.dcmt_oracle_img_completion_batch.restype = ctypes.c_int
        L.dcmt_oracle_dilate_mask5.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, u8p]
        L.dcmt_oracle_dilate_mask5.restype = None
        for n in ("dcmt_oracle_dilate_rect", "dcmt_oracle_erode_rect",
                  "dcmt_oracle_dilate_rect_bruteforce", "dcmt_oracle_erode_rect_bruteforce"):
            getattr(L, n).argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
            getattr(L, n).restype = None
        for n in ("dcmt_oracle_median5", "dcmt_oracle_median5_simple", "dcmt_oracle_gaussian5"):
            getattr(L, n).argtypes = [fp, fp, ctypes.c_int, ctypes.c_int]
            getattr(L, n).restype = None
        L.dcmt_oracle_use_definitional_median.argtypes = [ctypes.c_int]
        L.dcmt_oracle_use_definitional_median.restype = None
        L.dcmt_oracle_extend_columns.argtypes = [fp, ctypes.c_int, ctypes.c_int]
        L.dcmt_oracle_extend_columns.restype = None
        L.dcmt_oracle_normalize_minmax.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float]
        L.dcmt_oracle_normalize_minmax.restype = None
        L.dcmt_oracle_project_points.argtypes = [fp, ctypes.c_int, fp, fp, fp, ctypes.c_int, ctypes.c_int]
        L.dcmt_oracle_project_points.restype = None
        L.dcmt_oracle_slic.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, i32p,
                                       ctypes.POINTER(ctypes.c_double), ctypes.c_int]
        L.dcmt_oracle_slic.restype = ctypes.c_int
        L.dcmt_oracle_stereo_refine.argtypes = [fp, u8p, u8p, fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_float, ctypes.c_int]
        L.dcmt_oracle_stereo_refine.restype = None
        L.dcmt_oracle_synth_frame.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_uint64]
        L.dcmt_oracle_synth_frame.restype = None
        _libs[native] = L
    return _libs[native]


class definitional_median:
    """Context manager: the chain entry points run dcmt_oracle_median5_simple (gather 25, select the 13th) instead of the
    comparator networks -- an oracle that shares no network with the HIP kernels."""

    def __init__(self, native: bool = False):
        self._native = native

    def __enter__(self):
        lib(self._native).dcmt_oracle_use_definitional_median(1)
        return self

    def __exit__(self, *exc):
        lib(self._native).dcmt_oracle_use_definitional_median(0)
        return False


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _c32(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2
    return a


def default_params(k0: str = "as_compiled", blur: str = "gaussian", stop_after: int = STAGE_FINAL,
                   max_fill_iters: int = 64) -> Params:
    p = Params()
    lib().dcmt_oracle_default_params(ctypes.byref(p))
    if k0 == "diamond":
        lib().dcmt_oracle_k0_diamond(p.k0)
    elif k0 != "as_compiled":
        arr = np.asarray(k0, dtype=np.uint8).reshape(25)
        for i in range(25):
            p.k0[i] = int(arr[i])
    p.blur = BLUR_GAUSSIAN if blur == "gaussian" else BLUR_NONE
    p.stop_after = stop_after
    p.max_fill_iters = max_fill_iters
    return p


def k0_as_compiled() -> np.ndarray:
    k = (ctypes.c_uint8 * 25)()
    lib().dcmt_oracle_k0_as_compiled(k)
    return np.frombuffer(bytes(k), dtype=np.uint8).reshape(5, 5).copy()


def k0_diamond() -> np.ndarray:
    k = (ctypes.c_uint8 * 25)()
    lib().dcmt_oracle_k0_diamond(k)
    return np.frombuffer(bytes(k), dtype=np.uint8).reshape(5, 5).copy()


def img_completion(sparse, params: Params | None = None, return_info: bool = False):
    """LO/img_completion.cpp:17-204 on one frame (2-D f32 array)."""
    src = _c32(sparse)
    p = params or default_params()
    dst = np.empty_like(src)
    it, holes = ctypes.c_int(0), ctypes.c_int(0)
    rc = lib().dcmt_oracle_img_completion(_fp(src), _fp(dst), src.shape[0], src.shape[1],
                                          ctypes.byref(p), ctypes.byref(it), ctypes.byref(holes))
    if return_info:
        return dst, {"rc": rc, "fill_iters": it.value, "holes_after_extend": holes.value}
    return dst


def interpolate_with_superpixels(sparse, labels, n_labels: int, params: Params | None = None,
                                 use_superpixel: int = 1, bruteforce: bool = False,
                                 return_info: bool = False):
    """LC/img_completion_lc.cpp:34-203; labels int32[rows][cols]."""
    src = _c32(sparse)
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    assert lab.shape == src.shape
    p = params or default_params()
    dst = np.empty_like(src)
    it = ctypes.c_int(0)
    fn = (lib().dcmt_oracle_interpolate_with_superpixels_bruteforce if bruteforce
          else lib().dcmt_oracle_interpolate_with_superpixels)
    rc = fn(_fp(src), lab.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), int(n_labels), _fp(dst),
            src.shape[0], src.shape[1], ctypes.byref(p), int(use_superpixel), ctypes.byref(it))
    if return_info:
        return dst, {"rc": rc, "fill_iters": it.value}
    return dst


def img_completion_batch(frames, params: Params | None = None, threads: int = 1, native: bool = False):
    src = np.ascontiguousarray(frames, dtype=np.float32)
    assert src.ndim == 3
    p = params or default_params()
    dst = np.empty_like(src)
    rc = lib(native).dcmt_oracle_img_completion_batch(_fp(src), _fp(dst), src.shape[1], src.shape[2],
                                                      src.shape[0], ctypes.byref(p), int(threads))
    return dst, rc


def _unary(name, a, *extra):
    src = _c32(a)
    dst = np.empty_like(src)
    getattr(lib(), name)(_fp(src), _fp(dst), src.shape[0], src.shape[1], *extra)
    return dst


def dilate_mask5(a, k):
    k = np.ascontiguousarray(k, dtype=np.uint8).reshape(25)
    return _unary("dcmt_oracle_dilate_mask5", a, k.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))


def dilate_rect(a, ksize, bruteforce=False):
    return _unary("dcmt_oracle_dilate_rect" + ("_bruteforce" if bruteforce else ""), a, int(ksize))


def erode_rect(a, ksize, bruteforce=False):
    return _unary("dcmt_oracle_erode_rect" + ("_bruteforce" if bruteforce else ""), a, int(ksize))


def median5(a, simple=False):
    return _unary("dcmt_oracle_median5_simple" if simple else "dcmt_oracle_median5", a)


def gaussian5(a):
    return _unary("dcmt_oracle_gaussian5", a)


def extend_columns(a):
    x = _c32(a).copy()
    lib().dcmt_oracle_extend_columns(_fp(x), x.shape[0], x.shape[1])
    return x


def normalize_minmax(a, lo: float, hi: float) -> np.ndarray:
    """cv::normalize(a, dst, lo, hi, NORM_MINMAX), f32 (SL/main_sl.cpp:370, :523)."""
    return _unary("dcmt_oracle_normalize_minmax", a, ctypes.c_float(lo), ctypes.c_float(hi))


def project_points(points, T, P, rows: int, cols: int) -> np.ndarray:
    """SL/main_sl.cpp:478-520: points [n][4] f32, T 4x4, P 3x4 (row-major) -> sparse depth image."""
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4)
    t = np.ascontiguousarray(T, dtype=np.float32).reshape(16)
    p = np.ascontiguousarray(P, dtype=np.float32).reshape(12)
    dst = np.empty((rows, cols), dtype=np.float32)
    lib().dcmt_oracle_project_points(_fp(pts), pts.shape[0], _fp(t), _fp(p), _fp(dst), rows, cols)
    return dst


def slic(lab_image, step: int, nc: int, return_centers: bool = False):
    """Slic::generate_superpixels (LC/slic.cpp:101-182) on an 8-bit 3-channel image [rows][cols][3]:
    returns (labels int32 [rows][cols], n_centers[, centers float64 [n][5]])."""
    img = np.ascontiguousarray(lab_image, dtype=np.uint8)
    assert img.ndim == 3 and img.shape[2] == 3
    rows, cols = img.shape[:2]
    labels = np.empty((rows, cols), dtype=np.int32)
    cap = max(1, (cols // max(step, 1) + 1) * (rows // max(step, 1) + 1))
    centers = np.empty((cap, 5), dtype=np.float64)
    n = lib().dcmt_oracle_slic(img.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), rows, cols, int(step), int(nc),
                               labels.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                               centers.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), cap)
    if n < 0:
        raise ValueError("dcmt_oracle_slic: step < 6 or nc < 1")
    return (labels, n, centers[:n].copy()) if return_centers else (labels, n)


def stereo_refine(depth, left, right, baseline=0.54, focal=9.597910e+02, damp=500.0, max_depth=100.0, iterations=4):
    """SL/main_sl.cpp:715-885 as driven from :1165-1246: dense depth + grey stereo pair -> refined depth."""
    d = _c32(depth)
    l = np.ascontiguousarray(left, dtype=np.uint8)
    r = np.ascontiguousarray(right, dtype=np.uint8)
    assert l.shape == d.shape == r.shape
    dst = np.empty_like(d)
    u8 = ctypes.POINTER(ctypes.c_uint8)
    lib().dcmt_oracle_stereo_refine(_fp(d), l.ctypes.data_as(u8), r.ctypes.data_as(u8), _fp(dst), d.shape[0], d.shape[1],
                                    baseline, focal, damp, max_depth, int(iterations))
    return dst


def synth_frame(rows: int, cols: int, seed: int) -> np.ndarray:
    dst = np.empty((rows, cols), dtype=np.float32)
    lib().dcmt_oracle_synth_frame(_fp(dst), rows, cols, ctypes.c_uint64(seed))
    return dst
