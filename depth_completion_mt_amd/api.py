"""Python host side of the C ABI: mirrors the reference's two entry points.

    img_completion(sparse, extr=False, blur_type="gaussian") -> dense
        reference: src/DC_lidar_only/img_completion.cpp:17-20
    interpolate_with_superpixels(labels, sparse, blur_type="gaussian", use_superpixel=1) -> dense
        reference: src/DC_lidar_camera/img_completion_lc.cpp:34-38

numpy arrays go through the host entry point (dcmt_complete_f32: H2D, kernels, D2H, exact
hole-closure loop); torch CUDA tensors go through the device entry point on torch's current
stream (dcmt_complete_f32_dev: asynchronous).  PyTorch is only the owner of device memory
here.  All arithmetic happens in csrc/ (HIP); there is no CPU path.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib as L


class DcmtError(RuntimeError):
    def __init__(self, status: int, what: str = ""):
        super().__init__(f"{what}: {L.strerror(status)} (status {status})")
        self.status = status


def make_params(k0="as_compiled", blur_type: str = "gaussian", stop_after: int = L.STAGE_FINAL,
                max_fill_iters: int = 64, spec_fill_iters: int = 1, verbose: bool = False,
                max_depth: float = 100.0, valid_thresh: float = 0.1, force_staged: bool = False, force_fused: bool = False,
                normalize=None) -> L.Params:
    """normalize=(lo, hi): min-max normalise every frame first, as cv::normalize(src, dst, lo, hi, NORM_MINMAX)
    in front of the path does in the stereo-lidar executables (SL/main_sl.cpp:370, :523)."""
    p = L.Params()
    L.lib().dcmt_default_params(ctypes.byref(p))
    if isinstance(k0, str):
        if k0 == "diamond":
            L.lib().dcmt_k0_diamond(p.k0)
        elif k0 != "as_compiled":
            raise ValueError("k0 must be 'as_compiled', 'diamond' or a 5x5 array")
    else:
        arr = np.asarray(k0, dtype=np.uint8).reshape(25)
        for i in range(25):
            p.k0[i] = int(arr[i])
    # the reference compares strings: "bilateral" / "gaussian" / anything else = no blur
    p.blur = {"gaussian": L.BLUR_GAUSSIAN, "bilateral": L.BLUR_BILATERAL}.get(blur_type, L.BLUR_NONE)
    p.stop_after = int(stop_after)
    p.max_fill_iters = int(max_fill_iters)
    p.spec_fill_iters = int(spec_fill_iters)
    p.verbose = int(verbose)
    p.max_depth = float(max_depth)
    p.valid_thresh = float(valid_thresh)
    p.flags = (L.FLAG_FORCE_STAGED if force_staged else 0) | (L.FLAG_FORCE_FUSED if force_fused else 0)
    if normalize is not None:
        p.flags |= L.FLAG_NORMALIZE
        p.norm_lo, p.norm_hi = float(normalize[0]), float(normalize[1])
    return p


class Context:
    """One dcmt_ctx: bound to one GPU, owns the device scratch.  Not thread-safe."""

    def __init__(self, device: int = 0, max_rows: int = 352, max_cols: int = 1216, max_batch: int = 1):
        self._h = ctypes.c_void_p()
        self.device, self.max_rows, self.max_cols, self.max_batch = device, max_rows, max_cols, max_batch
        st = L.lib().dcmt_create(device, max_rows, max_cols, max_batch, ctypes.byref(self._h))
        if st != L.OK:
            raise DcmtError(st, "dcmt_create")

    def close(self):
        if self._h:
            L.lib().dcmt_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- host arrays ------------------------------------------------------------
    def complete(self, sparse: np.ndarray, params: L.Params | None = None, labels: np.ndarray | None = None,
                 n_labels: int = 0, use_superpixel: int = 1, allow_not_converged: bool = False) -> np.ndarray:
        """sparse: f32 [rows][cols] or [batch][rows][cols] (any row stride); returns a new array."""
        p = params or make_params()
        src = np.asarray(sparse, dtype=np.float32)
        single = src.ndim == 2
        if single:
            src = src[None]
        if src.ndim != 3 or src.strides[2] != 4:
            src = np.ascontiguousarray(src)
        b, r, c = src.shape
        dst = np.empty((b, r, c), dtype=np.float32)
        if labels is None:
            st = L.lib().dcmt_complete_f32(self._h, src.ctypes.data, src.strides[1], src.strides[0],
                                           dst.ctypes.data, dst.strides[1], dst.strides[0], r, c, b, ctypes.byref(p))
        else:
            lab = np.ascontiguousarray(np.asarray(labels, dtype=np.int32).reshape(b, r, c))
            st = L.lib().dcmt_complete_labeled_f32(self._h, src.ctypes.data, src.strides[1], src.strides[0],
                                                   lab.ctypes.data, lab.strides[1], lab.strides[0], int(n_labels),
                                                   dst.ctypes.data, dst.strides[1], dst.strides[0], r, c, b,
                                                   ctypes.byref(p), int(use_superpixel))
        if st != L.OK and not (allow_not_converged and st == L.E_NOT_CONVERGED):
            raise DcmtError(st, "dcmt_complete_f32")
        self.last_status = st
        return dst[0] if single else dst

    # ---- device tensors (torch only as the owner of device memory) ----------------
    def complete_dev(self, d_src, d_dst=None, params: L.Params | None = None, d_labels=None, n_labels: int = 0,
                     use_superpixel: int = 1, stream: int | None = None):
        """d_src/d_dst: contiguous f32 CUDA tensors [batch][rows][cols] (or [rows][cols]) on this
        context's GPU.  Enqueues on `stream` (a hipStream_t as int; default torch's current stream)
        and returns immediately."""
        import torch
        p = params or make_params()
        assert d_src.is_cuda and d_src.dtype == torch.float32 and d_src.is_contiguous()
        if d_dst is None:
            d_dst = torch.full_like(d_src, float("nan"))     # never mistake stale memory for output
        assert d_dst.is_cuda and d_dst.dtype == torch.float32 and d_dst.is_contiguous() and d_dst.shape == d_src.shape
        shp = d_src.shape if d_src.dim() == 3 else (1,) + tuple(d_src.shape)
        b, r, c = shp
        if stream is None:
            stream = torch.cuda.current_stream(d_src.device).cuda_stream
        if d_labels is None:
            st = L.lib().dcmt_complete_f32_dev(self._h, d_src.data_ptr(), d_dst.data_ptr(), r, c, b, ctypes.byref(p),
                                               ctypes.c_void_p(stream))
        else:
            assert d_labels.is_cuda and d_labels.dtype == torch.int32 and d_labels.is_contiguous()
            st = L.lib().dcmt_complete_labeled_f32_dev(self._h, d_src.data_ptr(), d_labels.data_ptr(), int(n_labels),
                                                       d_dst.data_ptr(), r, c, b, ctypes.byref(p), int(use_superpixel),
                                                       ctypes.c_void_p(stream))
        if st != L.OK:
            raise DcmtError(st, "dcmt_complete_f32_dev")
        return d_dst

    def complete_u16_dev(self, d_src16, scale: float = 1.0 / 256.0, d_dst=None, params: L.Params | None = None,
                         stream: int | None = None):
        """KITTI uint16 depth payload in (torch.uint16 or int16-viewed CUDA tensor [batch][rows][cols]), metres out:
        the reference's imread + convertTo(CV_32F, 1/256) (src/DC_lidar_only/main.cpp:75-82) fused into the first kernel."""
        import torch
        p = params or make_params()
        assert d_src16.is_cuda and d_src16.element_size() == 2 and d_src16.is_contiguous()
        shp = d_src16.shape if d_src16.dim() == 3 else (1,) + tuple(d_src16.shape)
        b, r, c = shp
        if d_dst is None:
            d_dst = torch.full(tuple(d_src16.shape), float("nan"), dtype=torch.float32, device=d_src16.device)
        if stream is None:
            stream = torch.cuda.current_stream(d_src16.device).cuda_stream
        st = L.lib().dcmt_complete_u16_dev(self._h, d_src16.data_ptr(), ctypes.c_float(scale), d_dst.data_ptr(), r, c, b,
                                           ctypes.byref(p), ctypes.c_void_p(stream))
        if st != L.OK:
            raise DcmtError(st, "dcmt_complete_u16_dev")
        return d_dst

    def project_points_dev(self, d_points, d_offsets, T, P, rows: int, cols: int, d_sparse=None, stream: int | None = None):
        """N2 (SL/main_sl.cpp:478-520): velodyne points -> sparse depth images on the device.  d_points: f32 CUDA tensor
        [n][4] (x, y, z, reflectance); d_offsets: int32 CUDA tensor [batch + 1], frame f owns points
        [offsets[f], offsets[f+1]); T 4x4, P 3x4 row-major.  Returns [batch][rows][cols] f32, 0 = no point."""
        import torch
        assert d_points.is_cuda and d_points.dtype == torch.float32 and d_points.is_contiguous() and d_points.shape[-1] == 4
        assert d_offsets.is_cuda and d_offsets.dtype == torch.int32 and d_offsets.is_contiguous()
        batch = d_offsets.numel() - 1
        n = d_points.numel() // 4
        if d_sparse is None:
            d_sparse = torch.full((batch, rows, cols), float("nan"), dtype=torch.float32, device=d_points.device)
        assert d_sparse.is_cuda and d_sparse.dtype == torch.float32 and d_sparse.is_contiguous() and tuple(d_sparse.shape) == (batch, rows, cols)
        t = np.ascontiguousarray(T, dtype=np.float32).reshape(16)
        p = np.ascontiguousarray(P, dtype=np.float32).reshape(12)
        if stream is None:
            stream = torch.cuda.current_stream(d_points.device).cuda_stream
        st = L.lib().dcmt_project_points_dev(self._h, d_points.data_ptr(), d_offsets.data_ptr(), n, batch, t.ctypes.data, p.ctypes.data,
                                             d_sparse.data_ptr(), rows, cols, ctypes.c_void_p(stream))
        if st != L.OK:
            raise DcmtError(st, "dcmt_project_points_dev")
        return d_sparse

    def slic_labels_dev(self, d_lab, step: int, nc: int, d_labels=None, return_centers: bool = False, stream: int | None = None):
        """N3, Slic::generate_superpixels (LC/slic.cpp:101-182) on the device.  d_lab: uint8 CUDA tensor [batch][rows][cols][3]
        (or [rows][cols][3]).  Returns (labels int32 [batch][rows][cols], n_centers[, centers float64 [batch][n][5]])."""
        import torch
        assert d_lab.is_cuda and d_lab.dtype == torch.uint8 and d_lab.is_contiguous() and d_lab.shape[-1] == 3
        shp = d_lab.shape if d_lab.dim() == 4 else (1,) + tuple(d_lab.shape)
        b, r, c = shp[0], shp[1], shp[2]
        n = L.lib().dcmt_slic_num_centers(r, c, int(step))
        if d_labels is None:
            d_labels = torch.full((b, r, c), -7, dtype=torch.int32, device=d_lab.device)
        assert d_labels.is_cuda and d_labels.dtype == torch.int32 and d_labels.is_contiguous() and tuple(d_labels.shape) == (b, r, c)
        d_cent = torch.empty((b, max(n, 1), 5), dtype=torch.float64, device=d_lab.device) if return_centers else None
        if stream is None:
            stream = torch.cuda.current_stream(d_lab.device).cuda_stream
        st = L.lib().dcmt_slic_labels_dev(self._h, d_lab.data_ptr(), r, c, b, int(step), int(nc), d_labels.data_ptr(),
                                          d_cent.data_ptr() if return_centers else None, ctypes.c_void_p(stream))
        if st != L.OK:
            raise DcmtError(st, "dcmt_slic_labels_dev")
        return (d_labels, n, d_cent[:, :n]) if return_centers else (d_labels, n)

    def stereo_refine_dev(self, d_depth, d_left, d_right, d_out=None, iterations: int | None = None, stream: int | None = None, **kw):
        """N4 (SL/main_sl.cpp:715-885): dense depth + grey stereo pair (uint8 CUDA tensors) -> refined depth.
        kw: baseline, focal, damp, max_depth override the reference's constants."""
        import torch
        assert d_depth.is_cuda and d_depth.dtype == torch.float32 and d_depth.is_contiguous()
        for t in (d_left, d_right):
            assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and t.shape == d_depth.shape
        shp = d_depth.shape if d_depth.dim() == 3 else (1,) + tuple(d_depth.shape)
        b, r, c = shp
        if d_out is None:
            d_out = torch.full_like(d_depth, float("nan"))
        sp = L.StereoParams()
        L.lib().dcmt_default_stereo_params(ctypes.byref(sp))
        for k, v in kw.items():
            setattr(sp, k, float(v))
        if iterations is not None:
            sp.iterations = int(iterations)
        if stream is None:
            stream = torch.cuda.current_stream(d_depth.device).cuda_stream
        st = L.lib().dcmt_stereo_refine_dev(self._h, d_depth.data_ptr(), d_left.data_ptr(), d_right.data_ptr(), d_out.data_ptr(),
                                            r, c, b, ctypes.byref(sp), ctypes.c_void_p(stream))
        if st != L.OK:
            raise DcmtError(st, "dcmt_stereo_refine_dev")
        return d_out

    def last_fill_iters(self, n: int):
        out = (ctypes.c_int * n)()
        st = L.lib().dcmt_last_fill_iters(self._h, out, n)
        if st not in (L.OK, L.E_NOT_CONVERGED):
            raise DcmtError(st, "dcmt_last_fill_iters")
        return list(out), st

    def set_kernel_timing(self, on: bool = True):
        """Measurement aid: HIP events around the kernel groups of the streaming path of every following *_dev call."""
        st = L.lib().dcmt_set_kernel_timing(self._h, int(bool(on)))
        if st != L.OK:
            raise DcmtError(st, "dcmt_set_kernel_timing")

    def last_kernel_times(self) -> dict:
        """Milliseconds of the last *_dev call's kernel groups (synchronises with its stream)."""
        ms = (ctypes.c_float * 4)()
        st = L.lib().dcmt_last_kernel_times(self._h, ms)
        if st != L.OK:
            raise DcmtError(st, "dcmt_last_kernel_times")
        return {"front": ms[0], "k_pre": ms[1], "k_fp": ms[2], "k_fp_s": ms[2], "behind": ms[3]}

    def last_path(self) -> str:
        """The kernels the last cascade call dispatched (dcmt_last_path)."""
        return L.lib().dcmt_last_path(self._h).decode()

    def last_holes_after_extend(self, n: int):
        out = (ctypes.c_int * n)()
        st = L.lib().dcmt_last_holes_after_extend(self._h, out, n)
        if st != L.OK:
            raise DcmtError(st, "dcmt_last_holes_after_extend")
        return list(out)


_default_ctx: dict = {}


def _ctx_for(rows: int, cols: int, batch: int, device: int = 0) -> Context:
    key = device
    c = _default_ctx.get(key)
    if c is None or c.max_rows < rows or c.max_cols < cols or c.max_batch < batch:
        if c is not None:
            c.close()
        c = Context(device, max(rows, 352), max(cols, 1216), max(batch, 1))
        _default_ctx[key] = c
    return c


def img_completion(sparse_r_img: np.ndarray, extr: bool = False, blur_type: str = "gaussian", **kw) -> np.ndarray:
    """Drop-in for the reference's img_completion (LO/img_completion.cpp:17): returns dense_r_img.
    `extr` is accepted and ignored, as in the reference (:19, never read)."""
    a = np.asarray(sparse_r_img, dtype=np.float32)
    b = 1 if a.ndim == 2 else a.shape[0]
    return _ctx_for(a.shape[-2], a.shape[-1], b).complete(a, make_params(blur_type=blur_type, **kw))


def interpolate_with_superpixels(labels: np.ndarray, n_labels: int, sparse_r_img: np.ndarray,
                                 blur_type: str = "gaussian", use_superpixel: int = 1, **kw) -> np.ndarray:
    """Drop-in for LC/img_completion_lc.cpp:34.  `labels` is int32 [rows][cols] (the reference's
    Slic::clusters is [col][row]: pass clusters.T), n_labels = slic.centers.size().  blur_type is
    accepted and ignored, as in the reference (:37, :183 always blurs)."""
    a = np.asarray(sparse_r_img, dtype=np.float32)
    b = 1 if a.ndim == 2 else a.shape[0]
    return _ctx_for(a.shape[-2], a.shape[-1], b).complete(a, make_params(blur_type="gaussian", **kw), labels=labels,
                                                          n_labels=n_labels, use_superpixel=use_superpixel)
