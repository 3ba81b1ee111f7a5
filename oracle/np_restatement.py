"""Second, independently written restatement of the reference cascade, in numpy.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED (OpenCV absent, the reference
ships no fixtures).  This file exists to (a) cross-check oracle/dcmt_oracle.c -- two
restatements written in different styles (whole-array shifted slices here, scalar loops
there) agreeing bit for bit is the best substitute for executing OpenCV that this image
allows -- and (b) generate the golden vectors in tests/golden/ (tests/golden/make_golden.py).

Follows /root/reference/src/DC_lidar_only/img_completion.cpp:17-204 and
src/DC_lidar_camera/img_completion_lc.cpp:34-203.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
FLT_MAX = np.finfo(np.float32).max

# LO/img_completion.cpp:71-77 -- int d[5][5] viewed as 25 bytes (little endian host)
K0_AS_COMPILED = np.frombuffer(
    np.array([0, 0, 1, 0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1, 0, 0, 0, 1, 0, 0],
             dtype="<i4").tobytes()[:25], dtype=np.uint8).reshape(5, 5).copy()
K0_DIAMOND = np.array([0, 0, 1, 0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1, 0, 0, 0, 1, 0, 0],
                      dtype=np.uint8).reshape(5, 5)


def _valid(x):   # `depth > 0.1`: float promoted to double against the double literal
    return x.astype(np.float64) > 0.1


def _hole(x):    # `depth < 0.1`
    return x.astype(np.float64) < 0.1


def invert(x, max_depth=F32(100.0)):
    """LO :55-67 / :191-202."""
    return np.where(_valid(x), F32(max_depth) - x, x).astype(F32)


def _shifted(x, dr, dc, fill):
    """y[r,c] = x[r+dr, c+dc], `fill` outside the image."""
    R, C = x.shape
    y = np.full_like(x, fill)
    rs0, rs1 = max(0, dr), min(R, R + dr)
    cs0, cs1 = max(0, dc), min(C, C + dc)
    if rs0 < rs1 and cs0 < cs1:
        y[rs0 - dr:rs1 - dr, cs0 - dc:cs1 - dc] = x[rs0:rs1, cs0:cs1]
    return y


def dilate(x, kernel):
    """cv::dilate, anchor centre, element not reflected, border -FLT_MAX (LO :80,:85,:90,:134)."""
    kh, kw = kernel.shape
    out = np.full_like(x, -FLT_MAX)
    for kr in range(kh):
        for kc in range(kw):
            if kernel[kr, kc]:
                out = np.maximum(out, _shifted(x, kr - kh // 2, kc - kw // 2, -FLT_MAX))
    return out


def erode(x, kernel):
    """cv::erode, border +FLT_MAX (second half of MORPH_CLOSE, LO :85)."""
    kh, kw = kernel.shape
    out = np.full_like(x, FLT_MAX)
    for kr in range(kh):
        for kc in range(kw):
            if kernel[kr, kc]:
                out = np.minimum(out, _shifted(x, kr - kh // 2, kc - kw // 2, FLT_MAX))
    return out


def _ones(k):
    return np.ones((k, k), dtype=np.uint8)


def dilate_rect_fast(x, k):
    """Same as dilate(x, ones(k,k)) via two 1-D passes (exactness of max)."""
    return dilate(dilate(x, np.ones((1, k), np.uint8)), np.ones((k, 1), np.uint8))


def fill(x, k):
    """s = dilate(clone(x), ones(k,k)); x = x<0.1 ? s : x (LO :88-100, :131-144)."""
    s = dilate_rect_fast(x, k)
    h = _hole(x)
    return np.where(h, s, x).astype(F32), int(h.sum())


def extend_columns(x):
    """LO :103-129."""
    x = x.copy()
    R, C = x.shape
    v = _valid(x)
    for j in range(C):
        idx = np.flatnonzero(v[:, j])
        if idx.size == 0:
            # first loop writes -1 from row 0, second overwrites rows R-1..0 with 100
            x[:, j] = F32(100.0)
            continue
        top, bot = idx[0], idx[-1]
        bv, tv = x[bot, j], x[top, j]
        x[bot:, j] = bv
        x[:top + 1, j] = tv
    return x


def normalize_minmax(x, lo, hi):
    """cv::normalize(x, dst, lo, hi, NORM_MINMAX) for CV_32F (SL/main_sl.cpp:370, :523): see dcmt_oracle.c."""
    x = np.asarray(x, dtype=F32)
    smin, smax = np.float64(x.min()), np.float64(x.max())
    dmin, dmax = np.float64(min(lo, hi)), np.float64(max(lo, hi))
    scale = (dmax - dmin) * (1.0 / (smax - smin) if smax - smin > np.finfo(np.float64).eps else 0.0)
    scale = np.float64(F32(scale))
    shift = np.float64(F32(dmin)) - np.float64(F32(smin * scale))
    return (x * F32(scale)).astype(F32) + F32(shift)


def project_points(points, T, P, rows, cols):
    """N2, SL/main_sl.cpp:478-520: whole-array f32 arithmetic (every numpy op rounds to f32 once, sums left to right),
    last point in file order wins a pixel.  See dcmt_oracle.c for the line-by-line citations."""
    pts = np.asarray(points, dtype=F32).reshape(-1, 4)
    T = np.asarray(T, dtype=F32).reshape(4, 4)
    P = np.asarray(P, dtype=F32).reshape(3, 4)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    dot = lambda m, a, b, c: ((m[0] * a + m[1] * b) + m[2] * c) + m[3]
    tx, ty, tz = dot(T[0], x, y, z), dot(T[1], x, y, z), dot(T[2], x, y, z)
    px, py, pz = dot(P[0], tx, ty, tz), dot(P[1], tx, ty, tz), dot(P[2], tx, ty, tz)
    with np.errstate(divide="ignore", invalid="ignore"):
        uf, vf = px / pz, py / pz
    ok = (tz > 0) & (uf >= 0) & (uf < F32(cols)) & (vf >= 0) & (vf < F32(rows))
    idx = np.nonzero(ok)[0]
    out = np.zeros((rows, cols), F32)
    winner = np.full(rows * cols, -1, np.int64)
    flat = vf[idx].astype(np.int64) * cols + uf[idx].astype(np.int64)
    np.maximum.at(winner, flat, idx)
    hit = winner >= 0
    out.reshape(-1)[hit] = pz[winner[hit]]
    return out


def slic(lab, step, nc, iterations=10):
    """N3, Slic::generate_superpixels (LC/slic.cpp:101-182): one numpy window operation per centre instead of the
    reference's scalar loops; float64 throughout.  Returns (labels int32 [rows][cols], n_centers, centers [n][5])."""
    img = np.asarray(lab, dtype=np.uint8)
    rows, cols = img.shape[:2]
    f = img.astype(np.float64)
    ns = step
    cents = []
    for i in range(step, cols - step // 2, step):          # init_data :33-56: x outer, y inner
        for j in range(step, rows - step // 2, step):
            best, loc = np.float64(np.finfo(np.float32).max), (i, j)
            for ii in range(i - 1, i + 2):                 # find_local_minimum :71-98
                for jj in range(j - 1, j + 2):
                    gsum = abs(f[jj + 1, ii, 0] - f[jj, ii, 0]) + abs(f[jj, ii + 1, 0] - f[jj, ii, 0])
                    if gsum < best:
                        best, loc = gsum, (ii, jj)
            cents.append([f[loc[1], loc[0], 0], f[loc[1], loc[0], 1], f[loc[1], loc[0], 2], loc[0], loc[1]])
    C = np.array(cents, dtype=np.float64).reshape(-1, 5)
    n = len(C)
    labels = np.full((rows, cols), -1, np.int32)
    for _ in range(iterations):
        dist = np.full((rows, cols), np.float64(np.finfo(np.float32).max))
        for j in range(n):
            cx, cy = C[j, 3], C[j, 4]
            if np.isnan(cx):
                continue
            k0, l0 = int(cx - step), int(cy - step)        # truncation, as `int k = centers[j][3] - step`
            ks = np.arange(k0, int(np.ceil(cx + step)) + 1)
            ls = np.arange(l0, int(np.ceil(cy + step)) + 1)
            ks = ks[(ks < cx + step) & (ks >= 0) & (ks < cols)]
            ls = ls[(ls < cy + step) & (ls >= 0) & (ls < rows)]
            if len(ks) == 0 or len(ls) == 0:
                continue
            win = f[ls[0]:ls[-1] + 1, ks[0]:ks[-1] + 1]
            d0, d1, d2 = C[j, 0] - win[..., 0], C[j, 1] - win[..., 1], C[j, 2] - win[..., 2]
            dc = np.sqrt(d0 * d0 + d1 * d1 + d2 * d2)
            e0 = (C[j, 3] - ks.astype(np.float64))[None, :]
            e1 = (C[j, 4] - ls.astype(np.float64))[:, None]
            ds = np.sqrt(e0 * e0 + e1 * e1)
            a, b = dc / np.float64(nc), ds / np.float64(ns)
            d = np.sqrt(a * a + b * b)
            sub_d = dist[ls[0]:ls[-1] + 1, ks[0]:ks[-1] + 1]
            sub_l = labels[ls[0]:ls[-1] + 1, ks[0]:ks[-1] + 1]
            better = d < sub_d                              # strict: ties stay with the lower centre index
            sub_d[better] = d[better]
            sub_l[better] = j
        yy, xx = np.indices((rows, cols))
        ok = labels >= 0
        idx = labels[ok]
        cnt = np.bincount(idx, minlength=n).astype(np.float64)
        newC = np.empty_like(C)
        for q, vals in enumerate((f[..., 0][ok], f[..., 1][ok], f[..., 2][ok], xx[ok].astype(np.float64), yy[ok].astype(np.float64))):
            with np.errstate(divide="ignore", invalid="ignore"):
                newC[:, q] = np.bincount(idx, weights=vals, minlength=n) / cnt      # integer-valued sums: exact in float64
        C = newC
    return labels, n, C


def stereo_refine(depth, left, right, baseline=0.54, focal=9.597910e+02, damp=500.0, max_depth=100.0, iterations=4):
    """N4, SL/main_sl.cpp:715-885 as driven from :1165-1246, whole-array: every pixel only touches its own disparity, so
    a sweep is one vector update.  f32 throughout (each numpy op rounds once); see dcmt_oracle.c for the citations."""
    d = np.asarray(depth, dtype=F32)
    rows, cols = d.shape
    fe = rows * cols
    gl = np.asarray(left, dtype=np.uint8).astype(F32).reshape(-1)
    gr = np.asarray(right, dtype=np.uint8).astype(F32).reshape(-1)
    g2 = gr.reshape(rows, cols)
    dxi = np.zeros((rows, cols), F32)                      # calculateMeasuementDerivatives: interior only
    dxi[1:-1, 1:-1] = F32(0.5) * g2[1:-1, 2:] - F32(0.5) * g2[1:-1, :-2]
    dxr = dxi.reshape(-1)
    bf = F32(baseline) * F32(focal)
    disp = np.zeros((rows, cols), F32)
    with np.errstate(divide="ignore"):
        disp[d > 0] = bf / d[d > 0]
    ii, jj = np.indices((rows, cols))
    for _ in range(iterations):
        c = jj.astype(F32) - disp
        c0 = np.trunc(c.astype(np.float64) + 0.5).astype(np.int64)
        e0 = ii * cols + c0
        e1 = e0 + 1
        ok = (c0 >= 0) & (c0 + 1 <= cols) & (disp != 0) & (e1 < fe)
        e0c, e1c = np.where(ok, e0, 0), np.where(ok, e1, 0)
        dc = c - c0.astype(F32)
        dc1 = F32(1.0) - dc
        value = gr[e0c] * dc1 + gr[e1c] * dc
        dx = dxr[e0c] * dc1 + dxr[e1c] * dc
        err = np.clip(value - gl.reshape(rows, cols), F32(-255.0), F32(255.0))
        jcr = F32(-1.0) * dx
        H = jcr * jcr + F32(damp)
        b = jcr * err
        disp = np.where(ok, disp + (-b / H), disp).astype(F32)
    out = np.zeros((rows, cols), F32)
    pos = disp > 0
    with np.errstate(divide="ignore"):
        out[pos] = np.minimum(bf / disp[pos], F32(max_depth))
    return out


def median5(x):
    """cv::medianBlur(x,x,5) on f32: exact median, BORDER_REPLICATE (LO :170)."""
    R, C = x.shape
    p = np.pad(x, 2, mode="edge")
    stack = np.stack([p[r:r + R, c:c + C] for r in range(5) for c in range(5)], axis=0)
    return np.sort(stack, axis=0)[12].astype(F32)


def gaussian5(x):
    """cv::GaussianBlur(x,x,Size(5,5),0): [1,4,6,4,1]/16, REFLECT_101, f32 (LO :179)."""
    k0, k1, k2 = F32(0.375), F32(0.25), F32(0.0625)
    R, C = x.shape

    def pad101(a, axis):
        n = a.shape[axis]
        if n == 1:
            return np.concatenate([a] * 5, axis=axis)
        idx = np.arange(-2, n + 2)
        for _ in range(4):
            idx = np.where(idx < 0, -idx, idx)
            idx = np.where(idx >= n, 2 * n - 2 - idx, idx)
        return np.take(a, idx, axis=axis)

    p = pad101(x, 1)
    t = (p[:, 2:2 + C] * k0).astype(F32)
    t = (t + ((p[:, 1:1 + C] + p[:, 3:3 + C]).astype(F32) * k1).astype(F32)).astype(F32)
    t = (t + ((p[:, 0:C] + p[:, 4:4 + C]).astype(F32) * k2).astype(F32)).astype(F32)
    p = pad101(t, 0)
    o = (p[2:2 + R] * k0).astype(F32)
    o = (o + ((p[1:1 + R] + p[3:3 + R]).astype(F32) * k1).astype(F32)).astype(F32)
    o = (o + ((p[0:R] + p[4:4 + R]).astype(F32) * k2).astype(F32)).astype(F32)
    return o


def _tail(x, stop_after, blur, max_fill_iters, info):
    x, _ = fill(x, 7)                                   # H5
    if stop_after <= 5:
        return x
    x = extend_columns(x)                               # H6
    if stop_after <= 6:
        return x
    x, n = fill(x, 31)                                  # H7
    info["holes_after_extend"] = n
    if stop_after <= 7:
        return x
    iters = 0
    while True:                                         # H8
        x, n = fill(x, 31)
        iters += 1
        if n == 0 or iters >= max_fill_iters:
            break
    info["fill_iters"] = iters
    if stop_after <= 8:
        return x
    x = median5(x)                                      # H9
    if stop_after <= 9:
        return x
    if blur == "gaussian":                              # H10
        g = gaussian5(x)
        x = np.where(_valid(x), g, x).astype(F32)
    if stop_after <= 10:
        return x
    return invert(x)                                    # H11


def img_completion(sparse, k0=K0_AS_COMPILED, blur="gaussian", stop_after=11, max_fill_iters=64,
                   info=None):
    info = {} if info is None else info
    x = invert(np.asarray(sparse, dtype=F32))           # H0, H2
    if stop_after <= 2:
        return x
    x = dilate(x, k0)                                   # H3
    if stop_after <= 3:
        return x
    x = erode(dilate(x, _ones(5)), _ones(5))            # H4
    if stop_after <= 4:
        return x
    return _tail(x, stop_after, blur, max_fill_iters, info)


def interpolate_with_superpixels(sparse, labels, n_labels, k0=K0_AS_COMPILED, use_superpixel=1,
                                 stop_after=11, max_fill_iters=64, info=None):
    """LC :34-203 (one whole-image pass per label, literally)."""
    info = {} if info is None else info
    x = invert(np.asarray(sparse, dtype=F32))
    if stop_after <= 2:
        return x
    if use_superpixel == 0:
        x = dilate(x, k0)
        x = erode(dilate(x, _ones(5)), _ones(5))
    else:
        labels = np.asarray(labels)
        for c in range(n_labels):
            m = labels == c
            if not m.any():
                continue
            region = np.where(m, x, F32(0)).astype(F32)
            region = erode(dilate(dilate(region, k0), _ones(5)), _ones(5))
            x = np.where(m, region, x).astype(F32)
    if stop_after <= 4:
        return x
    return _tail(x, stop_after, "gaussian", max_fill_iters, info)
