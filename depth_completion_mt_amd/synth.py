"""Deterministic KITTI-like synthetic sparse-depth frames.

Workload generation for tests and bench.py: 352x1216 f32, 0 = empty, ~4-5 % valid,
upper ~30 % of the image empty (like a projected velodyne scan), depths 0.5-85 m
quantised to 1/256 m (KITTI uint16 PNG / 256, reference src/DC_lidar_only/main.cpp:79).
Counter-based (SplitMix64 keyed by seed,row,col) so any frame can be produced anywhere
without a stream state; oracle/dcmt_oracle.c:dcmt_oracle_synth_frame is the same
function in C and tests check the two agree bit for bit.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_frame(rows: int = 352, cols: int = 1216, seed: int = 0) -> np.ndarray:
    key = _splitmix64(np.array([seed], dtype=np.uint64))[0]
    r = np.arange(rows, dtype=np.int64)
    re = (r * 352) // rows
    t = np.clip((re - 110) / 60.0, 0.0, 1.5)
    thr = np.floor(t * 0.05 * 16777216.0).astype(np.uint32)
    den = np.maximum(re - 172, 2)
    base = np.clip((1.65 * 721.5) / den, 2.0, 85.0)
    rc = (r.astype(np.uint64)[:, None] << np.uint64(32)) | np.arange(cols, dtype=np.uint64)[None, :]
    h = _splitmix64(key ^ rc)
    u1 = ((h >> np.uint64(40)) & np.uint64(0xFFFFFF)).astype(np.uint32)
    u2 = ((h >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float64)
    f = 0.6 + 0.8 * (u2 / 16777216.0)
    d = np.clip(base[:, None] * f, 0.5, 85.0)
    q = np.floor(d * 256.0 + 0.5)
    v = (q / 256.0).astype(np.float32)
    return np.where(u1 < thr[:, None], v, np.float32(0)).astype(np.float32)


def synth_batch(n: int, rows: int = 352, cols: int = 1216, seed0: int = 0) -> np.ndarray:
    """Frames seed0 .. seed0+n-1, shape [n][rows][cols]."""
    out = np.empty((n, rows, cols), dtype=np.float32)
    for i in range(n):
        out[i] = synth_frame(rows, cols, seed0 + i)
    return out


def synth_labels(rows: int = 352, cols: int = 1216, n_target: int = 1200, seed: int = 0) -> tuple[np.ndarray, int]:
    """SLIC-like label plane: a jittered grid of ~n_target cells (step = sqrt(W*H/n), as
    reference src/DC_lidar_camera/main_lc.cpp:188-197), int32[rows][cols], plus n_labels.
    Cell borders are perturbed per pixel so labels interleave like real superpixels."""
    step = max(2, int(np.sqrt(rows * cols / float(n_target))))
    gy, gx = (rows + step - 1) // step, (cols + step - 1) // step
    key = _splitmix64(np.array([seed ^ 0x5EED], dtype=np.uint64))[0]
    r = np.arange(rows, dtype=np.uint64)[:, None]
    c = np.arange(cols, dtype=np.uint64)[None, :]
    h = _splitmix64(key ^ ((r << np.uint64(32)) | c))
    jr = ((h >> np.uint64(8)) % np.uint64(5)).astype(np.int64) - 2
    jc = ((h >> np.uint64(24)) % np.uint64(5)).astype(np.int64) - 2
    rr = np.clip(np.arange(rows, dtype=np.int64)[:, None] + jr, 0, rows - 1)
    cc = np.clip(np.arange(cols, dtype=np.int64)[None, :] + jc, 0, cols - 1)
    lab = (rr // step) * gx + (cc // step)
    # a sprinkle of unassigned pixels (-1), as SLIC leaves some
    lab = np.where((h >> np.uint64(50)) % np.uint64(997) == 0, -1, lab)
    return lab.astype(np.int32), int(gy * gx)


# ---- N2: synthetic velodyne sweep + KITTI-like calibration (test / bench input only) ----------------
# velodyne frame: x forward, y left, z up; camera frame: x right, y down, z forward
KITTI_T_VELO_TO_CAM = np.array([[7.533745e-03, -9.999714e-01, -6.166020e-04, -4.069766e-03],
                                [1.480249e-02, 7.280733e-04, -9.998902e-01, -7.631618e-02],
                                [9.998621e-01, 7.523790e-03, 1.480755e-02, -2.717806e-01],
                                [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)
KITTI_P2 = np.array([[7.215377e+02, 0.0, 6.095593e+02, 4.485728e+01],
                     [0.0, 7.215377e+02, 1.728540e+02, 2.163791e-01],
                     [0.0, 0.0, 1.0, 2.745884e-03]], dtype=np.float32)


def synth_points(n: int, seed: int) -> np.ndarray:
    """[n][4] f32 (x, y, z, reflectance): an HDL-64-like sweep -- azimuth over the full circle (so about three quarters
    of the points fall behind or beside the camera, as in a real .bin), elevation -24.9..2 degrees, ranges 3..80 m
    cut off at the ground plane 1.73 m below the sensor."""
    g = np.random.Generator(np.random.PCG64(seed))
    az = g.uniform(-np.pi, np.pi, n)
    el = np.deg2rad(g.uniform(-24.9, 2.0, n))
    rng = g.uniform(3.0, 80.0, n)
    ground = np.where(el < 0, 1.73 / np.maximum(np.sin(-el), 1e-6), np.inf)
    rng = np.minimum(rng, ground)
    pts = np.empty((n, 4), np.float32)
    pts[:, 0] = rng * np.cos(el) * np.cos(az)
    pts[:, 1] = rng * np.cos(el) * np.sin(az)
    pts[:, 2] = rng * np.sin(el)
    pts[:, 3] = g.uniform(0.0, 1.0, n)
    return pts


def synth_lab(rows: int, cols: int, seed: int) -> np.ndarray:
    """[rows][cols][3] uint8: a stand-in for cv::cvtColor(BGR2Lab) output -- smooth colour regions (a random coarse
    grid, bilinearly interpolated) with edges (a few rectangles of constant colour) and a little noise."""
    g = np.random.Generator(np.random.PCG64(seed + 7919))
    gy, gx = rows // 40 + 2, cols // 40 + 2
    coarse = g.uniform(30, 220, (gy, gx, 3))
    ys = np.linspace(0, gy - 1.001, rows)
    xs = np.linspace(0, gx - 1.001, cols)
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    img = (coarse[y0][:, x0] * (1 - fy) * (1 - fx) + coarse[y0][:, x0 + 1] * (1 - fy) * fx
           + coarse[y0 + 1][:, x0] * fy * (1 - fx) + coarse[y0 + 1][:, x0 + 1] * fy * fx)
    for _ in range(12):
        r0, c0 = int(g.integers(0, rows - 8)), int(g.integers(0, cols - 8))
        r1, c1 = min(rows, r0 + int(g.integers(8, rows // 3 + 9))), min(cols, c0 + int(g.integers(8, cols // 4 + 9)))
        img[r0:r1, c0:c1] = g.uniform(20, 235, 3)
    img += g.normal(0, 2.0, img.shape)
    return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))


def synth_stereo(rows: int, cols: int, seed: int, baseline: float = 0.54, focal: float = 959.791):
    """(left uint8 [rows][cols], right uint8, depth_guess f32): a textured scene (sum of random sinusoids, so it can be
    sampled at fractional positions) seen by a rectified pair; the right image shows the texture shifted by the true
    disparity, and the depth guess is the true depth with a few per cent of smooth error -- what the stereo
    refinement (N4) starts from.  Some guess pixels are 0 (no depth), as in the path's all-zero columns."""
    g = np.random.Generator(np.random.PCG64(seed + 104729))
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float64)
    depth = 6.0 + 60.0 * (1.0 - yy / rows) ** 2 + 4.0 * np.sin(xx / 90.0 + seed)          # far at the top, near at the bottom
    depth[rows // 3: rows // 2, cols // 4: cols // 3] = 9.0                                # a near object: a disparity step
    disp = baseline * focal / depth
    waves = [(g.uniform(0.02, 0.35), g.uniform(0.02, 0.35), g.uniform(0, 6.28), g.uniform(8, 30)) for _ in range(10)]
    tex = lambda x, y: 128.0 + sum(a * np.sin(fx * x + fy * y + ph) for fx, fy, ph, a in waves)
    left = np.clip(np.rint(tex(xx, yy) + g.normal(0, 1.0, (rows, cols))), 0, 255).astype(np.uint8)
    right = np.clip(np.rint(tex(xx + disp, yy) + g.normal(0, 1.0, (rows, cols))), 0, 255).astype(np.uint8)
    guess = (depth * (1.0 + 0.04 * np.sin(xx / 37.0) * np.cos(yy / 23.0))).astype(np.float32)
    guess[:, : cols // 50] = 0.0
    guess[g.random((rows, cols)) < 0.002] = 0.0
    return np.ascontiguousarray(left), np.ascontiguousarray(right), np.ascontiguousarray(guess)
