"""Generates the golden vectors in this directory.

Run from the repo root:  python tests/golden/make_golden.py

The vectors are produced by oracle/np_restatement.py (the numpy restatement of the
reference cascade); the C oracle and the HIP path must both reproduce them (bit-exact
through the median stage; the Gaussian stage bit-exact against these two restatements and
within 1e-4 of any other conforming OpenCV build).  PARITY UNPINNED: the
reference ships no fixtures and OpenCV cannot be executed in this image, so these vectors
pin the restated semantics, not an OpenCV run.  Nothing here reads /root/reference.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from depth_completion_mt_amd import synth  # noqa: E402
from oracle import np_restatement as N  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def adversarial_cases():
    """Small inputs that force the rare branches."""
    cases = {}
    z = np.zeros((24, 40), np.float32)
    cases["all_empty"] = z.copy()                                   # K1
    a = z.copy(); a[12, 20] = 10.0
    cases["single_pixel"] = a                                       # K2 (small version)
    a = synth.synth_frame(40, 56, 11); a[:, 10:14] = 0; a[:, 55] = 0
    cases["empty_columns"] = a                                      # K6
    a = synth.synth_frame(40, 56, 12)
    a[5, 5] = 99.95; a[6, 7] = 100.0; a[30, 30] = 120.0; a[31, 31] = 99.9
    cases["near_max_depth"] = a                                     # K7
    a = z.copy(); a[3, 3] = np.float32(0.1); a[3, 20] = np.nextafter(np.float32(0.1), np.float32(0))
    a[20, 3] = 0.5; a[20, 30] = 0.100000024
    cases["threshold"] = a                                          # K4
    a = synth.synth_frame(40, 56, 13); a[a == 0] = 7.5
    cases["dense"] = a
    a = z.copy(); a[0, :] = 20.0; a[-1, :] = 30.0; a[:, 0] = 5.0; a[:, -1] = 6.0
    cases["border_only"] = a
    # tall gap: valid bands top and bottom only -> several H8 iterations
    a = np.zeros((200, 48), np.float32); a[0:2, :] = 40.0; a[198:200, :] = 12.0
    cases["tall_gap"] = a
    # never converges within the cap: a wide region whose neighbourhood max stays < 0.1
    a = np.zeros((120, 120), np.float32); a[:, :] = 120.0   # all > max_depth -> negative
    a[0, :] = 10.0; a[-1, :] = 10.0
    cases["negative_region"] = a
    a = np.float32(3.25) * np.ones((1, 9), np.float32); a[0, 4] = 0
    cases["one_row"] = a
    a = np.float32(3.25) * np.ones((9, 1), np.float32); a[4, 0] = 0
    cases["one_col"] = a
    cases["tiny_3x4"] = np.array([[0, 5, 0, 0], [0, 0, 0, 9], [1, 0, 0, 0]], np.float32)
    return cases


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def main():
    out = {}
    # per-stage goldens on a crop
    x = synth.synth_frame(48, 64, 3)
    out["crop48x64_in"] = x
    for st in range(2, 12):
        out[f"crop48x64_stage{st}"] = N.img_completion(x, stop_after=st)
    out["crop48x64_diamond"] = N.img_completion(x, k0=N.K0_DIAMOND)
    out["crop48x64_noblur"] = N.img_completion(x, blur="none")
    x = synth.synth_frame(33, 70, 7)
    out["odd33x70_in"] = x
    out["odd33x70_out"] = N.img_completion(x)
    out["odd33x70_diamond"] = N.img_completion(x, k0=N.K0_DIAMOND)
    # adversarial
    meta = {"adversarial": {}}
    for name, a in adversarial_cases().items():
        info = {}
        out[f"adv_{name}_in"] = a
        out[f"adv_{name}_out"] = N.img_completion(a, info=info, max_fill_iters=8)
        meta["adversarial"][name] = info
    # label-masked variant (LC)
    x = synth.synth_frame(40, 56, 21)
    lab, nl = synth.synth_labels(40, 56, 12, 21)
    out["lc40x56_in"] = x
    out["lc40x56_labels"] = lab
    out["lc40x56_stage4"] = N.interpolate_with_superpixels(x, lab, nl, stop_after=4)
    out["lc40x56_out"] = N.interpolate_with_superpixels(x, lab, nl)
    out["lc40x56_out_nosp"] = N.interpolate_with_superpixels(x, lab, nl, use_superpixel=0)
    meta["lc40x56_n_labels"] = nl
    # N1: min-max normalisation in front of the path (SL/main_sl.cpp:370 with (0,100), :523 with (0,80))
    x = synth.synth_frame(48, 64, 31)
    out["norm48x64_in"] = x
    out["norm48x64_n100"] = N.normalize_minmax(x, 0, 100)
    out["norm48x64_out100"] = N.img_completion(out["norm48x64_n100"])
    out["norm48x64_stage2_100"] = N.img_completion(out["norm48x64_n100"], stop_after=2)
    out["norm48x64_out80_diamond"] = N.img_completion(N.normalize_minmax(x, 0, 80), k0=N.K0_DIAMOND)
    x = synth.synth_frame(40, 56, 21)
    out["norm_lc40x56_out80"] = N.interpolate_with_superpixels(N.normalize_minmax(x, 0, 80), lab, nl)
    # no empty pixel, negative values: smin != 0, so the shift term is live (tolerance class, see dcmt_oracle.c)
    x = synth.synth_frame(40, 56, 13); x[x == 0] = 7.5; x[3, 4] = -2.25
    out["norm_dense40x56_in"] = x
    out["norm_dense40x56_n80"] = N.normalize_minmax(x, 0, 80)
    out["norm_dense40x56_out80"] = N.img_completion(out["norm_dense40x56_n80"])
    out["norm_flat_n"] = N.normalize_minmax(np.full((8, 8), 2.0, np.float32), 5, 80)     # constant frame -> dmin everywhere
    # N2: LiDAR points -> sparse depth image (SL/main_sl.cpp:478-520) on a 48x64 image: a synthetic sweep through a
    # camera with an 80-pixel focal length, plus hand-placed points for the rules (file order wins a pixel, z <= 0 is
    # dropped, the bounds are tested on the float coordinates before truncation)
    T = synth.KITTI_T_VELO_TO_CAM.copy()
    P = np.array([[80.0, 0.0, 32.0, 4.0], [0.0, 80.0, 20.0, 0.02], [0.0, 0.0, 1.0, 0.003]], np.float32)
    pts = synth.synth_points(4000, 41)
    extra = np.array([[10.0, 0.0, 0.0, 0.5], [20.0, 0.0, 0.0, 0.5], [15.0, 0.0, 0.0, 0.5],      # same pixel three times: the last (15 m) stays
                      [-5.0, 0.0, 0.0, 0.5],                                                   # behind the camera
                      [0.2717806, 0.0, 0.0, 0.5],                                              # t.z within rounding of 0
                      [10.0, 4.5, 0.0, 0.5], [10.0, -3.6, 0.0, 0.5],                           # near the left / right image edge
                      [10.0, 0.0, 2.6, 0.5], [10.0, 0.0, -3.3, 0.5]], np.float32)              # near the top / bottom edge
    pts = np.concatenate([pts[:2000], extra, pts[2000:], extra[:3][::-1]])
    out["proj_points"] = pts
    out["proj_T"] = T
    out["proj_P"] = P
    out["proj_sparse48x64"] = N.project_points(pts, T, P, 48, 64)
    out["proj_chain48x64"] = N.img_completion(N.normalize_minmax(out["proj_sparse48x64"], 0, 100))   # SL :370-386 end to end
    # N3: SLIC labels (LC/slic.cpp:101-182) on a 96x160 synthetic 8-bit image, step 12, nc 40; and the chain the
    # lidar-camera executable runs behind it (main_lc.cpp:200, :220): labels -> interpolate_with_superpixels
    img = synth.synth_lab(96, 160, 5)
    slab, sn, scent = N.slic(img, 12, 40)
    out["slic_lab96x160"] = img
    out["slic_labels96x160"] = slab
    out["slic_centers96x160"] = scent
    meta["slic96x160_n"] = int(sn)
    x = synth.synth_frame(96, 160, 5)
    out["slic_chain96x160"] = N.interpolate_with_superpixels(x, slab, sn)
    # N4: the stereo refinement behind the path (SL/main_sl.cpp:715-885) on a 48x64 synthetic rectified pair; a smaller
    # focal length keeps the disparities inside the small image
    sl, sr, sg = synth.synth_stereo(48, 64, 9, focal=60.0)
    out["stereo_left48x64"], out["stereo_right48x64"], out["stereo_guess48x64"] = sl, sr, sg
    out["stereo_refined48x64"] = N.stereo_refine(sg, sl, sr, focal=60.0)
    out["stereo_roundtrip48x64"] = N.stereo_refine(sg, sl, sr, focal=60.0, iterations=0)
    np.savez_compressed(os.path.join(HERE, "small_cases.npz"), **out)

    # full-size frames: checksums only (inputs come from the generator)
    full = {}
    for rows, cols, seed in [(352, 1216, 0), (352, 1216, 1), (375, 1242, 1)]:
        x = synth.synth_frame(rows, cols, seed)
        info = {}
        y9 = N.img_completion(x, stop_after=9, info=info)
        y = N.img_completion(x)
        full[f"{rows}x{cols}_seed{seed}"] = {
            "in_sha256": sha(x), "stage9_sha256": sha(y9), "out_sha256": sha(y),
            "holes_after_extend": info["holes_after_extend"], "fill_iters": info["fill_iters"],
            "out_sum_f64": float(y.astype(np.float64).sum()),
        }
    lab, nl = synth.synth_labels(352, 1216, 1200, 0)
    x = synth.synth_frame(352, 1216, 0)
    y = N.interpolate_with_superpixels(x, lab, nl)
    full["lc_352x1216_seed0"] = {"labels_sha256": hashlib.sha256(lab.tobytes()).hexdigest(),
                                 "n_labels": nl, "out_sha256": sha(y)}
    meta["full"] = full
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
