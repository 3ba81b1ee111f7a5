"""CPU tests of the oracle: C restatement == numpy restatement == committed goldens,
plus the known-answer tests derived in SURVEY.md section 8c (K1-K8).

PARITY UNPINNED: these pin the restated OpenCV semantics, not an OpenCV run."""
import hashlib

import numpy as np
import pytest

from conftest import assert_bit_equal
from depth_completion_mt_amd import synth
from oracle import np_restatement as N
from oracle import oracle as O

FLT_MAX = np.finfo(np.float32).max


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def test_synth_generator_c_matches_numpy():
    for rows, cols, seed in [(352, 1216, 0), (375, 1242, 1), (48, 64, 3), (33, 70, 7), (5, 5, 99)]:
        assert_bit_equal(synth.synth_frame(rows, cols, seed), O.synth_frame(rows, cols, seed), f"synth {rows}x{cols}")
    a = synth.synth_frame(352, 1216, 0)
    assert 0.03 < (a > 0).mean() < 0.06
    assert (a[:100] == 0).all()                       # upper part empty, like a velodyne scan
    assert np.array_equal(a * 256, np.round(a * 256))  # quantised to 1/256 m


def test_k3_as_compiled_kernel_bytes():
    want = np.zeros((5, 5), np.uint8)
    want[1, 3] = 1
    want[4, 4] = 1
    assert np.array_equal(O.k0_as_compiled(), want)
    assert np.array_equal(N.K0_AS_COMPILED, want)
    assert O.k0_diamond().sum() == 13 and np.array_equal(O.k0_diamond(), N.K0_DIAMOND)


def test_per_stage_goldens(golden):
    x = golden["crop48x64_in"]
    assert_bit_equal(x, synth.synth_frame(48, 64, 3), "golden input")
    for st in range(2, 12):
        got = O.img_completion(x, O.default_params(stop_after=st))
        assert_bit_equal(got, golden[f"crop48x64_stage{st}"], f"C oracle stage {st}")
    assert_bit_equal(O.img_completion(x, O.default_params(k0="diamond")), golden["crop48x64_diamond"], "diamond")
    assert_bit_equal(O.img_completion(x, O.default_params(blur="none")), golden["crop48x64_noblur"], "noblur")
    assert_bit_equal(O.img_completion(golden["odd33x70_in"]), golden["odd33x70_out"], "33x70")
    assert_bit_equal(O.img_completion(golden["odd33x70_in"], O.default_params(k0="diamond")),
                     golden["odd33x70_diamond"], "33x70 diamond")


def test_adversarial_goldens(golden, golden_meta):
    for name, info in golden_meta["adversarial"].items():
        x = golden[f"adv_{name}_in"]
        got, ginfo = O.img_completion(x, O.default_params(max_fill_iters=8), return_info=True)
        assert_bit_equal(got, golden[f"adv_{name}_out"], f"adversarial {name}")
        assert ginfo["fill_iters"] == info["fill_iters"], name
        assert ginfo["holes_after_extend"] == info["holes_after_extend"], name


def test_numpy_restatement_reproduces_goldens(golden):
    """make_golden.py is deterministic: regenerating must give the committed bytes."""
    x = golden["crop48x64_in"]
    for st in (4, 8, 9, 11):
        assert_bit_equal(N.img_completion(x, stop_after=st), golden[f"crop48x64_stage{st}"], f"np stage {st}")


def test_full_size_checksums(golden_meta):
    for key, m in golden_meta["full"].items():
        if key.startswith("lc_"):
            continue
        dims, seed = key.split("_seed")
        rows, cols = (int(v) for v in dims.split("x"))
        x = synth.synth_frame(rows, cols, int(seed))
        assert sha(x) == m["in_sha256"]
        y9 = O.img_completion(x, O.default_params(stop_after=O.STAGE_MEDIAN5))
        assert sha(y9) == m["stage9_sha256"], key
        y, info = O.img_completion(x, return_info=True)
        assert sha(y) == m["out_sha256"], key
        assert info["fill_iters"] == m["fill_iters"] and info["holes_after_extend"] == m["holes_after_extend"]


def test_k1_all_empty_gives_all_zero():
    y = O.img_completion(np.zeros((64, 96), np.float32))
    assert (y == 0).all()


def test_k2_single_pixel_known_answer():
    x = np.zeros((352, 1216), np.float32)
    x[200, 600] = 10.0
    y = O.img_completion(x)
    vals = set(np.unique(y).tolist())
    assert vals == {0.0, 0.625, 3.125, 6.875, 9.375, 10.0}


def test_k4_threshold_is_ge_0p1f():
    below = np.nextafter(np.float32(0.1), np.float32(0))
    x = np.zeros((1, 2), np.float32)
    x[0, 0], x[0, 1] = np.float32(0.1), below
    y = O.img_completion(x, O.default_params(stop_after=O.STAGE_INVERT))
    assert y[0, 0] == np.float32(100.0) - np.float32(0.1)   # 0.1f is valid
    assert y[0, 1] == below                                  # 0.099999994f is empty


def test_k6_empty_column_becomes_100_in_inverted_space():
    x = synth.synth_frame(40, 56, 11)
    x[:, 20] = 0
    # make sure nothing can spill into column 20 before the extension: clear a 10-px band
    x[:, 10:31] = 0
    y = O.img_completion(x, O.default_params(stop_after=O.STAGE_EXTEND))
    assert (y[:, 20] == 100.0).all()


def test_k7_near_max_depth():
    x = np.zeros((1, 3), np.float32)
    x[0] = [99.95, 120.0, 100.0]
    y = O.img_completion(x, O.default_params(stop_after=O.STAGE_INVERT))[0]
    assert y[0] < 0.1 and y[0] > 0          # 99.95 -> 0.05: a hole from now on
    assert y[1] == -20.0                    # beyond max_depth -> negative
    assert y[2] == 0.0


def test_k8_last_column_sentinel_vanishes_after_close():
    x = synth.synth_frame(48, 64, 3)
    y3 = O.img_completion(x, O.default_params(stop_after=O.STAGE_DILATE_K))
    assert (y3[:, -1] == -FLT_MAX).all()    # both taps of the as-compiled element are outside
    y4 = O.img_completion(x, O.default_params(stop_after=O.STAGE_CLOSE5))
    assert (y4 > -FLT_MAX).all()


def test_separable_equals_bruteforce():
    rng = np.random.default_rng(5)
    for shape in [(23, 41), (40, 33), (1, 50), (50, 1), (7, 7)]:
        a = rng.normal(0, 30, shape).astype(np.float32)
        for k in (5, 7, 31):
            assert_bit_equal(O.dilate_rect(a, k), O.dilate_rect(a, k, bruteforce=True), f"dilate {k} {shape}")
            assert_bit_equal(O.erode_rect(a, k), O.erode_rect(a, k, bruteforce=True), f"erode {k} {shape}")
            assert_bit_equal(O.dilate_rect(a, k), N.dilate(a, np.ones((k, k), np.uint8)), f"np dilate {k}")
        assert_bit_equal(O.median5(a), N.median5(a), f"median {shape}")
        assert_bit_equal(O.median5(a, simple=True), N.median5(a), f"median simple {shape}")
        assert_bit_equal(O.gaussian5(a), N.gaussian5(a), f"gauss {shape}")
        assert_bit_equal(O.dilate_mask5(a, N.K0_AS_COMPILED), N.dilate(a, N.K0_AS_COMPILED), "mask dilate")
        assert_bit_equal(O.dilate_mask5(a, N.K0_DIAMOND), N.dilate(a, N.K0_DIAMOND), "diamond dilate")
        assert_bit_equal(O.extend_columns(a), N.extend_columns(a), f"extend {shape}")


def test_gaussian_constant_and_impulse():
    c = np.full((9, 11), 7.25, np.float32)
    assert (O.gaussian5(c) == 7.25).all()             # dyadic weights sum to 1 exactly
    imp = np.zeros((9, 9), np.float32)
    imp[4, 4] = 256.0
    k = np.array([1, 4, 6, 4, 1], np.float32)
    assert np.array_equal(O.gaussian5(imp)[2:7, 2:7], np.outer(k, k))


def test_fill_loop_cap_reports_nonconvergence(golden):
    x = golden["adv_tall_gap_in"]
    _, info = O.img_completion(x, O.default_params(max_fill_iters=2), return_info=True)
    assert info["rc"] == -1 and info["fill_iters"] == 2
    _, info = O.img_completion(x, O.default_params(max_fill_iters=64), return_info=True)
    assert info["rc"] == 0 and info["fill_iters"] == 7


def test_lc_goldens_and_roi_equals_bruteforce(golden, golden_meta):
    x, lab = golden["lc40x56_in"], golden["lc40x56_labels"]
    nl = golden_meta["lc40x56_n_labels"]
    for brute in (False, True):
        assert_bit_equal(O.interpolate_with_superpixels(x, lab, nl, O.default_params(stop_after=O.STAGE_CLOSE5),
                                                        bruteforce=brute), golden["lc40x56_stage4"], "lc stage4")
        assert_bit_equal(O.interpolate_with_superpixels(x, lab, nl, bruteforce=brute), golden["lc40x56_out"], "lc out")
    assert_bit_equal(O.interpolate_with_superpixels(x, lab, nl, use_superpixel=0), golden["lc40x56_out_nosp"], "nosp")
    # use_superpixel=0 is the plain chain with the Gaussian forced on
    assert_bit_equal(golden["lc40x56_out_nosp"], O.img_completion(x), "nosp == img_completion")
    # random labels (adversarial: every pixel its own neighbourhood of labels)
    rng = np.random.default_rng(3)
    lab2 = rng.integers(-1, 9, size=x.shape).astype(np.int32)
    assert_bit_equal(O.interpolate_with_superpixels(x, lab2, 8),
                     O.interpolate_with_superpixels(x, lab2, 8, bruteforce=True), "random labels roi vs brute")
    assert_bit_equal(O.interpolate_with_superpixels(x, lab2, 8), N.interpolate_with_superpixels(x, lab2, 8), "np")


def test_lc_full_size_checksum(golden_meta):
    m = golden_meta["full"]["lc_352x1216_seed0"]
    lab, nl = synth.synth_labels(352, 1216, 1200, 0)
    assert nl == m["n_labels"] and hashlib.sha256(lab.tobytes()).hexdigest() == m["labels_sha256"]
    y = O.interpolate_with_superpixels(synth.synth_frame(352, 1216, 0), lab, nl)
    assert sha(y) == m["out_sha256"]


def test_batch_matches_single_and_threads():
    frames = synth.synth_batch(3, 48, 64, 100)
    got1, rc1 = O.img_completion_batch(frames, threads=1)
    got2, rc2 = O.img_completion_batch(frames, threads=3)
    assert rc1 == 0 and rc2 == 0
    for i in range(3):
        assert_bit_equal(got1[i], O.img_completion(frames[i]), f"batch frame {i}")
    assert_bit_equal(got1, got2, "threads")


def test_n1_normalize_goldens_and_known_answers(golden):
    """N1 (cv::normalize NORM_MINMAX in front of the path, SL/main_sl.cpp:370 / :523): the C oracle against the
    numpy restatement's goldens, and the properties the formula guarantees."""
    from oracle import oracle as O
    x = golden["norm48x64_in"]
    n100 = O.normalize_minmax(x, 0, 100)
    assert_bit_equal(n100, golden["norm48x64_n100"], "normalize (0,100)")
    assert_bit_equal(O.img_completion(n100), golden["norm48x64_out100"], "chain on normalised frame")
    assert_bit_equal(O.img_completion(n100, O.default_params(stop_after=2)), golden["norm48x64_stage2_100"], "H2 on normalised frame")
    assert_bit_equal(O.img_completion(O.normalize_minmax(x, 0, 80), O.default_params(k0="diamond")), golden["norm48x64_out80_diamond"], "(0,80) diamond")
    xd = golden["norm_dense40x56_in"]
    assert_bit_equal(O.normalize_minmax(xd, 0, 80), golden["norm_dense40x56_n80"], "normalize with smin != 0")
    assert_bit_equal(O.normalize_minmax(np.full((8, 8), 2.0, np.float32), 5, 80), golden["norm_flat_n"], "flat frame")
    # empty pixels stay exactly 0 (shift == 0), the maximum lands within one rounding of `hi`, the argument order is irrelevant
    assert (n100[x == 0] == 0).all() and abs(float(n100.max()) - 100.0) <= 1e-5
    assert_bit_equal(O.normalize_minmax(x, 100, 0), n100, "cv::normalize takes min/max of (alpha, beta)")
    assert (golden["norm_flat_n"] == 5.0).all()                 # smax - smin <= DBL_EPSILON: scale 0, everything = dmin
    # scaling by a power of two commutes exactly with the normalisation
    assert_bit_equal(O.normalize_minmax(x * np.float32(4.0), 0, 100), n100, "power-of-two input scale")


def test_n2_projection_goldens_and_rules(golden):
    """N2 (SL/main_sl.cpp:478-520, LiDAR points -> sparse depth image): C oracle == numpy restatement's goldens, and
    the rules of the reference loop one by one."""
    from oracle import oracle as O
    from depth_completion_mt_amd import synth
    pts, T, P = golden["proj_points"], golden["proj_T"], golden["proj_P"]
    got = O.project_points(pts, T, P, 48, 64)
    assert_bit_equal(got, golden["proj_sparse48x64"], "projection 48x64")
    assert_bit_equal(O.img_completion(O.normalize_minmax(got, 0, 100)), golden["proj_chain48x64"], "project -> normalize -> complete")
    one = lambda rows: O.project_points(np.asarray(rows, np.float32), T, P, 48, 64)
    a = one([[10, 0, 0, 0], [20, 0, 0, 0]])
    b = one([[20, 0, 0, 0], [10, 0, 0, 0]])
    assert (a > 0).sum() == 1 and (b > 0).sum() == 1 and a.max() > 19 and b.max() < 11    # file order: the later point stays
    assert not one([[-5, 0, 0, 0]]).any()                                                    # behind the camera (t.z <= 0)
    assert not one([[10, 40, 0, 0]]).any() and not one([[10, 0, 30, 0]]).any()               # outside the image
    assert not O.project_points(np.zeros((0, 4), np.float32), T, P, 48, 64).any()             # empty sweep -> empty image
    # full-size sweep: KITTI-like density, and depth = z in the camera frame (third row of P is ~[0 0 1 0])
    full = O.project_points(synth.synth_points(120000, 0), synth.KITTI_T_VELO_TO_CAM, synth.KITTI_P2, 375, 1242)
    assert 0.02 < (full > 0).mean() < 0.06 and full.max() < 81.0


def test_n3_slic_goldens_and_properties(golden, golden_meta):
    """N3 (Slic::generate_superpixels, LC/slic.cpp:101-182): the C oracle against the numpy restatement's goldens
    (labels and the f64 centres bit for bit), and the properties the algorithm guarantees."""
    from oracle import oracle as O
    img = golden["slic_lab96x160"]
    labels, n, cent = O.slic(img, 12, 40, return_centers=True)
    assert n == golden_meta["slic96x160_n"] == 12 * 7
    assert np.array_equal(labels, golden["slic_labels96x160"])
    want = golden["slic_centers96x160"]
    dead = np.isnan(want[:, 3])                    # a centre that lost all its pixels is NaN from then on (here: one)
    assert dead.sum() == 1 and np.array_equal(np.isnan(cent[:, 3]), dead)
    assert np.array_equal(cent[~dead].view(np.uint64), want[~dead].view(np.uint64))
    # every centre is the mean of the pixels carrying its label (the last update step), labels are in range
    assert labels.min() >= -1 and labels.max() < n
    ys, xs = np.indices(labels.shape)
    for c in (0, 17, n - 1):
        m = labels == c
        assert m.any()
        assert cent[c, 3] == xs[m].sum() / m.sum() and cent[c, 4] == ys[m].sum() / m.sum()
        assert cent[c, 0] == img[..., 0][m].astype(np.int64).sum() / m.sum()
    # a flat image: the colour term vanishes, every pixel takes the spatially nearest centre, ties to the lower index
    flat = np.full((60, 90, 3), 128, np.uint8)
    lf, nf, cf = O.slic(flat, 10, 40, return_centers=True)
    d2 = (cf[:, 3][None, None, :] - np.indices((60, 90))[1][..., None]) ** 2 + (cf[:, 4][None, None, :] - np.indices((60, 90))[0][..., None]) ** 2
    assert np.array_equal(lf, d2.argmin(axis=2).astype(np.int32))          # converged: labels = Voronoi cells of the final centres
    assert_bit_equal(O.interpolate_with_superpixels(synth_frame_96(), labels, n), golden["slic_chain96x160"], "labels -> LC chain")


def synth_frame_96():
    from depth_completion_mt_amd import synth
    return synth.synth_frame(96, 160, 5)


def test_n4_stereo_refinement_goldens_and_rules(golden):
    """N4 (SL/main_sl.cpp:715-885): C oracle == the numpy restatement's goldens; the rules of the reference loop."""
    from oracle import oracle as O
    l, r, g = golden["stereo_left48x64"], golden["stereo_right48x64"], golden["stereo_guess48x64"]
    assert_bit_equal(O.stereo_refine(g, l, r, focal=60.0), golden["stereo_refined48x64"], "refined")
    rt = O.stereo_refine(g, l, r, focal=60.0, iterations=0)
    assert_bit_equal(rt, golden["stereo_roundtrip48x64"], "depth -> disparity -> depth")
    assert (rt[g == 0] == 0).all()                                   # no depth: no disparity, stays 0 (:852, :868)
    bf = np.float32(0.54) * np.float32(60.0)
    assert_bit_equal(rt[g > 0], np.minimum(bf / (bf / g[g > 0]), np.float32(100.0)), "round trip arithmetic")
    # identical images and a guess whose disparity is a whole number of pixels: the photometric error is zero, nothing moves
    flat = np.full((20, 40), bf / np.float32(3.0), np.float32)
    same = np.tile(l[:1, :40], (20, 1))
    shifted = np.roll(same, -3, axis=1)
    moved = O.stereo_refine(flat, same, shifted, focal=60.0)
    assert_bit_equal(moved[:, 4:36], O.stereo_refine(flat, same, shifted, focal=60.0, iterations=0)[:, 4:36], "zero error: no update")
    # a depth beyond max_depth after refinement is capped at 100 (:875-878)
    far = O.stereo_refine(np.full((20, 40), 400.0, np.float32), same, same, focal=60.0)
    assert far.max() == 100.0


def test_primitives_match_scipy_ndimage():
    """A third, independently written implementation of the primitives (scipy.ndimage, which shares no code with the two
    restatements): rectangular dilate / erode with the constant border sentinels, the 5x5 median with replicated border
    (bit-exact), the [1 4 6 4 1]/16 Gaussian with reflect-101 border (scipy accumulates in f64: within 1e-5), and the
    as-compiled first element, whose orientation (dst(p) = max src(p + k - anchor), un-reflected) scipy reproduces as
    a grey dilation with the footprint rotated by 180 degrees."""
    from scipy import ndimage as ndi
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    x = np.where(rng.random((57, 83)) < 0.3, rng.uniform(0.2, 90.0, (57, 83)), 0.0).astype(np.float32)
    fmax = np.float32(np.finfo(np.float32).max)
    for k in (5, 7, 31):
        assert_bit_equal(O.dilate_rect(x, k), ndi.maximum_filter(x, size=k, mode="constant", cval=-fmax), f"dilate {k}")
        assert_bit_equal(O.erode_rect(x, k), ndi.minimum_filter(x, size=k, mode="constant", cval=fmax), f"erode {k}")
    assert_bit_equal(O.median5(x), ndi.median_filter(x, size=5, mode="nearest"), "median 5x5 replicate")
    taps = np.array([1, 4, 6, 4, 1], np.float64) / 16.0
    g = ndi.correlate1d(ndi.correlate1d(x.astype(np.float64), taps, axis=1, mode="mirror"), taps, axis=0, mode="mirror")
    assert np.abs(O.gaussian5(x).astype(np.float64) - g).max() <= 1e-5
    k0 = O.k0_as_compiled()
    want = ndi.grey_dilation(x, footprint=k0[::-1, ::-1].astype(bool), mode="constant", cval=-fmax)
    assert_bit_equal(O.dilate_mask5(x, k0), want, "first element, un-reflected")
    kd = O.k0_diamond()
    assert_bit_equal(O.dilate_mask5(x, kd), ndi.grey_dilation(x, footprint=kd.astype(bool), mode="constant", cval=-fmax), "diamond")


def test_generated_median_networks_are_reproducible_and_proven(tmp_path):
    """csrc/median_shared_nets3.h is exactly what tools/gen_median_3in.py generates (the generator re-proves every rewrite
    with the 0/1 principle on sorted inputs and cross-checks random floats while it runs), and the closing five-med3
    chain selects the 6th smallest of sorted 6 + sorted 5."""
    import importlib.util, os, shutil
    from conftest import ROOT
    src = os.path.join(ROOT, "depth_completion_mt_amd", "csrc")
    shutil.copy(os.path.join(ROOT, "tools", "gen_median_3in.py"), tmp_path / "gen_median_3in.py")
    shutil.copy(os.path.join(src, "median_shared_nets.h"), tmp_path / "median_shared_nets.h")
    spec = importlib.util.spec_from_file_location("gen3", str(tmp_path / "gen_median_3in.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    g.main()                                          # asserts inside: 0/1 proof + 20000 random merges per network + the final chain
    assert open(tmp_path / "median_shared_nets3.h").read() == open(os.path.join(src, "median_shared_nets3.h")).read()


def test_chain_with_definitional_median_matches_networks(golden_meta):
    """The chain oracle with dcmt_oracle_median5_simple (gather 25, select the 13th -- shares no comparator network with the HIP
    kernels) reproduces the committed full-size checksums, i.e. equals the network median bit for bit at both frame sizes."""
    for key, m in golden_meta["full"].items():
        if key.startswith("lc_"):
            continue
        dims, seed = key.split("_seed")
        rows, cols = (int(v) for v in dims.split("x"))
        x = synth.synth_frame(rows, cols, int(seed))
        with O.definitional_median():
            y = O.img_completion(x)
        assert sha(y) == m["out_sha256"], key


def test_oracle_under_address_and_ub_sanitizers():
    """the CPU oracle runs its whole test file under -fsanitize=address,undefined (oracle/Makefile,
    libdcmt_oracle_asan.so) in a child process that preloads the sanitizer runtimes; any report aborts the child."""
    import os, shutil, subprocess, sys
    from conftest import ROOT
    if os.environ.get("DCMT_ORACLE_TARGET"):
        pytest.skip("already inside the sanitizer run")
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    libs = [subprocess.run([gcc, f"-print-file-name={n}"], capture_output=True, text=True, check=True).stdout.strip()
            for n in ("libasan.so", "libubsan.so")]
    assert all(os.path.isabs(p) and os.path.exists(p) for p in libs), libs
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libdcmt_oracle_asan.so"], check=True, capture_output=True)
    env = dict(os.environ, DCMT_ORACLE_TARGET="libdcmt_oracle_asan.so", LD_PRELOAD=" ".join(libs),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "not definitional"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
