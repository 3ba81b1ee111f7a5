"""Parity tests proper: the HIP path, called through the C ABI, against the oracle and the
committed goldens.  Bit-exact for every stage (max/min/median/select are exact; the
Gaussian uses the same operation order without FMA as the oracle).  Against another OpenCV
build the Gaussian stage is a tolerance stage: |d| <= 1e-4."""
import ctypes
import hashlib

import numpy as np
import pytest

from conftest import assert_bit_equal
from depth_completion_mt_amd import _lib as L
from depth_completion_mt_amd import api, synth

pytestmark = pytest.mark.gpu

GAUSS_TOL = 1e-4     # stated float tolerance of the Gaussian stage vs. any conforming OpenCV


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0, 400, 1300, 16)
    yield c
    c.close()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def test_native_library_is_loaded():
    assert L.lib().dcmt_device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "libdcmt_hip.so" in f.read()


def test_per_stage_goldens(ctx, golden):
    x = golden["crop48x64_in"]
    for st in range(2, 12):
        got = ctx.complete(x, api.make_params(stop_after=st))
        assert_bit_equal(got, golden[f"crop48x64_stage{st}"], f"HIP stage {st}")
        if st >= 6:      # the fused streaming kernels on the same single frame
            got = ctx.complete(x, api.make_params(stop_after=st, force_fused=True))
            assert_bit_equal(got, golden[f"crop48x64_stage{st}"], f"HIP fused stage {st}")
    assert_bit_equal(ctx.complete(x, api.make_params(k0="diamond")), golden["crop48x64_diamond"], "diamond")
    assert_bit_equal(ctx.complete(x, api.make_params(blur_type="none")), golden["crop48x64_noblur"], "no blur")
    assert_bit_equal(ctx.complete(golden["odd33x70_in"]), golden["odd33x70_out"], "33x70")
    assert_bit_equal(ctx.complete(golden["odd33x70_in"], api.make_params(k0="diamond")), golden["odd33x70_diamond"], "33x70 diamond")


def test_adversarial_goldens(ctx, golden, golden_meta):
    for name, info in golden_meta["adversarial"].items():
        x = golden[f"adv_{name}_in"]
        for fused in (False, True):
            got = ctx.complete(x, api.make_params(max_fill_iters=8, force_fused=fused), allow_not_converged=True)
            assert_bit_equal(got, golden[f"adv_{name}_out"], f"adversarial {name} fused={fused}")
            iters, st = ctx.last_fill_iters(1)
            assert st == L.OK and iters[0] == info["fill_iters"], (name, iters, info)
            assert ctx.last_holes_after_extend(1)[0] == info["holes_after_extend"], name


def test_fill_loop_cap(ctx, golden, O):
    x = golden["adv_tall_gap_in"]
    want = O.img_completion(x, O.default_params(max_fill_iters=2))
    for fused in (False, True):
        got = ctx.complete(x, api.make_params(max_fill_iters=2, force_fused=fused), allow_not_converged=True)
        assert ctx.last_status == L.E_NOT_CONVERGED
        assert_bit_equal(got, want, f"capped loop fused={fused}")
    with pytest.raises(api.DcmtError):
        ctx.complete(x, api.make_params(max_fill_iters=2))


def test_stage_parity_full_size_vs_oracle(ctx, O, golden_meta):
    for rows, cols, seed in [(352, 1216, 0), (375, 1242, 1)]:
        x = synth.synth_frame(rows, cols, seed)
        for st in (4, 5, 6, 7, 9, 11):
            want = O.img_completion(x, O.default_params(stop_after=st))
            assert_bit_equal(ctx.complete(x, api.make_params(stop_after=st)), want, f"{rows}x{cols} stage {st}")
            if st >= 6:
                assert_bit_equal(ctx.complete(x, api.make_params(stop_after=st, force_fused=True)), want, f"{rows}x{cols} fused stage {st}")
        m = golden_meta["full"][f"{rows}x{cols}_seed{seed}"]
        assert sha(ctx.complete(x)) == m["out_sha256"]
        assert ctx.last_fill_iters(1)[0][0] == m["fill_iters"]
        assert ctx.last_holes_after_extend(1)[0] == m["holes_after_extend"]


def test_gaussian_stage_within_stated_tolerance_of_fp64(ctx):
    """The float stage against an fp64 evaluation of the same 5x5 binomial: the tolerance the
    north star asks to be written down."""
    x = synth.synth_frame(352, 1216, 5)
    med = ctx.complete(x, api.make_params(stop_after=L.STAGE_MEDIAN5)).astype(np.float64)
    blur = ctx.complete(x, api.make_params(stop_after=L.STAGE_BLUR)).astype(np.float64)
    k = np.array([1, 4, 6, 4, 1], np.float64) / 16
    p = np.pad(med, 2, mode="reflect")
    t = sum(k[i] * p[:, i:i + med.shape[1]] for i in range(5))
    g = sum(k[i] * t[i:i + med.shape[0], :] for i in range(5))
    want = np.where(med >= np.float32(0.1), g, med)
    assert np.abs(blur - want).max() <= GAUSS_TOL


def test_device_entry_point_batch(ctx, O):
    import torch
    frames = synth.synth_batch(6, 352, 1216, 40)
    frames[2] = 0                                   # an empty frame in the middle of the batch
    frames[4, :, 100:140] = 0                       # empty columns
    d = torch.from_numpy(frames).cuda()
    want = [O.img_completion(frames[i]) for i in range(frames.shape[0])]
    for fused in (True, False):                     # 6 frames: the default picks the staged kernels, force the streaming ones too
        out = ctx.complete_dev(d, params=api.make_params(force_fused=fused))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for i in range(frames.shape[0]):
            assert_bit_equal(got[i], want[i], f"device batch frame {i} fused={fused}")
        iters, st = ctx.last_fill_iters(6)
        assert st == L.OK and iters == [1] * 6
        assert (got[2] == 0).all()
    # a non-default stream
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out2 = ctx.complete_dev(d)
    s.synchronize()
    assert torch.equal(out, out2)
    # frames are independent: a permuted batch gives permuted results
    perm = [3, 0, 5, 1, 4, 2]
    out3 = ctx.complete_dev(d[perm].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(out3, out[perm])


def test_device_entry_point_speculative_loop(ctx, golden, O):
    """tall_gap needs 7 loop iterations: with 1 speculative application the device path reports
    non-convergence; with enough of them it matches the oracle."""
    import torch
    x = golden["adv_tall_gap_in"]
    d = torch.from_numpy(x[None].copy()).cuda()
    for fused in (False, True):
        ctx.complete_dev(d, params=api.make_params(spec_fill_iters=1, force_fused=fused))
        iters, st = ctx.last_fill_iters(1)
        assert st == L.E_NOT_CONVERGED and iters == [-1]
        out = ctx.complete_dev(d, params=api.make_params(spec_fill_iters=8, force_fused=fused))
        iters, st = ctx.last_fill_iters(1)
        assert st == L.OK and iters == [7]
        assert_bit_equal(out.cpu().numpy()[0], O.img_completion(x), f"speculative loop fused={fused}")


def test_strided_host_input_and_output_independence(ctx, O):
    big = np.zeros((60, 200), np.float32)
    big[:, :] = -7.0
    x = synth.synth_frame(60, 90, 9)
    view = big[:, 50:140]
    view[:] = x
    assert view.strides[0] == 800
    got = ctx.complete(view)
    assert_bit_equal(got, O.img_completion(x), "strided input")
    assert (big[:, :50] == -7).all() and (big[:, 140:] == -7).all()


def test_labeled_variant(ctx, golden, golden_meta, O):
    x, lab = golden["lc40x56_in"], golden["lc40x56_labels"]
    nl = golden_meta["lc40x56_n_labels"]
    rng = np.random.default_rng(3)
    lab2 = rng.integers(-1, 9, size=x.shape).astype(np.int32)
    for fused in (False, True):        # the general per-tile kernel and the per-label streaming kernels
        mk = lambda **kw: api.make_params(force_fused=fused, **kw)
        got4 = ctx.complete(x, mk(stop_after=L.STAGE_CLOSE5), labels=lab, n_labels=nl)
        assert_bit_equal(got4, golden["lc40x56_stage4"], f"LC stage 4 fused={fused}")
        assert_bit_equal(ctx.complete(x, mk(), labels=lab, n_labels=nl), golden["lc40x56_out"], f"LC out fused={fused}")
        assert_bit_equal(ctx.complete(x, mk(), labels=lab, n_labels=nl, use_superpixel=0), golden["lc40x56_out_nosp"], "LC nosp")
        # blur_type is ignored by the reference's LC function: the Gaussian always runs
        assert_bit_equal(ctx.complete(x, mk(blur_type="none"), labels=lab, n_labels=nl), golden["lc40x56_out"], "LC blur forced")
        assert_bit_equal(ctx.complete(x, mk(k0="diamond"), labels=lab, n_labels=nl),
                         O.interpolate_with_superpixels(x, lab, nl, O.default_params(k0="diamond")), f"LC diamond fused={fused}")
        # adversarial: every pixel its own neighbourhood of labels, some unlabeled
        assert_bit_equal(ctx.complete(x, mk(), labels=lab2, n_labels=8), O.interpolate_with_superpixels(x, lab2, 8), f"random labels fused={fused}")
        # more labels than the LDS bounding-box table holds (the direct global-atomic variant of the label pass)
        lab3 = rng.integers(0, 5000, size=x.shape).astype(np.int32)
        assert_bit_equal(ctx.complete(x, mk(), labels=lab3, n_labels=5000), O.interpolate_with_superpixels(x, lab3, 5000), f"5000 labels fused={fused}")
        # config 3 / 4 shapes
        for rows, cols, nt, seed in [(352, 1216, 1200, 0), (375, 1242, 100, 2)]:
            xf = synth.synth_frame(rows, cols, seed)
            labf, nlf = synth.synth_labels(rows, cols, nt, seed)
            got = ctx.complete(xf, mk(), labels=labf, n_labels=nlf)
            assert_bit_equal(got, O.interpolate_with_superpixels(xf, labf, nlf), f"LC {rows}x{cols} fused={fused}")
    m = golden_meta["full"]["lc_352x1216_seed0"]
    labf, nlf = synth.synth_labels(352, 1216, 1200, 0)
    assert sha(ctx.complete(synth.synth_frame(352, 1216, 0), labels=labf, n_labels=nlf)) == m["out_sha256"]


def test_labeled_device_batch(O):
    """Config 3 as a device-resident batch (the fast LC path by default: 16 >= 12 frames), labels differing per frame."""
    import torch
    n = 16
    frames = synth.synth_batch(n, 352, 1216, 900)
    labs = np.stack([synth.synth_labels(352, 1216, 1200, 900 + i)[0] for i in range(n)])
    nl = synth.synth_labels(352, 1216, 1200, 900)[1]
    labs[5][labs[5] == 17] = -1                      # a label that owns no pixel in one frame
    with api.Context(0, 352, 1216, n) as c:
        out = c.complete_dev(torch.from_numpy(frames).cuda(), d_labels=torch.from_numpy(labs).cuda(), n_labels=nl)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
    for i in (0, 5, 15):
        assert_bit_equal(got[i], O.interpolate_with_superpixels(frames[i], labs[i], nl), f"LC device batch frame {i}")


def test_errors(ctx):
    x = np.zeros((8, 8), np.float32)
    with pytest.raises(api.DcmtError) as e:
        ctx.complete(x, api.make_params(blur_type="bilateral"))
    assert e.value.status == L.E_UNSUPPORTED        # the reference's bilateral call throws too
    with pytest.raises(api.DcmtError) as e:
        ctx.complete(np.zeros((401, 8), np.float32))
    assert e.value.status == L.E_INVALID
    p = api.make_params()
    for i in range(25):
        p.k0[i] = 0
    with pytest.raises(api.DcmtError):
        ctx.complete(x, p)


def test_full_size_properties_large_batch():
    """BASELINE sizes (a device-resident batch of 352x1216 frames), size-independent checks:
    known answers for empty / single-pixel frames, per-frame independence, and the sha256 of
    known frames inside a big batch."""
    import torch
    n = 96
    with api.Context(0, 352, 1216, n) as c:
        frames = synth.synth_batch(8, 352, 1216, 0)
        big = np.concatenate([frames] * (n // 8))
        big[17] = 0
        big[33] = 0
        big[33, 200, 600] = 10.0
        d = torch.from_numpy(big).cuda()
        out = c.complete_dev(d)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        iters, st = c.last_fill_iters(n)
        assert st == L.OK and set(iters) == {1}
        import json, os
        from conftest import GOLDEN_DIR
        meta = json.load(open(os.path.join(GOLDEN_DIR, "meta.json")))["full"]
        assert sha(got[0]) == meta["352x1216_seed0"]["out_sha256"]
        assert sha(got[1]) == meta["352x1216_seed1"]["out_sha256"]
        assert sha(got[88]) == meta["352x1216_seed0"]["out_sha256"]
        for i in range(8, n):
            if i in (17, 33):
                continue
            assert np.array_equal(got[i], got[i % 8]), i          # same input, same output, anywhere in the batch
        assert (got[17] == 0).all()                               # K1
        assert set(np.unique(got[33]).tolist()) == {0.0, 0.625, 3.125, 6.875, 9.375, 10.0}   # K2
        assert (got >= 0).all() and got.max() <= 100.0


def test_fused_path_odd_sizes_and_presets(ctx, O):
    """The fused streaming kernels against the oracle on awkward geometries (strip / chunk tails,
    images narrower than a strip, the smallest sizes the fused path accepts) and both k0 presets;
    and the staged kernels forced on the same inputs."""
    rng = np.random.default_rng(11)
    for rows, cols in [(8, 8), (9, 200), (100, 8), (65, 129), (31, 47), (64, 48), (17, 1216), (352, 60), (128, 132)]:
        x = synth.synth_frame(rows, cols, rows * 1000 + cols)
        x[rng.integers(0, rows, 5), rng.integers(0, cols, 5)] = 30.0      # something valid even in tiny frames
        for k0 in ("as_compiled", "diamond"):
            want = O.img_completion(x, O.default_params(k0=k0))
            assert_bit_equal(ctx.complete(x, api.make_params(k0=k0, force_fused=True)), want, f"fused {rows}x{cols} {k0}")
            assert_bit_equal(ctx.complete(x, api.make_params(k0=k0, force_staged=True)), want, f"staged {rows}x{cols} {k0}")
    x = synth.synth_frame(352, 1216, 77)
    assert_bit_equal(ctx.complete(x, api.make_params(k0="diamond", force_fused=True)), O.img_completion(x, O.default_params(k0="diamond")), "diamond full size")
    assert_bit_equal(ctx.complete(x, api.make_params(blur_type="none", force_fused=True)), O.img_completion(x, O.default_params(blur="none")), "no blur full size")
    for st in (6, 7, 8, 9, 10):
        assert_bit_equal(ctx.complete(x, api.make_params(stop_after=st, force_fused=True)), O.img_completion(x, O.default_params(stop_after=st)), f"fused stage {st}")


def test_dense_and_adversarial_values_full_size(ctx, O):
    rng = np.random.default_rng(5)
    x = rng.uniform(0.0, 130.0, size=(352, 1216)).astype(np.float32)     # dense, incl. depths beyond max_depth
    x[rng.random(x.shape) < 0.3] = 0
    x[100:140, 300:420] = 0                                               # a 40 x 120 hole: needs the loop
    want, info = O.img_completion(x, return_info=True)
    for fused in (False, True):
        got = ctx.complete(x, api.make_params(force_fused=fused))
        assert_bit_equal(got, want, f"dense random fused={fused}")
        assert ctx.last_fill_iters(1)[0][0] == info["fill_iters"]


def test_dispatch_toggles_give_identical_results(O):
    """The environment toggles of the fused path (separate k_fill_s + k_post_s instead of k_fp_s, dword row
    loads instead of the LDS-DMA ring, plain instead of XCD-aware block mapping) only pick other kernels for
    the same arithmetic: every combination must reproduce the oracle.  Run in a child process each, because
    the toggles are read when a context is created."""
    import os, subprocess, sys, textwrap
    from conftest import ROOT
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {ROOT!r})
        import numpy as np
        from depth_completion_mt_amd import Context, make_params, synth
        from oracle import oracle as O
        frames = synth.synth_batch(16, 352, 1216, 500)
        frames[3, 120:170, 400:520] = 0            # needs the hole-closure loop: exercises the redo path
        with Context(0, 352, 1216, 16) as ctx:
            got = ctx.complete(frames, make_params())
        for i in (0, 3, 15):
            assert np.array_equal(got[i].view(np.uint32), O.img_completion(frames[i]).view(np.uint32)), i
        print("OK")
    """)
    for env in ({"DCMT_FUSE_FP": "0"}, {"DCMT_WIDE": "0"}, {"DCMT_XCD_MAP": "0"}, {"DCMT_FUSE_FP": "0", "DCMT_WIDE": "0", "DCMT_XCD_MAP": "0"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "OK" in r.stdout, (env, r.stdout[-500:], r.stderr[-1500:])


def test_label_stage_wave_modes_give_identical_results():
    """The label-masked stage runs one wave per label pair (two boxes side by side in one wave where they fit) or
    one wave per label; the host picks by label size, DCMT_LABEL_PAIRS forces either.  Both must reproduce the
    oracle on small (packable) and large (chunked) labels, for both structuring elements."""
    import os, subprocess, sys, textwrap
    from conftest import ROOT
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {ROOT!r})
        import numpy as np
        from depth_completion_mt_amd import Context, make_params, synth
        from oracle import oracle as O
        for rows, cols, nt in ((352, 1216, 1200), (375, 1242, 100), (200, 333, 40)):
            x = synth.synth_frame(rows, cols, 77)
            lab, nl = synth.synth_labels(rows, cols, nt, 77)
            lab[lab == 3] = -1                      # an empty label inside a pair
            with Context(0, rows, cols, 1) as ctx:
                for k0 in ("as_compiled", "diamond"):
                    got = ctx.complete(x, make_params(force_fused=True, k0=k0), labels=lab, n_labels=nl)
                    ref = O.interpolate_with_superpixels(x, lab, nl, O.default_params(k0=k0))
                    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (rows, cols, nt, k0)
        print("OK")
    """)
    for env in ({"DCMT_LABEL_PAIRS": "0"}, {"DCMT_LABEL_PAIRS": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "OK" in r.stdout, (env, r.stdout[-500:], r.stderr[-1500:])


def test_uint16_ingest(O):
    """The reference's ingest (main.cpp:75-82: uint16 PNG payload, convertTo(CV_32F, 1/256)) fused into the first
    kernel: same bits as converting on the host and calling the f32 entry point -- streaming path (16 frames) and
    staged path (2 frames)."""
    import torch
    for n in (16, 2):
        u16 = np.round(synth.synth_batch(n, 352, 1216, 300) * 256.0).astype(np.uint16)
        u16[0, 200, 300] = 65535                                  # 255.996 m: beyond max_depth
        as_f32 = (u16.astype(np.float32) * np.float32(1.0 / 256.0)).astype(np.float32)
        with api.Context(0, 352, 1216, n) as c:
            d = torch.from_numpy(u16.view(np.int16)).cuda()       # torch has no uint16 arithmetic; the bytes are what matters
            out = c.complete_u16_dev(d, 1.0 / 256.0)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
        for i in (0, n - 1):
            assert_bit_equal(got[i], O.img_completion(as_f32[i]), f"u16 ingest n={n} frame {i}")


NORM_TOL = 1e-5     # relative; only where the shift term is live (smin != 0): an FMA build of OpenCV rounds once, this path twice


def test_n1_normalize_fused_in_front_of_the_path(ctx, golden, golden_meta, O):
    """DCMT_FLAG_NORMALIZE = cv::normalize(NORM_MINMAX) + the path (SL/main_sl.cpp:370 -> img_completion,
    :523 -> interpolate_with_superpixels).  Bit-exact against the goldens and the oracle on every dispatch path."""
    x = golden["norm48x64_in"]
    for fused in (False, True):
        mk = lambda **kw: api.make_params(force_fused=fused, **kw)
        assert_bit_equal(ctx.complete(x, mk(normalize=(0, 100), stop_after=L.STAGE_NORMALIZE)), golden["norm48x64_n100"], "normalised frame")
        assert_bit_equal(ctx.complete(x, mk(normalize=(0, 100), stop_after=2)), golden["norm48x64_stage2_100"], "H2 of normalised frame")
        assert_bit_equal(ctx.complete(x, mk(normalize=(0, 100))), golden["norm48x64_out100"], f"chain (0,100) fused={fused}")
        assert_bit_equal(ctx.complete(x, mk(normalize=(100, 0))), golden["norm48x64_out100"], "argument order")
        assert_bit_equal(ctx.complete(x, mk(normalize=(0, 80), k0="diamond")), golden["norm48x64_out80_diamond"], f"(0,80) diamond fused={fused}")
        # labeled variant, as SL/main_sl.cpp:523-540 chains them
        xl, lab, nl = golden["lc40x56_in"], golden["lc40x56_labels"], golden_meta["lc40x56_n_labels"]
        assert_bit_equal(ctx.complete(xl, mk(normalize=(0, 80)), labels=lab, n_labels=nl), golden["norm_lc40x56_out80"], f"LC (0,80) fused={fused}")
        # smin != 0 (no empty pixel, a negative value): still the oracle's bits; the stated tolerance is for other OpenCV builds
        xd = golden["norm_dense40x56_in"]
        got = ctx.complete(xd, mk(normalize=(0, 80), stop_after=L.STAGE_NORMALIZE))
        assert_bit_equal(got, golden["norm_dense40x56_n80"], "normalise, live shift")
        ref64 = (xd.astype(np.float64) - xd.min()) * (80.0 / (float(xd.max()) - float(xd.min())))
        assert np.abs(got - ref64).max() <= NORM_TOL * 80.0
        assert_bit_equal(ctx.complete(xd, mk(normalize=(0, 80))), golden["norm_dense40x56_out80"], f"chain, live shift fused={fused}")
        flat = ctx.complete(np.full((8, 8), 2.0, np.float32), mk(normalize=(5, 80), stop_after=L.STAGE_NORMALIZE))
        assert_bit_equal(flat, golden["norm_flat_n"], "flat frame")
    # config 4 shape, frames with different extrema in one device batch (per-frame coefficients), both presets
    import torch
    n = 16
    frames = synth.synth_batch(n, 375, 1242, 700) * np.linspace(0.5, 1.5, n, dtype=np.float32)[:, None, None]
    with api.Context(0, 375, 1242, n) as c:
        for k0 in ("as_compiled", "diamond"):
            out = c.complete_dev(torch.from_numpy(frames).cuda(), params=api.make_params(normalize=(0, 80), k0=k0))
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            for i in (0, 7, 15):
                assert_bit_equal(got[i], O.img_completion(O.normalize_minmax(frames[i], 0, 80), O.default_params(k0=k0)), f"device batch frame {i} {k0}")
        labs = np.stack([synth.synth_labels(375, 1242, 100, 700 + i)[0] for i in range(n)])
        nl = synth.synth_labels(375, 1242, 100, 700)[1]
        out = c.complete_dev(torch.from_numpy(frames).cuda(), params=api.make_params(normalize=(0, 80)),
                             d_labels=torch.from_numpy(labs).cuda(), n_labels=nl)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for i in (0, 9):
            assert_bit_equal(got[i], O.interpolate_with_superpixels(O.normalize_minmax(frames[i], 0, 80), labs[i], nl), f"LC device batch frame {i}")
        # the two ingests are never combined by the reference: refused, not silently ignored
        with pytest.raises(api.DcmtError) as e:
            c.complete_u16_dev(torch.zeros((n, 375, 1242), dtype=torch.int16, device="cuda"), params=api.make_params(normalize=(0, 80)))
        assert e.value.status == L.E_UNSUPPORTED
    with pytest.raises(api.DcmtError) as e:       # stage 1 only exists with the flag
        ctx.complete(x, api.make_params(stop_after=L.STAGE_NORMALIZE))
    assert e.value.status == L.E_INVALID


def test_n2_lidar_projection_to_sparse_image(ctx, golden, O):
    """dcmt_project_points_dev = the loop that builds the path's input in the stereo-lidar executables
    (SL/main_sl.cpp:478-520): bit-exact against the goldens and the oracle, including the file-order rule for points
    that share a pixel, ragged batches and an empty sweep; then the whole :370-386 sequence on the device."""
    import torch
    pts, T, P = golden["proj_points"], golden["proj_T"], golden["proj_P"]
    off = torch.tensor([0, len(pts)], dtype=torch.int32, device="cuda")
    got = ctx.project_points_dev(torch.from_numpy(pts).cuda(), off, T, P, 48, 64)
    torch.cuda.synchronize()
    assert_bit_equal(got.cpu().numpy()[0], golden["proj_sparse48x64"], "projection 48x64")
    # ragged batch at config 4's size: sweeps of different lengths, one of them empty, many collisions
    sweeps = [synth.synth_points(n, 50 + i) for i, n in enumerate((120000, 0, 60000, 250000))]
    sweeps[2][:, :3] = sweeps[2][0, :3]                           # 60000 points marching through a few dozen pixels:
    sweeps[2][:, 0] += np.linspace(0, 5, len(sweeps[2]), dtype=np.float32)   # hundreds of writers per pixel, the last in file order stays
    allp = np.concatenate(sweeps)
    offs = np.cumsum([0] + [len(s) for s in sweeps]).astype(np.int32)
    T4, P4 = synth.KITTI_T_VELO_TO_CAM, synth.KITTI_P2
    with api.Context(0, 375, 1242, 4) as c:
        d_sparse = c.project_points_dev(torch.from_numpy(allp).cuda(), torch.from_numpy(offs).cuda(), T4, P4, 375, 1242)
        torch.cuda.synchronize()
        sp = d_sparse.cpu().numpy()
        for f, s in enumerate(sweeps):
            assert_bit_equal(sp[f], O.project_points(s, T4, P4, 375, 1242), f"sweep {f}")
        assert not sp[1].any() and 1 <= (sp[2] > 0).sum() < 200
        # project -> normalize(0, 100) -> img_completion, all on the device (main_sl.cpp:320-386)
        dense = c.complete_dev(d_sparse, params=api.make_params(normalize=(0, 100), force_fused=True))
        torch.cuda.synchronize()
        dn = dense.cpu().numpy()
        for f in (0, 3):
            assert_bit_equal(dn[f], O.img_completion(O.normalize_minmax(sp[f], 0, 100)), f"chain on sweep {f}")
        assert not dn[1].any()                                       # empty sweep -> empty image -> all zero (K1)
    with pytest.raises(api.DcmtError) as e:                          # image larger than the context was created for
        ctx.project_points_dev(torch.from_numpy(pts).cuda(), off, T, P, 4000, 64)
    assert e.value.status == L.E_INVALID


def test_n3_slic_labels_on_the_device(golden, golden_meta, O):
    """dcmt_slic_labels_dev = Slic::generate_superpixels (LC/slic.cpp:101-182): labels equal to the oracle's pixel for
    pixel and the f64 centres bit for bit -- on the golden (which has a centre that dies), at both executables'
    settings (1200 superpixels / nc 50 at 352x1216, 100 / nc 40 at 375x1242), batched with different images; then
    SLIC -> interpolate_with_superpixels on the device, the two steps main_lc.cpp:200 and :220 chain."""
    import torch
    img = golden["slic_lab96x160"]
    with api.Context(0, 96, 160, 1) as c:
        lab, n, cent = c.slic_labels_dev(torch.from_numpy(img).cuda(), 12, 40, return_centers=True)
        torch.cuda.synchronize()
        assert n == golden_meta["slic96x160_n"]
        assert np.array_equal(lab.cpu().numpy()[0], golden["slic_labels96x160"])
        got, want = cent.cpu().numpy()[0], golden["slic_centers96x160"]
        dead = np.isnan(want[:, 3])
        assert np.array_equal(np.isnan(got[:, 3]), dead)
        assert np.array_equal(got[~dead].view(np.uint64), want[~dead].view(np.uint64))
        x = synth.synth_frame(96, 160, 5)
        dense = c.complete_dev(torch.from_numpy(x).cuda()[None], d_labels=lab, n_labels=n, params=api.make_params(force_fused=True))
        torch.cuda.synchronize()
        assert_bit_equal(dense.cpu().numpy()[0], golden["slic_chain96x160"], "SLIC -> interpolate_with_superpixels")
    # tiny steps: a tile then spans more cells than are staged in LDS and its pixels walk all centres from global
    # memory (the same walk a cell-list overflow falls back to); steps that do not divide the tile size
    for step, nc in ((6, 40), (7, 10), (11, 25)):
        img = np.ascontiguousarray(synth.synth_lab(75, 131, 40 + step))
        with api.Context(0, 75, 131, 1) as c:
            lab, n, cent = c.slic_labels_dev(torch.from_numpy(img).cuda(), step, nc, return_centers=True)
            torch.cuda.synchronize()
            wl, wn, wc = O.slic(img, step, nc, return_centers=True)
            assert wn == n and np.array_equal(lab.cpu().numpy()[0], wl), (step, int((lab.cpu().numpy()[0] != wl).sum()))
            ok = ~np.isnan(wc[:, 3])
            assert np.array_equal(cent.cpu().numpy()[0][ok].view(np.uint64), wc[ok].view(np.uint64))
    for rows, cols, nsp, nc in ((352, 1216, 1200, 50), (375, 1242, 100, 40)):
        step = int(np.sqrt(rows * cols / nsp))                        # the callers' double step, truncated at the call
        imgs = np.ascontiguousarray(np.stack([synth.synth_lab(rows, cols, 20 + i) for i in range(3)]))
        with api.Context(0, rows, cols, 3) as c:
            lab, n, cent = c.slic_labels_dev(torch.from_numpy(imgs).cuda(), step, nc, return_centers=True)
            torch.cuda.synchronize()
            got_l, got_c = lab.cpu().numpy(), cent.cpu().numpy()
            for f in range(3):
                wl, wn, wc = O.slic(imgs[f], step, nc, return_centers=True)
                assert wn == n
                assert np.array_equal(got_l[f], wl), (rows, cols, f, int((got_l[f] != wl).sum()))
                ok = ~np.isnan(wc[:, 3])
                assert np.array_equal(got_c[f][ok].view(np.uint64), wc[ok].view(np.uint64))
            with pytest.raises(api.DcmtError) as e:
                c.slic_labels_dev(torch.from_numpy(imgs).cuda(), 5, nc)        # the reference's 3x3 probe would leave the image
            assert e.value.status == L.E_INVALID


def test_n3_slic_cell_list_overflow_falls_back_to_the_full_walk():
    """With cells twice / three times the step, four or nine centres share a cell, the 4-entry lists overflow and
    the frame walks all centres; with ample cells (scale 1) nothing overflows.  Same labels either way."""
    import os, subprocess, sys, textwrap
    from conftest import ROOT
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {ROOT!r})
        import numpy as np, torch
        from depth_completion_mt_amd import Context, synth
        from oracle import oracle as O
        img = np.ascontiguousarray(synth.synth_lab(120, 200, 3))
        with Context(0, 120, 200, 1) as c:
            lab, n = c.slic_labels_dev(torch.from_numpy(img).cuda(), 12, 40)
            torch.cuda.synchronize()
        wl, wn = O.slic(img, 12, 40)
        assert n == wn and np.array_equal(lab.cpu().numpy()[0], wl)
        print("OK")
    """)
    for scale in ("1", "2", "3"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DCMT_SLIC_CELL_SCALE=scale), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "OK" in r.stdout, (scale, r.stdout[-500:], r.stderr[-1500:])


def test_n4_stereo_refinement_on_the_device(golden, O):
    """dcmt_stereo_refine_dev = the refinement DC_stereo_lidar runs on the path's output (SL/main_sl.cpp:715-885, driven
    from :1165-1246): bit-exact against the goldens and the oracle, batched at config 4's size, and chained behind the
    path on the device (complete -> refine)."""
    import torch
    l, r, g = golden["stereo_left48x64"], golden["stereo_right48x64"], golden["stereo_guess48x64"]
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    with api.Context(0, 48, 64, 1) as c:
        got = c.stereo_refine_dev(cu(g), cu(l), cu(r), focal=60.0)
        pre = c.stereo_refine_dev(cu(g), cu(l), cu(r), focal=60.0, iterations=0)
        torch.cuda.synchronize()
        assert_bit_equal(got.cpu().numpy(), golden["stereo_refined48x64"], "refined 48x64")
        assert_bit_equal(pre.cpu().numpy(), golden["stereo_roundtrip48x64"], "round trip 48x64")
    rows, cols, n = 375, 1242, 5
    trip = [synth.synth_stereo(rows, cols, 30 + i) for i in range(n)]
    L_, R_, G_ = (np.stack([t[k] for t in trip]) for k in range(3))
    G_[1] = 0                                                         # a frame without any depth stays all zero
    with api.Context(0, rows, cols, n) as c:
        got = c.stereo_refine_dev(cu(G_), cu(L_), cu(R_))
        # the path in front: sparse (4 % of the guess) -> img_completion -> refinement, all on the device
        sparse = np.where(np.random.default_rng(1).random(G_.shape) < 0.04, G_, 0).astype(np.float32)
        dense = c.complete_dev(cu(sparse))
        both = c.stereo_refine_dev(dense, cu(L_), cu(R_))
        torch.cuda.synchronize()
        got, dn, both = got.cpu().numpy(), dense.cpu().numpy(), both.cpu().numpy()
    for f in range(n):
        assert_bit_equal(got[f], O.stereo_refine(G_[f], L_[f], R_[f]), f"refined frame {f}")
        assert_bit_equal(both[f], O.stereo_refine(dn[f], L_[f], R_[f]), f"complete -> refine frame {f}")
    assert not got[1].any()
