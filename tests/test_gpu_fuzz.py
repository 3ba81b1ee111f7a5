"""Differential fuzz of the HIP path against the oracle: random image sizes, densities, value ranges, structuring
elements, stages, batch sizes (both dispatch paths), label planes, the normalise flag -- every case seeded, every
result compared bit for bit (through the C ABI, like all parity tests)."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from depth_completion_mt_amd import _lib as L
from depth_completion_mt_amd import api

pytestmark = pytest.mark.gpu


def _frame(g, rows, cols):
    kind = g.integers(0, 5)
    density = (0.0, 0.01, 0.05, 0.3, 1.0)[g.integers(0, 5)]
    x = np.where(g.random((rows, cols)) < density, g.uniform(0.2, 85.0, (rows, cols)), 0.0).astype(np.float32)
    if kind == 1:                                   # values around the threshold and around max_depth
        m = g.random((rows, cols)) < 0.02
        x[m] = g.choice(np.array([0.05, 0.1, 0.099999994, 99.9, 99.95, 100.0, 100.5, 130.0], np.float32), int(m.sum()))
    elif kind == 2:                                 # empty columns / rows, a filled block
        x[:, g.integers(0, cols):] = 0
        x[: g.integers(0, rows)] = 0
    elif kind == 3 and rows > 40:                   # a tall gap: the hole-closure loop has to run
        x[8:-8] = 0
    return x


def _labels(g, rows, cols):
    kind = g.integers(0, 3)
    if kind == 0:                                   # jittered blocks of random size
        bh, bw = int(g.integers(3, 40)), int(g.integers(3, 70))
        lab = (np.arange(rows)[:, None] // bh) * ((cols + bw - 1) // bw) + np.arange(cols)[None, :] // bw
        lab = np.roll(lab, int(g.integers(0, 5)), axis=1)
    elif kind == 1:                                 # salt and pepper: every pixel its own neighbourhood of labels
        lab = g.integers(0, 9, (rows, cols))
    else:                                           # two big labels and a stripe
        lab = (np.arange(cols)[None, :] > cols // 2) + np.zeros((rows, 1), int)
        lab[rows // 3: rows // 3 + 2] = 2
    lab = lab.astype(np.int32)
    n = int(lab.max()) + 1
    lab[g.random((rows, cols)) < 0.03] = -1         # unlabeled pixels
    if g.random() < 0.5:
        lab[lab == n - 1] = n + 3                   # labels beyond n_labels are ignored like -1
    return lab, n


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_against_oracle(seed, monkeypatch):
    from oracle import oracle as O
    g = np.random.Generator(np.random.PCG64(1000 + seed))
    gb = np.random.Generator(np.random.PCG64(7000 + seed))       # row bands of k_fp_s (read by dcmt_create): 0 = by batch size
    for case in range(10):
        monkeypatch.setenv("DCMT_FBANDS", str(int(gb.integers(0, 7))))
        rows, cols = int(g.integers(8, 180)), int(g.integers(8, 400))
        batch = int(g.choice([1, 2, 13]))
        k0 = ("as_compiled", "diamond")[g.integers(0, 2)]
        blur = ("gaussian", "none")[g.integers(0, 2)]
        stop = int(g.choice([11, 11, 11, 9, 7, 6, 4, 2]))
        norm = (None, (0, 80), (0, 100))[g.integers(0, 3)]
        labeled = g.random() < 0.4
        frames = np.stack([_frame(g, rows, cols) for _ in range(batch)])
        what = f"seed {seed} case {case}: {rows}x{cols} b{batch} {k0} {blur} stop{stop} norm{norm} labeled{labeled}"
        kw = dict(k0=k0, blur_type=blur, stop_after=stop, max_fill_iters=6, force_fused=bool(g.integers(0, 2)))
        if norm is not None:
            kw["normalize"] = norm
        op = O.default_params(k0=k0, blur=blur, stop_after=stop, max_fill_iters=6)
        with api.Context(0, rows, cols, batch) as c:
            if labeled:
                if stop in (2, 6, 7, 9):            # per-stage dumps of the labeled variant: stage 4 and the whole chain only
                    kw["stop_after"] = 11; op.stop_after = 11
                lab, n = _labels(g, rows, cols)
                got = c.complete(frames, api.make_params(**kw), labels=np.broadcast_to(lab, frames.shape), n_labels=n, allow_not_converged=True)
            else:
                got = c.complete(frames, api.make_params(**kw), allow_not_converged=True)
        for f in range(batch):
            src = frames[f] if norm is None else O.normalize_minmax(frames[f], *norm)
            ref = O.interpolate_with_superpixels(src, lab, n, op) if labeled else O.img_completion(src, op)
            assert_bit_equal(got[f], ref, what + f" frame {f}")


def _grid_frame(g, rows, cols):
    """depths that are multiples of 1/256 m (a uint16 payload / 256), with the values where the 16-bit codes have their edges"""
    density = (0.0, 0.01, 0.05, 0.3, 1.0)[g.integers(0, 5)]
    k = np.where(g.random((rows, cols)) < density, g.integers(52, 22000, (rows, cols)), 0)
    kind = g.integers(0, 4)
    if kind == 1:       # around the threshold (25 / 26), around max_depth (25600), the largest payload (65535)
        m = g.random((rows, cols)) < 0.03
        k[m] = g.choice(np.array([1, 25, 26, 27, 25599, 25600, 25601, 30719, 30720, 40000, 65535]), int(m.sum()))
    elif kind == 2:
        k[:, g.integers(0, cols):] = 0
        k[: g.integers(0, rows)] = 0
    elif kind == 3 and rows > 40:
        k[8:-8] = 0
    return (k.astype(np.float32) / np.float32(256.0)).astype(np.float32)


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_16_bit_codes_against_oracle(seed, monkeypatch):
    """The 16-bit form of X6 (k_pre_p<Q16OUT> -> k_fp_q, by default only on batches of more than about six hundred frames) forced onto
    small random batches: frames on the 1/256 m grid with values at every edge of the code range, some batches with a frame off
    the grid (the gated f32 rerun), both structuring elements, with and without the blur, plain and label-masked, through the
    host entry point (synchronised hole-closure loop) and the device entry point."""
    import torch
    from oracle import oracle as O
    monkeypatch.setenv("DCMT_Q16_MIN_WAVES", "0")
    g = np.random.Generator(np.random.PCG64(7000 + seed))
    for case in range(6):
        rows, cols = int(g.integers(8, 180)), 2 * int(g.integers(4, 200))
        batch = int(g.choice([8, 9, 16]))
        k0 = ("as_compiled", "diamond")[g.integers(0, 2)]
        blur = ("gaussian", "none")[g.integers(0, 2)]
        labeled = g.random() < 0.3
        frames = np.stack([_grid_frame(g, rows, cols) for _ in range(batch)])
        if g.random() < 0.3:
            f = int(g.integers(0, batch))
            frames[f] = (frames[f] * np.float32(1.003)).astype(np.float32)      # one frame off the grid
        what = f"seed {seed} case {case}: {rows}x{cols} b{batch} {k0} {blur} labeled{labeled}"
        kw = dict(k0=k0, blur_type=blur, max_fill_iters=6)
        op = O.default_params(k0=k0, blur=blur, max_fill_iters=6)
        with api.Context(0, rows, cols, batch) as c:
            if labeled:
                lab, n = _labels(g, rows, cols)
                got = c.complete(frames, api.make_params(**kw), labels=np.broadcast_to(lab, frames.shape), n_labels=n, allow_not_converged=True)
                dev = None
            else:
                got = c.complete(frames, api.make_params(**kw), allow_not_converged=True)
                dev = c.complete_dev(torch.from_numpy(frames).cuda(), None, api.make_params(spec_fill_iters=6, **kw))
                torch.cuda.synchronize()
                dev = dev.cpu().numpy()
        for f in range(batch):
            ref = O.interpolate_with_superpixels(frames[f], lab, n, op) if labeled else O.img_completion(frames[f], op)
            assert_bit_equal(got[f], ref, what + f" frame {f} (host entry point)")
            if dev is not None:
                assert_bit_equal(dev[f], ref, what + f" frame {f} (device entry point)")


def test_mixed_calls_on_one_context_keep_no_state():
    """One long-lived context, forty calls of every kind in random order and random sizes (host and device entry points, labels,
    normalise, uint16 ingest, projection, SLIC, stereo refinement): scratch reuse, table regrowth and leftovers of an earlier
    call must never leak into a later result."""
    import torch
    from oracle import oracle as O
    from depth_completion_mt_amd import synth
    g = np.random.Generator(np.random.PCG64(77))
    with api.Context(0, 160, 320, 13) as c:
        for call in range(40):
            kind = int(g.integers(0, 7))
            rows, cols = int(g.integers(16, 160)), int(g.integers(16, 320))
            what = f"call {call} kind {kind} {rows}x{cols}"
            if kind == 0:                                            # host entry, small batch (staged path)
                b = int(g.integers(1, 4))
                x = np.stack([_frame(g, rows, cols) for _ in range(b)])
                got = c.complete(x, api.make_params(max_fill_iters=6), allow_not_converged=True)
                for f in range(b):
                    assert_bit_equal(got[f], O.img_completion(x[f], O.default_params(max_fill_iters=6)), what)
            elif kind == 1:                                          # device entry, batch 13 (streaming path), normalise on/off
                x = np.stack([_frame(g, rows, cols) for _ in range(13)])
                norm = (None, (0, 80))[g.integers(0, 2)]
                out = c.complete_dev(torch.from_numpy(x).cuda(), params=api.make_params(normalize=norm, spec_fill_iters=5, max_fill_iters=6))
                torch.cuda.synchronize()
                got = out.cpu().numpy()
                for f in (0, 12):
                    src = x[f] if norm is None else O.normalize_minmax(x[f], *norm)
                    assert_bit_equal(got[f], O.img_completion(src, O.default_params(max_fill_iters=6)), what)
            elif kind == 2:                                          # labeled, host
                x = _frame(g, rows, cols)
                lab, n = _labels(g, rows, cols)
                got = c.complete(x, api.make_params(max_fill_iters=6, force_fused=bool(g.integers(0, 2))), labels=lab, n_labels=n, allow_not_converged=True)
                assert_bit_equal(got, O.interpolate_with_superpixels(x, lab, n, O.default_params(max_fill_iters=6)), what)
            elif kind == 3:                                          # uint16 ingest, batch 13
                u16 = (np.stack([_frame(g, rows, cols) for _ in range(13)]) * 256).astype(np.uint16)
                out = c.complete_u16_dev(torch.from_numpy(u16.view(np.int16)).cuda(), 1.0 / 256.0, params=api.make_params(spec_fill_iters=5, max_fill_iters=6))
                torch.cuda.synchronize()
                f32 = (u16.astype(np.float32) * np.float32(1.0 / 256.0)).astype(np.float32)
                assert_bit_equal(out.cpu().numpy()[5], O.img_completion(f32[5], O.default_params(max_fill_iters=6)), what)
            elif kind == 4:                                          # projection
                pts = synth.synth_points(int(g.integers(0, 5000)), int(g.integers(0, 1000)))
                P = np.array([[60.0, 0, cols / 2, 3.0], [0, 60.0, rows / 2, 0.01], [0, 0, 1, 0.002]], np.float32)
                off = torch.tensor([0, len(pts)], dtype=torch.int32, device="cuda")
                out = c.project_points_dev(torch.from_numpy(pts).cuda(), off, synth.KITTI_T_VELO_TO_CAM, P, rows, cols)
                torch.cuda.synchronize()
                assert_bit_equal(out.cpu().numpy()[0], O.project_points(pts, synth.KITTI_T_VELO_TO_CAM, P, rows, cols), what)
            elif kind == 5:                                          # SLIC
                step = int(g.integers(6, 30))
                if rows <= 2 * step or cols <= 2 * step:
                    continue
                img = synth.synth_lab(rows, cols, int(g.integers(0, 1000)))
                lab, n = c.slic_labels_dev(torch.from_numpy(img).cuda(), step, int(g.integers(5, 60)) if False else 40)
                torch.cuda.synchronize()
                wl, wn = O.slic(img, step, 40)
                assert n == wn and np.array_equal(lab.cpu().numpy()[0], wl), what
            else:                                                    # stereo refinement
                l, r, d = synth.synth_stereo(rows, cols, int(g.integers(0, 1000)), focal=80.0)
                out = c.stereo_refine_dev(torch.from_numpy(d).cuda(), torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), focal=80.0)
                torch.cuda.synchronize()
                assert_bit_equal(out.cpu().numpy(), O.stereo_refine(d, l, r, focal=80.0), what)


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_rows_around_the_path_against_oracle(seed):
    """N2 projection, N3 SLIC, N4 stereo refinement on random sizes, parameters and contents (noise, flat patches with exact
    ties, few colours, colour weight 1, perturbed calibrations, empty sweeps, disparities that leave the image, 0..6 sweeps)."""
    import torch
    from depth_completion_mt_amd import synth
    from oracle import oracle as O
    rng = np.random.default_rng(1000 + seed)
    rows, cols = int(rng.integers(24, 200)), int(rng.integers(24, 300))
    step = int(rng.integers(6, min(rows, cols) // 2))
    nc = int(rng.choice([1, 5, 20, 40, 50, 200]))
    kind = seed % 4
    if kind == 0:
        img = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    elif kind == 1:
        img = np.ascontiguousarray(synth.synth_lab(rows, cols, seed))
    elif kind == 2:
        img = np.zeros((rows, cols, 3), np.uint8)
        for _ in range(6):
            y, x = int(rng.integers(0, rows)), int(rng.integers(0, cols))
            img[y:y + int(rng.integers(4, 60)), x:x + int(rng.integers(4, 90))] = rng.integers(0, 256, 3)
    else:
        img = (rng.integers(0, 4, (rows, cols, 3)) * 80).astype(np.uint8)
    with api.Context(0, rows, cols, 2) as c:
        lab, n, cent = c.slic_labels_dev(torch.from_numpy(img).cuda(), step, nc, return_centers=True)
        torch.cuda.synchronize()
        wl, wn, wc = O.slic(img, step, nc, return_centers=True)
        assert wn == n and np.array_equal(lab.cpu().numpy()[0], wl), ("SLIC", rows, cols, step, nc, kind, int((lab.cpu().numpy()[0] != wl).sum()))
        ok = ~np.isnan(wc[:, 3])
        assert np.array_equal(cent.cpu().numpy()[0][ok].view(np.uint64), wc[ok].view(np.uint64)), ("SLIC centres", rows, cols, step, nc)
        npts = int(rng.integers(0, 40000))
        pts = synth.synth_points(npts, seed) if npts else np.zeros((0, 4), np.float32)
        T = (synth.KITTI_T_VELO_TO_CAM + rng.normal(0, 0.01, synth.KITTI_T_VELO_TO_CAM.shape)).astype(np.float32)
        P = synth.KITTI_P2.copy().astype(np.float32)
        P[0, 2], P[1, 2] = cols / 2, rows / 2
        P[0, 0] = P[1, 1] = float(rng.uniform(50, 400))
        off = torch.tensor([0, npts, npts], dtype=torch.int32, device="cuda")
        dp = torch.from_numpy(pts).cuda() if npts else torch.zeros((0, 4), dtype=torch.float32, device="cuda")
        sp = c.project_points_dev(dp, off, T, P, rows, cols).cpu().numpy()
        assert_bit_equal(sp[0], O.project_points(pts, T, P, rows, cols), f"projection {rows}x{cols}, {npts} points")
        assert not sp[1].any()
        l = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        r = np.ascontiguousarray(np.roll(l, -int(rng.integers(0, 9)), axis=1)) if seed % 2 else rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        depth = rng.uniform(0.0, 120.0, (rows, cols)).astype(np.float32)
        depth[rng.random((rows, cols)) < 0.1] = 0.0
        if seed % 5 == 0:
            depth[:, : cols // 3] = rng.uniform(0.01, 0.5, (rows, cols // 3))
        kw = dict(baseline=float(rng.uniform(0.1, 1.0)), focal=float(rng.uniform(50, 1200)), damp=float(rng.choice([1.0, 50.0, 500.0])),
                  max_depth=float(rng.choice([50.0, 100.0])))
        it = int(rng.integers(0, 7))
        cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()[None]
        got = c.stereo_refine_dev(cu(depth), cu(l), cu(r), iterations=it, **kw).cpu().numpy()[0]
        assert_bit_equal(got, O.stereo_refine(depth, l, r, iterations=it, **kw), f"stereo {rows}x{cols}, {it} sweeps, {kw}")
