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
def test_fuzz_against_oracle(seed):
    from oracle import oracle as O
    g = np.random.Generator(np.random.PCG64(1000 + seed))
    for case in range(10):
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
