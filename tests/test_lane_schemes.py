"""Index arithmetic of the kernels' lane schemes, restated in numpy and checked against the definition on the CPU (the
kernels themselves are checked bit for bit against the oracle by the GPU tests).

k_fp_p (csrc/dcmt_kernels_fp_pair.h): the 31-wide horizontal maximum with two columns per lane -- exclusive prefix / suffix
scans inside 16-lane DPP rows, the halo columns wrapped onto lanes 0..7 / 56..63 of a second register, two ds_bpermute
addresses -- and the sorted 5-windows of a lane's two columns from one shared sort of four values."""
import itertools

import numpy as np

NEG = -3.0e38


def _rows(x, fn):
    y = np.empty(64)
    for r in range(0, 64, 16):
        y[r:r + 16] = fn(x[r:r + 16])
    return y


def _prefix(x): return _rows(x, np.maximum.accumulate)
def _suffix(x): return _rows(x, lambda v: np.maximum.accumulate(v[::-1])[::-1])
def _shr1(x): return _rows(x, lambda v: np.concatenate(([NEG], v[:-1])))     # row_shr:1, the lane without a source keeps -FLT_MAX
def _shl1(x): return _rows(x, lambda v: np.concatenate((v[1:], [NEG])))
def _ror8(x): return _rows(x, lambda v: np.roll(v, -8))


def test_fp_pair_horizontal_31_max_matches_the_definition():
    rng = np.random.default_rng(5)
    lane = np.arange(64)
    for trial in range(300):
        cols = rng.integers(-50, 50, size=160).astype(float)      # virtual columns gx0-16 .. gx0+143
        if trial % 3 == 0:
            cols[rng.integers(0, 160, size=150)] = -50            # long runs of equal values
        A = cols[16:144]
        E, O = A[0::2].copy(), A[1::2].copy()
        B = rng.integers(-50, 50, size=64).astype(float)          # dead lanes hold anything
        k = np.arange(8)
        B[k] = cols[144 + 2 * k]; B[8 + k] = cols[144 + 2 * k + 1]          # right halo: even columns, odd columns
        B[56 + k] = cols[2 * k]; B[48 + k] = cols[2 * k + 1]                # left halo (from gx0-16): even, odd
        bo = _ror8(B)
        pa, sa = _prefix(np.maximum(E, O)), _suffix(np.maximum(E, O))
        pb, sb = _prefix(np.maximum(B, bo)), _suffix(np.maximum(B, bo))
        pxa, sxa, pxb, sxb = _shr1(pa), _shl1(sa), _shr1(pb), _shl1(sb)
        sx = np.where(lane >= 56, sxb, sxa); so = np.maximum(np.where(lane >= 56, bo, O), sx)
        px = np.where(lane < 8, pxb, pxa); pe = np.maximum(np.where(lane < 8, B, E), px)
        m8, p8 = (lane - 8) % 64, (lane + 8) % 64
        dE, dO = np.maximum(so[m8], px[p8]), np.maximum(sx[m8], pe[p8])
        for l in range(64):
            c = 16 + 2 * l
            assert dE[l] == cols[c - 15:c + 16].max(), (trial, l)
            assert dO[l] == cols[c - 14:c + 17].max(), (trial, l)


def test_fp_pair_shared_sort_matches_sorted():
    def med3(a, b, c): return sorted((a, b, c))[1]
    for v in itertools.product(range(5), repeat=6):               # E[l-1], O[l-1], E, O, E[l+1], O[l+1] with ties
        el, a, e, o, d, orr = v
        lo1, hi1, lo2, hi2 = min(a, e), max(a, e), min(d, o), max(d, o)
        y = min(hi1, hi2)
        s = [min(lo1, lo2), med3(lo1, lo2, y), max(lo1, lo2, y), max(hi1, hi2)]
        assert s == sorted((a, e, o, d))
        for x, want in ((el, sorted((el, a, e, o, d))), (orr, sorted((a, e, o, d, orr)))):
            got = [min(x, s[0]), med3(x, s[0], s[1]), med3(x, s[1], s[2]), med3(x, s[2], s[3]), max(x, s[3])]
            assert got == want
