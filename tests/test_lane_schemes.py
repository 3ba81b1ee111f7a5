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


def test_sort5_from_three_input_minima_maxima_and_xors():
    """q_sort5 (csrc/dcmt_kernels_fp_q16.h): sort3 of three values as min3 / max3 / XOR, sort2 of the other two, the merge of 3 + 2 by
    rank, the middle one as the XOR of all five and the other four.  Exhaustive on five values of five levels (every tie pattern)."""
    def sort5x(v):
        v0, v1, v2, v3, v4 = v
        a, c, t = min(v0, v1, v2), max(v0, v1, v2), v0 ^ v1 ^ v2
        b = t ^ a ^ c
        d, e = min(v3, v4), max(v3, v4)
        s0, s4 = min(a, d), max(c, e)
        s1, s3 = min(max(a, d), b, e), max(min(c, e), b, d)
        return [s0, s1, (t ^ v3 ^ v4) ^ (s0 ^ s1 ^ s3) ^ s4, s3, s4]
    for v in itertools.product((3, 1029, 7000, 7001, 31743), repeat=5):
        assert sort5x(v) == sorted(v), v


def test_k_fp_h_stage_windows_cover_31_columns():
    """k_fp_h (csrc/dcmt_kernels_fp_h16.h): W3 -> W7 -> W19 -> out through column shifts (1, 2), (2, 4), (6, 12), (-15, -3): the closing
    maximum is the 31-wide window centred on the column, and the values an output of the strip's 128 columns depends on lie inside the
    160 columns the stage arrays hold."""
    rng = np.random.default_rng(9)
    n = 160 + 32
    for _ in range(50):
        v0 = rng.integers(0, 1000, size=n)
        w3 = np.array([v0[c:c + 3].max() for c in range(n - 2)])
        w7 = np.array([max(w3[c], w3[c + 2], w3[c + 4]) for c in range(len(w3) - 4)])
        w19 = np.array([max(w7[c], w7[c + 6], w7[c + 12]) for c in range(len(w7) - 12)])
        for c in range(16, 16 + 128):                       # array slot of the strip's columns
            assert max(w19[c - 15], w19[c - 3]) == v0[c - 15:c + 16].max()
            assert c + 15 < 160 and c - 15 >= 1
