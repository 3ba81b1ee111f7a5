"""Generates median_shared_nets.h in the current directory (checked in as depth_completion_mt_amd/csrc/median_shared_nets.h
for the kernels and, with its own header comment, as oracle/median_nets.h for the CPU oracle): the comparator networks of the time-shared exact
5x5 median used by the streaming post kernel, and verifies them.

Scheme (per lane = per image column, rows arrive one per step):
  S(v)   = the 5 horizontal neighbours of row v, sorted            (SORT5, every row)
  P(m)   = merge of S(2m+1), S(2m+2)  -> 10 sorted                 (MERGE55, every 2nd row)
  C(m)   = ranks 8..13 (1-based) of the 20 values P(m-1) u P(m)    (MID20, every 2nd row)
           = the only values of the 4-row core rows 2m-1..2m+2 that can be the median of a
             25-window containing the core: a core value of core-rank r has window rank
             r..r+5, and the median has rank 13
  median(window 2m)   = 6th smallest of C(m) u S(2m-2)             (8 ops, closed form)
  median(window 2m+1) = 6th smallest of C(m) u S(2m+3)
Networks are built from Batcher odd-even merges of padded lists, pruned by liveness of the
needed outputs; each is verified exhaustively with the 0/1 principle on sorted inputs, and
the whole scheme is checked against numpy's median on random data.
"""
import itertools
import random

def oddeven_merge_net(n):
    """Batcher odd-even merge network for two sorted halves of a power-of-two n."""
    net = []
    def merge(lo, n, r):
        m = r * 2
        if m < n:
            merge(lo, n, m)
            merge(lo + r, n, m)
            for i in range(lo + r, lo + n - r, m):
                net.append((i, i + r))
        else:
            net.append((lo, lo + r))
    merge(0, n, 1)
    return net

def merge_net(na, nb, want):
    """Network merging sorted A (wires 0..na-1) and sorted B (wires na..na+nb-1); returns
    (ops, outmap): ops on wire ids, outmap[k] = wire holding the k-th smallest, only for k in want.
    Built on padded power-of-two halves; pads are +inf and never move."""
    h = 1
    while h < max(na, nb):
        h *= 2
    # padded layout: A at 0..h-1 (pads at na..h-1), B at h..2h-1 (pads at h+nb..)
    def real(w):
        if w < h:
            return w if w < na else None
        return na + (w - h) if w - h < nb else None
    raw = oddeven_merge_net(2 * h)
    # simulate which padded wires hold +inf pads: pads stay at the top positions of the output:
    # a comparator (a,b) with b a pad-only wire is a no-op.  Track "is pad" symbolically by running
    # on the all-real-finite assumption: output positions >= na+nb are pads.  We instead run the
    # network on symbolic sets: wire value = 'pad' or 'real'.
    state = ['real' if real(w) is not None else 'pad' for w in range(2 * h)]
    # map padded wire -> a physical register name (wire id in [0, na+nb)) dynamically
    phys = [real(w) for w in range(2 * h)]
    ops = []
    for (a, b) in raw:
        sa, sb = state[a], state[b]
        if sa == 'pad' and sb == 'pad':
            continue
        if sb == 'pad':            # min stays in a, pad stays in b
            continue
        if sa == 'pad':            # real value moves down to a, pad moves up to b
            state[a], state[b] = 'real', 'pad'
            phys[a], phys[b] = phys[b], None
            continue
        ops.append((phys[a], phys[b]))
    outmap = {}
    k = 0
    for w in range(2 * h):
        if state[w] == 'real':
            outmap[k] = phys[w]
            k += 1
    assert k == na + nb
    # liveness pruning
    live = {outmap[k] for k in want}
    kept = []
    for (a, b) in reversed(ops):
        la, lb = a in live, b in live
        if not (la or lb):
            continue
        kept.append(("CX" if la and lb else ("CMIN" if la else "CMAX"), a, b))
        live.add(a); live.add(b)
    kept.reverse()
    return kept, {k: outmap[k] for k in want}

def run(net, v):
    v = list(v)
    for kind, a, b in net:
        lo, hi = min(v[a], v[b]), max(v[a], v[b])
        if kind == "CX":
            v[a], v[b] = lo, hi
        elif kind == "CMIN":
            v[a], v[b] = lo, None
        else:
            v[a], v[b] = None, hi
    return v

def nops(net):
    return sum(2 if k == "CX" else 1 for k, _, _ in net)

SORT5 = [("CX", a, b) for a, b in [(0, 1), (3, 4), (2, 4), (2, 3), (0, 3), (0, 2), (1, 4), (1, 3), (1, 2)]]
for perm in itertools.permutations(range(5)):
    assert run(SORT5, perm) == [0, 1, 2, 3, 4]

M55, M55_OUT = merge_net(5, 5, range(10))
for ta in range(6):
    for tb in range(6):
        a = [0] * (5 - ta) + [1] * ta
        b = [0] * (5 - tb) + [1] * tb
        out = run(M55, a + b)
        assert [out[M55_OUT[k]] for k in range(10)] == sorted(a + b)

MID_WANT = list(range(7, 13))        # 0-based ranks 7..12 = 1-based 8..13
MID, MID_OUT = merge_net(10, 10, MID_WANT)
for ta in range(11):
    for tb in range(11):
        a = [0] * (10 - ta) + [1] * ta
        b = [0] * (10 - tb) + [1] * tb
        out = run(MID, a + b)
        s = sorted(a + b)
        assert [out[MID_OUT[k]] for k in MID_WANT] == [s[k] for k in MID_WANT]

def final6(c, a):
    """6th smallest of sorted c (6 values) u sorted a (5 values)."""
    return min(c[5], max(a[0], c[4]), max(a[1], c[3]), max(a[2], c[2]), max(a[3], c[1]), max(a[4], c[0]))

# whole scheme on random data (with many ties)
random.seed(1)
for trial in range(3000):
    rows = [[random.choice([random.random(), round(random.random() * 4) / 4]) for _ in range(5)] for _ in range(6)]
    S = [run(SORT5, r) for r in rows]
    # core = rows 1..4, windows rows 0..4 and 1..5
    pa = run(M55, S[1] + S[2]); pa = [pa[M55_OUT[k]] for k in range(10)]
    pb = run(M55, S[3] + S[4]); pb = [pb[M55_OUT[k]] for k in range(10)]
    m = run(MID, pa + pb); core = [m[MID_OUT[k]] for k in MID_WANT]
    for extra, lo in ((0, 0), (5, 1)):
        want = sorted(sum(rows[lo:lo + 5], []))[12]
        assert final6(core, S[extra]) == want

print("sort5", nops(SORT5), "ops; merge55", len(M55), "exchanges", nops(M55), "ops; mid20", len(MID), "exchanges", nops(MID), "ops")
print("per 2 rows:", 2 * nops(SORT5) + nops(M55) + nops(MID) + 2 * 8, "ops ->", (2 * nops(SORT5) + nops(M55) + nops(MID) + 16) / 2, "per row")

def emit(name, net):
    lines = [f"#define {name}(CX, CMIN, CMAX) \\"]
    for i in range(0, len(net), 6):
        lines.append("  " + " ".join(f"{k}({a},{b})" for k, a, b in net[i:i + 6]) + " \\")
    lines.append("  /* end */")
    return "\n".join(lines)

with open("median_shared_nets.h", "w") as f:
    f.write("/* GENERATED by tools/gen_median_shared.py -- do not edit.  See that file for the scheme.\n"
            " * CX(a,b): v[a],v[b] = min,max.  CMIN(a,b): only v[a] = min is live.  CMAX(a,b): only v[b] = max.\n"
            f" * SORT5: {nops(SORT5)} ops.  MERGE55 (wires 0-4 = A sorted, 5-9 = B sorted): {nops(M55)} ops.\n"
            f" * MID20 (wires 0-9 = Pa sorted, 10-19 = Pb sorted): {nops(MID)} ops; ranks 8..13 (1-based) of the 20. */\n")
    f.write(emit("DCMT_SORT5_NET", SORT5) + "\n")
    f.write(emit("DCMT_MERGE55_NET", M55) + "\n")
    f.write("/* wire holding the k-th smallest (k = 0..9) after DCMT_MERGE55_NET */\n")
    f.write("#define DCMT_MERGE55_OUT { " + ", ".join(str(M55_OUT[k]) for k in range(10)) + " }\n")
    f.write(emit("DCMT_MID20_NET", MID) + "\n")
    f.write("/* wires holding ranks 8,9,10,11,12,13 (1-based, ascending) after DCMT_MID20_NET */\n")
    f.write("#define DCMT_MID20_OUT { " + ", ".join(str(MID_OUT[k]) for k in MID_WANT) + " }\n")
