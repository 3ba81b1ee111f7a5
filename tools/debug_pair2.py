import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from depth_completion_mt_amd import Context, make_params
def ctx(pair, rows, cols, b, wide=1):
    os.environ["DCMT_PAIR"] = str(pair); os.environ["DCMT_WIDE"] = str(wide)
    return Context(0, rows, cols, b)
np.set_printoptions(linewidth=250, precision=1, suppress=True)
rows, cols, b = 40, 128, 16
for wide in (1, 0):
    c0, c1 = ctx(0, rows, cols, b), ctx(1, rows, cols, b, wide)
    for pc in (126, 125, 120, 60, 3):
        fr = np.zeros((b, rows, cols), np.float32)
        fr[:, 20, pc] = 30.0
        d = torch.from_numpy(fr).cuda()
        p = make_params(stop_after=6, force_fused=True)
        a = c0.complete_dev(d, params=p).cpu().numpy(); torch.cuda.synchronize()
        g = c1.complete_dev(d, params=p).cpu().numpy(); torch.cuda.synchronize()
        lo, hi = max(pc - 12, 0), min(pc + 12, cols)
        neq = int((a.view(np.uint32) != g.view(np.uint32)).sum())
        print(f"wide{wide} pixel at col {pc}: {neq} differ; row 20 (and row 0) cols {lo}..{hi-1}")
        print("  want", a[0, 20, lo:hi]); print("  got ", g[0, 20, lo:hi]); print("  want0", a[0, 0, lo:hi]); print("  got0 ", g[0, 0, lo:hi])
    c0.close(); c1.close()
