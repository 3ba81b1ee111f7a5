"""k_pre_p against k_pre_s on the same inputs: where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from depth_completion_mt_amd import Context, make_params, synth
from depth_completion_mt_amd import _lib as L
def ctx(pair, rows, cols, b, wide=1):
    os.environ["DCMT_PAIR"] = str(pair); os.environ["DCMT_WIDE"] = str(wide)
    return Context(0, rows, cols, b)
rng = np.random.default_rng(0)
np.set_printoptions(linewidth=220, precision=3, suppress=True)
for rows, cols, b, wide in ((40, 128, 16, 1), (40, 128, 16, 0), (40, 130, 16, 1)):
    fr = np.where(rng.random((b, rows, cols)) < 0.06, rng.uniform(1, 80, (b, rows, cols)), 0).astype(np.float32)
    d = torch.from_numpy(fr).cuda()
    c0, c1 = ctx(0, rows, cols, b), ctx(1, rows, cols, b, wide)
    for k0 in ("as_compiled",):
        for stop in (6,):
            p = make_params(k0=k0, stop_after=stop, force_fused=True)
            a = c0.complete_dev(d, params=p).cpu().numpy(); torch.cuda.synchronize()
            g = c1.complete_dev(d, params=p).cpu().numpy(); torch.cuda.synchronize()
            neq = a.view(np.uint32) != g.view(np.uint32)
            print(f"{rows}x{cols} b{b} wide{wide} {k0} stop{stop}: {int(neq.sum())} differ")
            if neq.any():
                f = int(np.argwhere(neq)[0][0])
                cs = sorted(set(np.argwhere(neq[f])[:, 1]))
                print(" frame", f, "cols", cs[:20], "rows", sorted(set(np.argwhere(neq[f])[:, 0]))[:40])
                c_lo, c_hi = max(cs[0] - 2, 0), min(cs[0] + 10, cols)
                print(" want\n", a[f, :, c_lo:c_hi][::3]); print(" got\n", g[f, :, c_lo:c_hi][::3])
    c0.close(); c1.close()
