"""frames/s of the staged vs the fused path as a function of the batch size (picks min_fused_batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_completion_mt_amd import Context, make_params, synth
src = torch.from_numpy(synth.synth_batch(64, 352, 1216, 0)).cuda()
for B in (tuple(int(a) for a in sys.argv[1:]) or (1, 2, 4, 8, 16, 32, 64)):
    ctx = Context(0, 352, 1216, B)
    d = src[:B].contiguous(); o = torch.empty_like(d)
    row = []
    for name, p in (("staged", make_params(force_staged=True)), ("fused", make_params(force_fused=True))):
        for _ in range(5): ctx.complete_dev(d, o, p)
        torch.cuda.synchronize()
        n = max(20, 400 // B)
        t0 = time.perf_counter()
        for _ in range(n): ctx.complete_dev(d, o, p)
        torch.cuda.synchronize()
        row.append(f"{name} {B * n / (time.perf_counter() - t0):9.0f} f/s")
    print(f"batch {B:3d}: " + "   ".join(row))
    ctx.close()
