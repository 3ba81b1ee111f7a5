"""frames/s of the chain with N1's min-max normalisation fused in front (DCMT_FLAG_NORMALIZE) vs the plain chain
(device-resident batch of 1024, 352x1216)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_completion_mt_amd import Context, make_params, synth
B = 1024
f = synth.synth_batch(32, 352, 1216, 0)
d = torch.from_numpy(f).cuda().repeat(B // 32, 1, 1).contiguous()
o = torch.empty_like(d)
ctx = Context(0, 352, 1216, B)
for name, p in (("plain", make_params()), ("normalize(0,80)", make_params(normalize=(0, 80)))):
    for _ in range(3): ctx.complete_dev(d, o, p)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ctx.complete_dev(d, o, p)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    gbs = B * 352 * 1216 * 8 / dt / 1e9       # algorithmic: sparse frame in, dense frame out (the min-max pass re-reads the input: not counted)
    print(f"{name}: {B / dt:.0f} frames/s ({dt * 1e3:.3f} ms per 1024 frames; {gbs:.0f} GB/s of the 8 B/px = {gbs / 8000:.3f} of 8 TB/s)")
