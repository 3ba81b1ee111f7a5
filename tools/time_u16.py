"""frames/s of the uint16-ingest entry point vs the f32 one (device-resident batch of 1024)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from depth_completion_mt_amd import Context, make_params, synth
B = 1024
f = synth.synth_batch(32, 352, 1216, 0)
u = torch.from_numpy(np.round(f * 256).astype(np.uint16).view(np.int16)).cuda().repeat(B // 32, 1, 1).contiguous()
d = torch.from_numpy(f).cuda().repeat(B // 32, 1, 1).contiguous()
o = torch.empty_like(d)
ctx = Context(0, 352, 1216, B)
p = make_params()
for name, fn in (("f32", lambda: ctx.complete_dev(d, o, p)), ("u16", lambda: ctx.complete_u16_dev(u, 1 / 256., o, p))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {B / dt:.0f} frames/s ({dt * 1e3:.3f} ms per 1024 frames)")
