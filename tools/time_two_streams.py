"""Does k_pre (memory-bound) of one batch part overlap k_fp_s (VALU-bound) of another?  N independent parts of a 1024-frame
batch on N streams (one context each), against the whole batch on one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_completion_mt_amd import Context, make_params, synth
B = int(os.environ.get("B", "1024"))
d = torch.from_numpy(synth.synth_batch(32, 352, 1216, 0)).cuda().repeat(B // 32, 1, 1).contiguous()
o = torch.empty_like(d)
p = make_params()
for parts in (1, 2, 3, 4, 6, 8):
    if B % parts: continue
    n = B // parts
    ctxs = [Context(0, 352, 1216, n) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def step():
        for k in range(parts):
            ctxs[k].complete_dev(d[k * n:(k + 1) * n], o[k * n:(k + 1) * n], p, stream=streams[k].cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{parts} part(s) of {n} frames on {parts} stream(s): {dt * 1e3:.3f} ms per {B} frames -> {B / dt:.0f} frames/s")
    for c in ctxs: c.close()
