"""How long a freshly started process needs to reach its steady step time: 20-step windows of the 1024-frame step from a cold start."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_completion_mt_amd import Context, make_params, synth
B = int(os.environ.get("B", "1024"))
d = torch.from_numpy(synth.synth_batch(32, 352, 1216, 0)).cuda().repeat((B + 31) // 32, 1, 1)[:B].contiguous()
o = torch.empty_like(d)
ctx = Context(0, 352, 1216, B); p = make_params()
if os.environ.get("PRE") == "torch":        # keep the GPU busy with something else first: clocks or first touch?
    z = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        z.add_(1.0); torch.cuda.synchronize()
if os.environ.get("PRE") == "one":          # a single step first (touches every scratch page once), then 0.3 s of idling
    ctx.complete_dev(d, o, p); torch.cuda.synchronize(); time.sleep(0.3)
torch.cuda.synchronize()
t_start = time.perf_counter()
for w in range(int(os.environ.get("W", "16"))):
    t0 = time.perf_counter()
    for _ in range(20): ctx.complete_dev(d, o, p)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"steps {w * 20:4d}..{w * 20 + 19:4d} ({(t0 - t_start) * 1e3:6.0f} ms after the first launch): {dt * 1e3:.4f} ms per step, {B / dt:.0f} frames/s")
