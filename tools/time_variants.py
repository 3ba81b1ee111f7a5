"""Per-kernel times of the 1024-frame step (the library's own events) under the dispatch toggles given in the environment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_completion_mt_amd import Context, make_params, synth
B = int(os.environ.get("B", "1024"))
d = torch.from_numpy(synth.synth_batch(32, 352, 1216, 0)).cuda().repeat((B + 31) // 32, 1, 1)[:B].contiguous()
o = torch.empty_like(d)
ctx = Context(0, 352, 1216, B)
p = make_params()
for _ in range(3): ctx.complete_dev(d, o, p)
torch.cuda.synchronize()
ctx.set_kernel_timing(True)
acc = None
n = 10
for _ in range(n):
    ctx.complete_dev(d, o, p)
    t = ctx.last_kernel_times()
    acc = t if acc is None else {k: acc[k] + t[k] for k in t}
tot = sum(acc.values()) / n
print({k: round(v / n, 4) for k, v in acc.items()}, f"total {tot:.4f} ms -> {B / tot * 1e3:.0f} frames/s", {k: v for k, v in os.environ.items() if k.startswith("DCMT_")})
