"""Times the fused path's kernels by difference: stop_after 6 (k_pre_s), 7 (+k_fill_t), 11 (+loop+k_post_s)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_completion_mt_amd import Context, make_params, synth
B = int(os.environ.get("B", "512"))
d = torch.from_numpy(synth.synth_batch(16, 352, 1216, 0)).cuda().repeat(B // 16, 1, 1).contiguous()
o = torch.empty_like(d)
ctx = Context(0, 352, 1216, B)
res = {}
for stop in (6, 7, 11):
    p = make_params(stop_after=stop)
    for _ in range(3): ctx.complete_dev(d, o, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ctx.complete_dev(d, o, p)
    e1.record(); torch.cuda.synchronize()
    res[stop] = e0.elapsed_time(e1) / 10
print(f"B={B} pre {res[6]:.3f} ms  fill {res[7]-res[6]:.3f} ms  post+loop {res[11]-res[7]:.3f} ms  total {res[11]:.3f} ms  ({B/res[11]*1e3:.0f} frames/s)")
