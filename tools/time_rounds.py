"""Step time against the batch size around whole rounds of k_fp_s waves (22 strips per frame, 4096 wave slots per GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_completion_mt_amd import Context, make_params, synth
base = torch.from_numpy(synth.synth_batch(32, 352, 1216, 0)).cuda()
p = make_params()
for B in (744, 838, 931, 978, 1024, 1070, 1117, 1210, 1303):
    d = base.repeat((B + 31) // 32, 1, 1)[:B].contiguous(); o = torch.empty_like(d)
    ctx = Context(0, 352, 1216, B)
    for _ in range(3): ctx.complete_dev(d, o, p)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ctx.complete_dev(d, o, p)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"batch {B}: {B * 22 / 4096:.2f} rounds of k_fp_s waves, {ms:.4f} ms per step, {ms / B * 1e3:.4f} us per frame, {B / ms:.1f} k frames/s")
    ctx.close(); del d, o
