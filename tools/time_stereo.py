"""Throughput of N4 (dcmt_stereo_refine_dev) on 256 rectified 1242x375 pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from depth_completion_mt_amd import Context, synth
B, rows, cols = 256, 375, 1242
trip = [synth.synth_stereo(rows, cols, i) for i in range(4)]
cu = lambda k: torch.from_numpy(np.stack([t[k] for t in trip])).cuda().repeat(B // 4, 1, 1).contiguous()
l, r, g = cu(0), cu(1), cu(2)
out = torch.empty_like(g)
ctx = Context(0, rows, cols, B)
for _ in range(3): ctx.stereo_refine_dev(g, l, r, out)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
for _ in range(n): ctx.stereo_refine_dev(g, l, r, out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
px = B * rows * cols
print(f"stereo refinement: {B / dt:.0f} pairs/s ({dt * 1e3:.3f} ms per {B} pairs, {px * 10 / dt / 1e9:.0f} GB/s of the 10 B/px it must move)")
