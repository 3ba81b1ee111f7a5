"""N3 at small batches (a caller that streams frames one at a time): ms per call and per image."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from depth_completion_mt_amd import Context, synth
rows, cols, step, nc = 352, 1216, 18, 50
base = np.ascontiguousarray(np.stack([synth.synth_lab(rows, cols, i) for i in range(4)]))
for B in (1, 2, 4, 8, 16, 64):
    imgs = torch.from_numpy(base).cuda().repeat((B + 3) // 4, 1, 1, 1)[:B].contiguous()
    ctx = Context(0, rows, cols, B)
    lab = torch.empty((B, rows, cols), dtype=torch.int32, device="cuda")
    for _ in range(3): ctx.slic_labels_dev(imgs, step, nc, lab)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 10
    for _ in range(n): ctx.slic_labels_dev(imgs, step, nc, lab)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"batch {B}: {dt * 1e3:.3f} ms per call, {dt * 1e3 / B:.3f} ms per image")
    ctx.close()
