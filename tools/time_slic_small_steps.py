import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from depth_completion_mt_amd import Context, synth
rows, cols, B = 352, 1216, 8
imgs = torch.from_numpy(np.ascontiguousarray(np.stack([synth.synth_lab(rows, cols, i) for i in range(4)]))).cuda().repeat(B // 4, 1, 1, 1).contiguous()
for step in (6, 7, 8, 12):
    ctx = Context(0, rows, cols, B)
    lab = torch.empty((B, rows, cols), dtype=torch.int32, device="cuda")
    ctx.slic_labels_dev(imgs, step, 40, lab); torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.slic_labels_dev(imgs, step, 40, lab); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"step {step}: {dt * 1e3 / B:.3f} ms per image"); ctx.close()
