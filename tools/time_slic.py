"""Throughput of N3 (dcmt_slic_labels_dev) at the two executables' settings, batch 64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from depth_completion_mt_amd import Context, synth
B = 64
for rows, cols, nsp, nc in ((352, 1216, 1200, 50), (375, 1242, 100, 40)):
    step = int(np.sqrt(rows * cols / nsp))
    imgs = torch.from_numpy(np.ascontiguousarray(np.stack([synth.synth_lab(rows, cols, i) for i in range(4)]))).cuda().repeat(B // 4, 1, 1, 1).contiguous()
    ctx = Context(0, rows, cols, B)
    lab = torch.empty((B, rows, cols), dtype=torch.int32, device="cuda")
    for _ in range(2): ctx.slic_labels_dev(imgs, step, nc, lab)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 3
    for _ in range(n): ctx.slic_labels_dev(imgs, step, nc, lab)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    # algorithmic bytes: every one of the 10 iterations reads the 3-byte Lab pixel and writes the 4-byte label
    gbs = B * rows * cols * 70 / dt / 1e9
    print(f"SLIC {cols}x{rows}, step {step}, nc {nc}, batch {B}: {B / dt:.0f} images/s ({dt * 1e3 / B:.3f} ms per image; "
          f"{gbs:.0f} GB/s of the 70 B/px it must move = {gbs / 8000:.3f} of 8 TB/s)")
    ctx.close()
