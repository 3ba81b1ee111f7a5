"""Throughput of N2 (dcmt_project_points_dev): 256 sweeps of 120 k points -> 375x1242 sparse images, and of the whole
stereo-lidar front half on the device: project -> normalize(0,100) -> img_completion."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from depth_completion_mt_amd import Context, make_params, synth
B, N, rows, cols = 256, 120000, 375, 1242
base = [synth.synth_points(N, i) for i in range(8)]
pts = torch.from_numpy(np.concatenate(base)).cuda().repeat(B // 8, 1).contiguous()
off = torch.arange(0, (B + 1) * N, N, dtype=torch.int32, device="cuda")
ctx = Context(0, rows, cols, B)
sp = torch.empty((B, rows, cols), dtype=torch.float32, device="cuda")
out = torch.empty_like(sp)
p = make_params(normalize=(0, 100))
def run(chain):
    ctx.project_points_dev(pts, off, synth.KITTI_T_VELO_TO_CAM, synth.KITTI_P2, rows, cols, sp)
    if chain: ctx.complete_dev(sp, out, p)
for name, chain in (("project", False), ("project + normalize + complete", True)):
    for _ in range(3): run(chain)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run(chain)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    # algorithmic bytes of the projection: 16 B per point read, 4 B per pixel of the sparse image written (the chain adds 8 B/px)
    gbs = B * (N * 16 + rows * cols * (12 if chain else 4)) / dt / 1e9
    print(f"{name}: {B / dt:.0f} sweeps/s, {B * N / dt / 1e9:.2f} G points/s ({dt * 1e3:.3f} ms per {B} sweeps; "
          f"{gbs:.0f} GB/s algorithmic = {gbs / 8000:.3f} of 8 TB/s)")
