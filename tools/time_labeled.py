"""frames/s of the label-masked variant (BASELINE configs 2 and 3) on device-resident batches, with the library's own
per-kernel-group events (label stage = k_label_bbox + k_label_stage_*; k_pre = H5..H6 behind it; k_fp_s)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_completion_mt_amd import Context, make_params, synth
CONFIGS = {"2": (352, 1216, 1200, 256), "3": (375, 1242, 100, 256)}        # BASELINE configs[2] / [3]; LC_CONFIG=2|3 runs one of them
for rows, cols, nt, B in [CONFIGS[k] for k in sorted(CONFIGS) if os.environ.get("LC_CONFIG", k) == k]:
    lab, nl = synth.synth_labels(rows, cols, nt, 0)
    d = torch.from_numpy(synth.synth_batch(8, rows, cols, 0)).cuda().repeat(B // 8, 1, 1).contiguous()
    dl = torch.from_numpy(lab).cuda()[None].repeat(B, 1, 1).contiguous()
    o = torch.empty_like(d)
    ctx = Context(0, rows, cols, B)
    p = make_params()
    for _ in range(2): ctx.complete_dev(d, o, p, d_labels=dl, n_labels=nl)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 5
    for _ in range(n): ctx.complete_dev(d, o, p, d_labels=dl, n_labels=nl)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.set_kernel_timing(True)
    ctx.complete_dev(d, o, p, d_labels=dl, n_labels=nl)
    kt = {k: round(v, 4) for k, v in ctx.last_kernel_times().items()}
    print(f"labeled {cols}x{rows}, {nl} labels, batch {B}: {B * n / dt:.0f} frames/s ({dt / n * 1e3:.3f} ms per batch) {kt}", {k: v for k, v in os.environ.items() if k.startswith("DCMT_")})
    ctx.close()
