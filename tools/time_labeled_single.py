import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from depth_completion_mt_amd import Context, make_params, synth
for rows, cols, nt in ((352, 1216, 1200), (375, 1242, 100)):
    lab, nl = synth.synth_labels(rows, cols, nt, 0)
    d = torch.from_numpy(synth.synth_frame(rows, cols, 0)).cuda()[None].contiguous()
    dl = torch.from_numpy(lab).cuda()[None].contiguous()
    o = torch.empty_like(d)
    ctx = Context(0, rows, cols, 1)
    for name, p in (("staged", make_params(force_staged=True)), ("fused", make_params(force_fused=True)), ("default", make_params())):
        for _ in range(5): ctx.complete_dev(d, o, p, d_labels=dl, n_labels=nl)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 100
        for _ in range(n): ctx.complete_dev(d, o, p, d_labels=dl, n_labels=nl)
        torch.cuda.synchronize()
        print(f"labeled single frame {cols}x{rows} {nl} labels {name}: {(time.perf_counter()-t0)/n*1e6:.0f} us")
    ctx.close()
