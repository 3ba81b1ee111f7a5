"""What the launches behind the two large kernels cost per 1024-frame step: spec_fill_iters = 1 (default: three redo launches that return at once) against
spec_fill_iters = 0 (none; the kernel then keeps the select of LO :184), with and without the 16-bit attempt (two more gated launches).  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_completion_mt_amd import Context, make_params, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
src = torch.from_numpy(synth.synth_batch(32, 352, 1216, 0)).cuda().repeat(B // 32, 1, 1).contiguous()
dst = torch.empty_like(src)
for rep in range(3):
    for q16 in ("1", "0"):
        os.environ["DCMT_FP_Q16"] = q16
        ctx = Context(0, 352, 1216, B)
        for spec in (1, 0):
            p = make_params(spec_fill_iters=spec)
            for _ in range(30): ctx.complete_dev(src, dst, p)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); n = 40
            for _ in range(n): ctx.complete_dev(src, dst, p)
            torch.cuda.synchronize()
            print(f"rep {rep} q16={q16} spec_fill_iters={spec}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per step   {ctx.last_path()}", flush=True)
        ctx.close()
