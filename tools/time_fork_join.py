"""One 1024-frame step as TWO independent parts on two side streams, forked from and joined to the caller's stream inside every step
(what a single dcmt_complete_f32_dev call could do internally), against the single call.  SPLIT=a,b sets the part sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_completion_mt_amd import Context, make_params, synth
B = 1024
d = torch.from_numpy(synth.synth_batch(32, 352, 1216, 0)).cuda().repeat(B // 32, 1, 1).contiguous()
o = torch.empty_like(d)
p = make_params()
main = torch.cuda.current_stream()
def timed(fn, n=40, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(n): fn()
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
one = Context(0, 352, 1216, B)
ms = timed(lambda: one.complete_dev(d, o, p))
print(f"single call: {ms:.4f} ms -> {B / ms * 1e3:.0f} frames/s")
one.close()
for split in os.environ.get("SPLITS", "512,512;384,640;256,768;640,384;341,341,342;256,256,256,256").split(";"):
    sizes = [int(x) for x in split.split(",")]
    offs = [sum(sizes[:k]) for k in range(len(sizes))]
    ctxs = [Context(0, 352, 1216, s) for s in sizes]
    streams = [torch.cuda.Stream() for _ in sizes]
    fork = torch.cuda.Event()
    joins = [torch.cuda.Event() for _ in sizes]
    def step():
        fork.record(main)
        for c, s, o0, n, j in zip(ctxs, streams, offs, sizes, joins):
            s.wait_event(fork)
            c.complete_dev(d[o0:o0 + n], o[o0:o0 + n], p, stream=s.cuda_stream)
            j.record(s)
        for j in joins: main.wait_event(j)
    ms = timed(step)
    print(f"fork/join {split}: {ms:.4f} ms -> {B / ms * 1e3:.0f} frames/s")
    for c in ctxs: c.close()
