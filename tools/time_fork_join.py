"""One 1024-frame step as several parts on TWO side streams (each stream runs its parts one after the other), forked from and joined
to the caller's stream inside every step.  Unequal first parts put the two streams out of phase, so that k_pre of one part runs
beside k_fp of another.  PLANS="a,b,c|d,e;..." (parts of stream 1 | parts of stream 2)."""
import os, sys
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
plans = os.environ.get("PLANS", "512|512;256,256|256,256;128,256,128|256,256;128,384|512;256,384|384;192,320|256,256;128,128,128,128|256,256;341|341|342;256|256|256|256;128,256|256,128|256")
for plan in plans.split(";"):
    per_stream = [[int(x) for x in s.split(",")] for s in plan.split("|")]
    assert sum(sum(s) for s in per_stream) == B, plan
    ctxs, offs, off = [], [], 0
    for s in per_stream:
        ctxs.append([Context(0, 352, 1216, n) for n in s]); offs.append([])
        for n in s: offs[-1].append(off); off += n
    streams = [torch.cuda.Stream() for _ in per_stream]
    fork = torch.cuda.Event(); joins = [torch.cuda.Event() for _ in per_stream]
    def step():
        fork.record(main)
        for k, st in enumerate(streams):
            st.wait_event(fork)
        for k, st in enumerate(streams):
            for c, o0, n in zip(ctxs[k], offs[k], per_stream[k]):
                c.complete_dev(d[o0:o0 + n], o[o0:o0 + n], p, stream=st.cuda_stream)
            joins[k].record(st)
        for j in joins: main.wait_event(j)
    ms = timed(step)
    print(f"fork/join {plan}: {ms:.4f} ms -> {B / ms * 1e3:.0f} frames/s")
    for cs in ctxs:
        for c in cs: c.close()
