import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from depth_completion_mt_amd import Context, make_params, synth
d = torch.from_numpy(synth.synth_batch(32, 352, 1216, 0)).cuda().repeat(4, 1, 1).contiguous()
o = torch.empty_like(d)
ctx = Context(0, 352, 1216, 128)
p = make_params()
for _ in range(30): ctx.complete_dev(d, o, p)
torch.cuda.synchronize()
