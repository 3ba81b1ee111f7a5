import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from depth_completion_mt_amd import Context, make_params, synth
d = torch.from_numpy(synth.synth_frame(352, 1216, 0)).cuda()
o = torch.empty_like(d)
ctx = Context(0, 352, 1216, 1)
p = make_params()
for _ in range(50): ctx.complete_dev(d, o, p)
torch.cuda.synchronize()
