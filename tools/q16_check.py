import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from depth_completion_mt_amd import Context, make_params, synth
from oracle import oracle as O
frames = synth.synth_batch(16, 352, 1216, 500)
assert np.array_equal(np.round(frames * 256) / 256, frames)
frames[5, :, 600:640] = 0
frames[7] = 0
frames[9, :40] = 0; frames[9, 60:] = 0
frames[11, 200, 300] = 255.99609375
with Context(0, 352, 1216, 16) as ctx:
    out = ctx.complete_dev(torch.from_numpy(frames).cuda()); torch.cuda.synchronize()
    got = out.cpu().numpy()
bad = 0
for i in (0, 5, 7, 9, 11, 15):
    want = O.img_completion(frames[i])
    ok = np.array_equal(got[i].view(np.uint32), want.view(np.uint32))
    if not ok:
        d = np.argwhere(got[i].view(np.uint32) != want.view(np.uint32))
        print("frame", i, "differs at", len(d), "pixels; first", d[:5].tolist(), got[i][tuple(d[0])], want[tuple(d[0])])
        bad += 1
print("Q16 check:", "OK" if not bad else f"{bad} frames differ", os.environ.get("DCMT_FP_Q16"))
