"""Builder's check of the 16-bit X6 flow (k_pre_p<Q16OUT> -> k_fp_q, f32 rerun behind the flag) against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from depth_completion_mt_amd import Context, make_params, synth
from oracle import oracle as O

def run(frames, tag, idx):
    with Context(0, frames.shape[1], frames.shape[2], frames.shape[0]) as ctx:
        out = ctx.complete_dev(torch.from_numpy(frames).cuda()); torch.cuda.synchronize()
        got = out.cpu().numpy()
        iters, st = ctx.last_fill_iters(frames.shape[0])
    bad = 0
    for i in idx:
        want, info = O.img_completion(frames[i], return_info=True)
        if not np.array_equal(got[i].view(np.uint32), want.view(np.uint32)):
            d = np.argwhere(got[i].view(np.uint32) != want.view(np.uint32))
            print(tag, "frame", i, "differs at", len(d), "pixels; first", d[:3].tolist(), got[i][tuple(d[0])], want[tuple(d[0])]); bad += 1
        if iters[i] != info["fill_iters"]:
            print(tag, "frame", i, "fill iters", iters[i], info["fill_iters"]); bad += 1
    print(tag, "OK" if not bad else f"{bad} problems", "DCMT_FP_Q16 =", os.environ.get("DCMT_FP_Q16"))

frames = synth.synth_batch(16, 352, 1216, 500)
assert np.array_equal(np.round(frames * 256) / 256, frames)
frames[3, 120:170, 400:520] = 0            # needs the hole-closure loop
frames[5, :, 600:640] = 0                  # empty columns
frames[7] = 0
frames[9, :40] = 0; frames[9, 60:] = 0
frames[11, 200, 300] = 255.99609375
run(frames, "codes", (0, 3, 5, 7, 9, 11, 15))
f2 = frames.copy()
f2[2][f2[2] > 0] += np.float32(0.001)      # one frame that is no multiple of 1/256: the f32 kernels rerun
f2[13, 100, 100] = 300.0                   # out of the code range
run(f2, "mixed", (0, 2, 3, 5, 13, 15))
f3 = synth.synth_batch(16, 375, 1242, 7)
run(f3, "1242x375", (0, 8, 15))
f4 = synth.synth_batch(8, 64, 96, 9)
f4[1, 20:60, 10:80] = 0
run(f4, "96x64", range(8))
