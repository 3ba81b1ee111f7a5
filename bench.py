#!/usr/bin/env python3
"""bench.py -- frames/s of the img_completion cascade on device-resident batches.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A step = one pass of the hot path (dcmt_complete_f32_dev, all kernels of the cascade) over the device-resident frames
this rank owns.  Default = BASELINE configs[4] as written: `--total-frames 1024` synthetic 352x1216 sparse-depth frames
sharded per frame over the N ranks (depth_completion_mt_amd.sharding.shard_range: 1024 / 512 / 256 / 128 per GPU at
N = 1 / 2 / 4 / 8), i.e. STRONG scaling; at N = 1 that is the 1024-frame batch the metric is quoted on.  `--weak` gives
every rank its own `--batch` frames instead (and says so in "scaling").  Frames are independent: NO collective on the
data path; torch.distributed carries the barriers and the max-over-ranks of the elapsed time only.

Prints ONE JSON line (rank 0).  value = frames/s over all ranks.
  roofline      whole cascade of one step on one GPU: algorithmic bytes (8 B/px: sparse f32 in + dense f32 out, intermediates
                count zero) / mean GPU time per step, HIP events on the launch stream; .per_kernel = the library's own events
                around its two kernels (dcmt_set_kernel_timing; named by dcmt_last_path) in an extra untimed pass; .valu = the instruction-issue
                side (what actually binds both kernels), from the SQ counters of profiles/valu_latest.json; .traffic from the
                PMC passes of profiles/traffic_latest.json (both written by tools/collect_profiles.sh on the builder's box).
  configs       (rank 0, N = 1) the other BASELINE configs, each with its own roofline: [1] batch = 1 streamed,
                [2] DC_lidar_camera 352x1216 + labels, [3] DC_stereo_lidar 375x1242 + labels; plus the per-GPU workload of
                configs[4] at 8 GPUs (128 frames per step) on this one GPU.
  cpu_baseline  the CPU oracle (a port: OpenCV is not installed, the reference cannot be built) on this box's host cores,
                frame-parallel, bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ROWS, COLS = 352, 1216
BYTES_PER_FRAME = ROWS * COLS * 8           # read 1,712,128 + write 1,712,128
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0               # same guide: measured float4-copy ceiling
N_SIMD, CLOCK_HZ = 1024, 2.4e9              # 256 CUs x 4 SIMDs, max clock
METRIC = "depth frames/sec at 1216x352 (KITTI); achieved HBM GB/s vs peak"


def cpu_baseline(n_threads: int, cpu_work_s: float = 16.0):
    """Oracle (kind 'port') on the host cores: a bounded sample (about `cpu_work_s` seconds of CPU work) of the
    same synthetic frames, frame-parallel over n_threads OpenMP threads."""
    import numpy as np
    from depth_completion_mt_amd import synth
    from oracle import oracle as O          # checker/baseline only; never on the product path
    try:
        O.lib(native=True)
        native = True
    except Exception:
        native = False
    one = synth.synth_batch(1, ROWS, COLS, 0)
    O.img_completion_batch(one, threads=1, native=native)             # page the library in
    t0 = time.perf_counter()
    O.img_completion_batch(one, threads=1, native=native)
    t1 = time.perf_counter() - t0                       # single-thread seconds per frame
    n = int(max(2 * n_threads, min(cpu_work_s / max(t1, 1e-3), 1024)))
    n -= n % n_threads
    uniq = synth.synth_batch(min(n, 64), ROWS, COLS, 1000)
    frames = np.concatenate([uniq] * ((n + len(uniq) - 1) // len(uniq)))[:n]
    t0 = time.perf_counter()
    O.img_completion_batch(frames, threads=n_threads, native=native)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": n_threads, "kind": "port",
            "sample": f"{n} synthetic {COLS}x{ROWS} frames ({len(uniq)} distinct), frame-parallel OpenMP over {n_threads} threads, "
                      f"oracle/dcmt_oracle.c ({'-O3 -march=native' if native else '-O2'}), ~{n * t1:.0f} s of CPU work; "
                      f"OpenCV absent so the reference itself cannot run",
            "single_thread_frames_per_s": 1.0 / t1}


def frames_of_rank(args, rank: int, world: int) -> int:
    """Frames this rank owns per step: a shard of --total-frames (strong scaling, the default), or --batch (--weak)."""
    from depth_completion_mt_amd import sharding
    if args.weak:
        return args.batch
    b, e = sharding.shard_range(args.total_frames, rank, world)
    return e - b


def all_reduce_max(dist, values, device=None):
    """MAX over ranks of a list of floats (identity without a process group)."""
    if dist is None:
        return list(values)
    import torch
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def dry_run(args, rank, world):
    """The rank protocol of the real run -- shard, barrier, K timed steps, barrier, max over ranks (status first), one JSON
    line from rank 0 -- over gloo with the GPU step replaced by a sleep.  Exercised by tests/test_sharding.py on CPU."""
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B = frames_of_rank(args, rank, world)
    total = args.batch * world if args.weak else args.total_frames
    for _ in range(args.warmup):
        time.sleep(0.001)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (rank + 1))
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    failed = 1.0 if (args.dry_run_fail_rank is not None and rank == args.dry_run_fail_rank) else 0.0
    failed, elapsed = all_reduce_max(dist, [failed, elapsed])          # every rank learns of a failure BEFORE anyone leaves
    if dist is not None:
        dist.destroy_process_group()
    if failed:
        raise SystemExit("a rank reported a failed step: no result")
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": total * args.steps / elapsed,
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if args.weak else "strong",
                          "vs_baseline": None, "dtype": "f32", "data": "dry run (no GPU work)",
                          "config": {"workload": "dry run", "total_frames_per_step": total, "frames_per_gpu_per_step": B}}))


def timed(torch, fn, n, stream):
    """Mean milliseconds of n calls of fn, HIP events on `stream` (after 2 untimed calls)."""
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(n):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def verify_against_oracle(np, out_frames, in_frames, what: str):
    """Output frames of the TIMED buffers against the CPU oracle on the same inputs, bit for bit (the oracle is the checker here,
    never the thing measured).  Returns None if all agree, else a description of the first difference."""
    from oracle import oracle as O          # checker only
    for k, (got, x) in enumerate(zip(out_frames, in_frames)):
        want = O.img_completion(np.ascontiguousarray(x))
        if not np.array_equal(np.ascontiguousarray(got).view(np.uint32), want.view(np.uint32)):
            bad = int((got.view(np.uint32) != want.view(np.uint32)).sum())
            return f"{what}: checked frame {k} differs from the oracle in {bad} pixels"
    return None


def hbm_roofline(bytes_per_step: float, ms: float) -> dict:
    a = bytes_per_step / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
            "frac_of_measured_copy_ceiling": a / HBM_COPY_CEILING_GBS, "gpu_ms_per_step": ms, "algorithmic_bytes_per_step": bytes_per_step}


def other_configs(torch, local_rank, d_src, d_dst, params, stream, steps):
    """BASELINE configs [1], [2], [3] and the per-GPU share of [4] at 8 GPUs, on this one GPU, each with its own roofline
    (reference call sites: DC_lidar_only/main.cpp:93, DC_lidar_camera/main_lc.cpp:220, DC_stereo_lidar/main_sl.cpp:540)."""
    from depth_completion_mt_amd import Context, make_params, synth
    out = {}
    B = d_src.shape[0]
    # [1] DC_lidar_only, batch = 1 streamed: one frame per call, back to back on one stream (device-resident frames; the
    # PCIe-inclusive forms are in --batch1).  Latency-bound by construction: a frame's ideal HBM time is 0.43 us.
    c1 = Context(local_rank, ROWS, COLS, 1)
    k = [0]
    def one():
        c1.complete_dev(d_src[k[0] % B], d_dst[k[0] % B], params, stream=stream.cuda_stream)
        k[0] += 1
    ms = timed(torch, one, 200, stream)
    out["1"] = {"workload": f"DC_lidar_only, {COLS}x{ROWS}, batch=1 streamed (one dcmt_complete_f32_dev call per frame, one stream)",
                "value": 1e3 / ms, "unit": "frames/s", "us_per_frame": ms * 1e3, "kernels": c1.last_path(), "roofline": hbm_roofline(BYTES_PER_FRAME, ms)}
    c1.close()
    # 8 frames per call: the smallest batches go through the streaming kernels in row bands (from 3 frames on)
    b8 = min(8, B)
    c8 = Context(local_rank, ROWS, COLS, b8)
    ms = timed(torch, lambda: c8.complete_dev(d_src[:b8], d_dst[:b8], params, stream=stream.cuda_stream), 200, stream)
    out["8_frames_per_call"] = {"workload": f"DC_lidar_only, {COLS}x{ROWS}, {b8} device-resident frames per call, one stream",
                                "value": b8 * 1e3 / ms, "unit": "frames/s", "us_per_frame": ms * 1e3 / b8, "kernels": c8.last_path(),
                                "roofline": hbm_roofline(b8 * BYTES_PER_FRAME, ms)}
    c8.close()
    # [4] at 8 GPUs = 128 frames per GPU per step: that per-GPU workload on this GPU
    b128 = min(128, B)
    c4 = Context(local_rank, ROWS, COLS, b128)
    ms = timed(torch, lambda: c4.complete_dev(d_src[:b128], d_dst[:b128], params, stream=stream.cuda_stream), steps, stream)
    out["4_per_gpu_share"] = {"workload": f"DC_lidar_only, {COLS}x{ROWS}, {b128} frames per step (the per-GPU shard of 1024 frames on 8 GPUs)",
                              "value": b128 * 1e3 / ms, "unit": "frames/s", "kernels": c4.last_path(), "roofline": hbm_roofline(b128 * BYTES_PER_FRAME, ms)}
    c4.close()
    # the same 128 frames as two contexts x 64 frames on two streams, never joined: what a rank that streams its shard through two
    # contexts gets (one call's tail under the next call's head; the stream-ordered single call above cannot have that)
    if b128 >= 2:
        h = b128 // 2
        cs2 = [Context(local_rank, ROWS, COLS, h) for _ in range(2)]
        ss2 = [torch.cuda.Stream() for _ in range(2)]
        def two():
            for k2 in range(2):
                cs2[k2].complete_dev(d_src[k2 * h:(k2 + 1) * h], d_dst[k2 * h:(k2 + 1) * h], params, stream=ss2[k2].cuda_stream)
        for _ in range(40):
            two()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            two()
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t0) / 200 * 1e3
        out["4_per_gpu_share_in_flight"] = {"workload": f"DC_lidar_only, {COLS}x{ROWS}, {2 * h} frames per step as 2 contexts x {h} frames on 2 streams, never joined; "
                                                        "40 untimed + 200 timed steps, host clock around a device synchronise",
                                            "value": 2 * h * 1e3 / ms2, "unit": "frames/s", "roofline": hbm_roofline(2 * h * BYTES_PER_FRAME, ms2)}
        for c in cs2:
            c.close()
    # the same step on frames that are NOT multiples of 1/256 m (every depth scaled by 1.001): the 16-bit form of X6 does not apply, the
    # first call pays the attempt and the f32 rerun, the following ones go straight to the f32 kernels (k_pre_p -> k_fp_s) -- the
    # rate a caller with arbitrary f32 depths gets
    cg = Context(local_rank, ROWS, COLS, B)
    d_off = d_src * 1.001
    for _ in range(2):
        cg.complete_dev(d_off, d_dst, params, stream=stream.cuda_stream)
        torch.cuda.synchronize()
    ms = timed(torch, lambda: cg.complete_dev(d_off, d_dst, params, stream=stream.cuda_stream), steps, stream)
    out["4_off_grid_f32"] = {"workload": f"DC_lidar_only, {COLS}x{ROWS}, {B} frames per step whose depths are no multiples of 1/256 m: f32 kernels throughout (X6 as f32)",
                             "value": B * 1e3 / ms, "unit": "frames/s", "roofline": hbm_roofline(B * BYTES_PER_FRAME, ms)}
    out["4_off_grid_f32"]["kernels"] = cg.last_path()
    # frames that are valid from row 0 on (the lower half of every frame twice): nothing for the leading-row scan of k_pre_p and the
    # top-zone skip of k_fp_* to leave out -- what the headline owes to the generator's empty upper third
    d_full = torch.cat([d_src[:, ROWS // 2:], d_src[:, ROWS // 2:]], dim=1).contiguous()
    for _ in range(2):
        cg.complete_dev(d_full, d_dst, params, stream=stream.cuda_stream)
        torch.cuda.synchronize()
    ms = timed(torch, lambda: cg.complete_dev(d_full, d_dst, params, stream=stream.cuda_stream), steps, stream)
    it_full, st_full = cg.last_fill_iters(B)
    out["4_dense_from_row_0"] = {"workload": f"DC_lidar_only, {COLS}x{ROWS}, {B} frames per step, every frame valid from row 0 on (its lower half stacked twice: ~7 % valid, no empty upper third)",
                                 "value": B * 1e3 / ms, "unit": "frames/s", "converged": st_full == 0, "kernels": cg.last_path(), "roofline": hbm_roofline(B * BYTES_PER_FRAME, ms)}
    cg.close()
    del d_off, d_full
    # the same 1024-frame step as four parts of 256 frames in flight on four streams, after the GPU has been busy for a while
    if B >= 1024:
        n4 = B // 4
        cs = [Context(local_rank, ROWS, COLS, n4) for _ in range(4)]
        ss = [torch.cuda.Stream() for _ in range(4)]
        def four():
            for k in range(4):
                cs[k].complete_dev(d_src[k * n4:(k + 1) * n4], d_dst[k * n4:(k + 1) * n4], params, stream=ss[k].cuda_stream)
        for _ in range(40):
            four()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        for _ in range(60):
            four()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 60 * 1e3
        out["4_in_flight"] = {"workload": f"DC_lidar_only, {COLS}x{ROWS}, {4 * n4} frames per step as 4 contexts x {n4} frames on 4 streams, never joined "
                                          "(k_pre of one part beside k_fp_s of another); 40 untimed + 60 timed steps, host clock around a device synchronise",
                              "value": 4 * n4 * 1e3 / ms, "unit": "frames/s", "roofline": hbm_roofline(4 * n4 * BYTES_PER_FRAME, ms)}
        for c in cs:
            c.close()
    # [2] DC_lidar_camera 352x1216 + int32 label plane (~1200 superpixels, main_lc.cpp:188-197); [3] DC_stereo_lidar 375x1242 (~100, main_sl.cpp:443)
    for key, rows, cols, nt, name in (("2", 352, 1216, 1200, "DC_lidar_camera"), ("3", 375, 1242, 100, "DC_stereo_lidar")):
        Bl = 256
        lab, nl = synth.synth_labels(rows, cols, nt, 0)
        d = torch.from_numpy(synth.synth_batch(8, rows, cols, 0)).cuda().repeat(Bl // 8, 1, 1).contiguous()
        dl = torch.from_numpy(lab).cuda()[None].repeat(Bl, 1, 1).contiguous()
        o = torch.empty_like(d)
        c = Context(local_rank, rows, cols, Bl)
        ms = timed(torch, lambda: c.complete_dev(d, o, params, d_labels=dl, n_labels=nl, stream=stream.cuda_stream), max(steps // 2, 5), stream)
        c.set_kernel_timing(True)
        c.complete_dev(d, o, params, d_labels=dl, n_labels=nl, stream=stream.cuda_stream)
        kt = c.last_kernel_times()
        iters, st = c.last_fill_iters(Bl)
        out[key] = {"workload": f"{name} interpolate_with_superpixels, {cols}x{rows} f32 + int32 labels ({nl} labels), device-resident batch of {Bl}",
                    "value": Bl * 1e3 / ms, "unit": "frames/s", "converged": st == 0, "kernels": c.last_path(),
                    "roofline": dict(hbm_roofline(Bl * rows * cols * 12, ms), per_kernel_ms={"label_stage": kt["front"], "k_pre": kt["k_pre"], "k_fp": kt["k_fp"]})}
        c.close()
        del d, dl, o
        # the same at 1024 frames per step (the batch the headline is quoted on: 256 frames are 1.4 rounds of k_fp_s waves)
        Bb = 1024
        d = torch.from_numpy(synth.synth_batch(8, rows, cols, 0)).cuda().repeat(Bb // 8, 1, 1).contiguous()
        dl = torch.from_numpy(lab).cuda()[None].repeat(Bb, 1, 1).contiguous()
        o = torch.empty_like(d)
        c = Context(local_rank, rows, cols, Bb)
        ms = timed(torch, lambda: c.complete_dev(d, o, params, d_labels=dl, n_labels=nl, stream=stream.cuda_stream), 5, stream)
        out[key]["at_batch_1024"] = {"value": Bb * 1e3 / ms, "unit": "frames/s", "roofline": hbm_roofline(Bb * rows * cols * 12, ms)}
        c.close()
        del d, dl, o
    return out


def next_rows(torch, np, local_rank, d_src, d_dst, stream):
    """The rows either side of the path (tools/time_norm.py, time_project.py, time_slic.py, time_stereo.py
    are the stand-alone forms): N1 min-max normalisation fused in front, N2 LiDAR projection, N3 SLIC labels, N4 stereo
    refinement -- each with the algorithmic bytes it must move and its fraction of the HBM peak."""
    from depth_completion_mt_amd import Context, make_params, synth
    out, cs = {}, stream.cuda_stream
    B = d_src.shape[0]
    c = Context(local_rank, ROWS, COLS, B)
    pn = make_params(normalize=(0, 80))
    ms = timed(torch, lambda: c.complete_dev(d_src, d_dst, pn, stream=cs), 10, stream)
    out["N1_normalize_complete"] = {"workload": f"cv::normalize(0, 80) + img_completion, {COLS}x{ROWS}, {B} frames per step (SL/main_sl.cpp:370)",
                                    "value": B * 1e3 / ms, "unit": "frames/s", "roofline": hbm_roofline(B * BYTES_PER_FRAME, ms)}
    # the KITTI uint16 payload as input (LO/main.cpp:75-82: imread(ANYDEPTH) + convertTo(CV_32F, 1/256) fused into the first load): 6 B/px
    u16 = torch.round(d_src * 256.0).to(torch.int32).to(torch.int16)          # the synthetic depths are multiples of 1/256 below 128 m
    ms = timed(torch, lambda: c.complete_u16_dev(u16, 1.0 / 256.0, d_dst, make_params(), stream=cs), 10, stream)
    out["N1_uint16_ingest_complete"] = {"workload": f"uint16 depth payload -> metres -> img_completion, {COLS}x{ROWS}, {B} frames per step (LO/main.cpp:75-93)",
                                        "value": B * 1e3 / ms, "unit": "frames/s", "roofline": hbm_roofline(B * ROWS * COLS * 6, ms)}
    del u16
    c.close()
    rows, cols, Bp, N = 375, 1242, 256, 120000
    pts = torch.from_numpy(np.concatenate([synth.synth_points(N, i) for i in range(8)])).cuda().repeat(Bp // 8, 1).contiguous()
    off = torch.arange(0, (Bp + 1) * N, N, dtype=torch.int32, device="cuda")
    sp = torch.empty((Bp, rows, cols), dtype=torch.float32, device="cuda")
    c = Context(local_rank, rows, cols, Bp)
    ms = timed(torch, lambda: c.project_points_dev(pts, off, synth.KITTI_T_VELO_TO_CAM, synth.KITTI_P2, rows, cols, sp, stream=cs), 10, stream)
    out["N2_project"] = {"workload": f"velodyne sweep -> sparse depth image, {Bp} sweeps x {N} points -> {cols}x{rows} (SL/main_sl.cpp:478-520)",
                         "value": Bp * 1e3 / ms, "unit": "sweeps/s", "roofline": hbm_roofline(Bp * (N * 16 + rows * cols * 4), ms)}
    del pts, sp
    trip = [synth.synth_stereo(rows, cols, i) for i in range(4)]
    cu = lambda k: torch.from_numpy(np.stack([t[k] for t in trip])).cuda().repeat(Bp // 4, 1, 1).contiguous()
    l, r, g = cu(0), cu(1), cu(2)
    o = torch.empty_like(g)
    ms = timed(torch, lambda: c.stereo_refine_dev(g, l, r, o, stream=cs), 10, stream)
    out["N4_stereo_refine"] = {"workload": f"photometric refinement, {Bp} rectified pairs of {cols}x{rows}, 4 sweeps (SL/main_sl.cpp:715-885)",
                               "value": Bp * 1e3 / ms, "unit": "pairs/s", "roofline": hbm_roofline(Bp * rows * cols * 10, ms)}
    c.close()
    del l, r, g, o
    Bs = 64
    for key, rows, cols, nsp, nc, name in (("N3_slic_lidar_camera", 352, 1216, 1200, 50, "LC/main_lc.cpp:188-200"), ("N3_slic_stereo_lidar", 375, 1242, 100, 40, "SL/main_sl.cpp:443")):
        step = int(np.sqrt(rows * cols / nsp))
        imgs = torch.from_numpy(np.ascontiguousarray(np.stack([synth.synth_lab(rows, cols, i) for i in range(4)]))).cuda().repeat(Bs // 4, 1, 1, 1).contiguous()
        lab = torch.empty((Bs, rows, cols), dtype=torch.int32, device="cuda")
        c = Context(local_rank, rows, cols, Bs)
        ms = timed(torch, lambda: c.slic_labels_dev(imgs, step, nc, lab, stream=cs), 3, stream)
        out[key] = {"workload": f"Slic::generate_superpixels, {cols}x{rows} Lab, step {step}, nc {nc}, 10 iterations, {Bs} images per call ({name})",
                    "value": Bs * 1e3 / ms, "unit": "images/s", "roofline": hbm_roofline(Bs * rows * cols * 70, ms)}
        c.close()
        del imgs, lab
    return out


def batch1_extras(torch, np, L, local_rank, host, uniq, d_src, d_dst, params):
    """--batch1: BASELINE configs[1] in its PCIe-inclusive forms (never the reported value)."""
    from depth_completion_mt_amd import Context, make_params
    import ctypes
    extra = {}
    B = d_src.shape[0]
    n1 = 400
    for nstreams in (1, 4):
        ctxs = [Context(local_rank, ROWS, COLS, 1) for _ in range(nstreams)]
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        def run(n):
            for i in range(n):
                k = i % nstreams
                ctxs[k].complete_dev(d_src[i % B], d_dst[i % B], params, stream=streams[k].cuda_stream)
        run(4 * nstreams)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(n1)
        torch.cuda.synchronize()
        extra["batch1_streamed_frames_per_s" if nstreams == 1 else f"batch1_streamed_{nstreams}streams_frames_per_s"] = n1 / (time.perf_counter() - t1)
        for c in ctxs:
            c.close()
    # host-pinned input and output, frames arriving one at a time: per frame an H2D copy, the cascade on that one frame, a D2H
    # copy, all stream-ordered; four streams keep the PCIe copies of one frame under the kernels of another
    for nstreams in (1, 4):
        ctxs = [Context(local_rank, ROWS, COLS, 1) for _ in range(nstreams)]
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        h_in = [torch.from_numpy(host[i % uniq]).pin_memory() for i in range(nstreams)]
        h_out = [torch.empty((ROWS, COLS), dtype=torch.float32).pin_memory() for _ in range(nstreams)]
        dv_in = [torch.empty((ROWS, COLS), dtype=torch.float32, device="cuda") for _ in range(nstreams)]
        dv_out = [torch.empty_like(dv_in[0]) for _ in range(nstreams)]
        def run_pinned(n):
            for i in range(n):
                k = i % nstreams
                with torch.cuda.stream(streams[k]):
                    dv_in[k].copy_(h_in[k], non_blocking=True)
                    ctxs[k].complete_dev(dv_in[k], dv_out[k], params, stream=streams[k].cuda_stream)
                    h_out[k].copy_(dv_out[k], non_blocking=True)
        run_pinned(4 * nstreams)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_pinned(n1)
        torch.cuda.synchronize()
        extra["batch1_host_pinned_frames_per_s" if nstreams == 1 else f"batch1_host_pinned_{nstreams}streams_frames_per_s"] = n1 / (time.perf_counter() - t1)
        for c in ctxs:
            c.close()
    # the drop-in itself: dcmt_complete_f32 on pageable host arrays, one synchronous call per frame (what the cv::Mat shim does)
    hctx = Context(local_rank, ROWS, COLS, 1)
    hp = make_params()
    h_dst = np.empty((ROWS, COLS), np.float32)
    def host_call(i):
        src = host[i % uniq]
        st = L.lib().dcmt_complete_f32(hctx._h, src.ctypes.data, src.strides[0], 0, h_dst.ctypes.data, h_dst.strides[0], 0,
                                       ROWS, COLS, 1, ctypes.byref(hp))
        assert st == L.OK
    for i in range(8):
        host_call(i)
    t1 = time.perf_counter()
    for i in range(n1):
        host_call(i)
    extra["batch1_host_api_frames_per_s"] = n1 / (time.perf_counter() - t1)
    hctx.close()
    # the same single-frame call captured once into a HIP graph and replayed
    gctx = Context(local_rank, ROWS, COLS, 1)
    gs = torch.cuda.Stream()
    gsrc, gdst = d_src[0].clone(), torch.empty_like(d_src[0])
    with torch.cuda.stream(gs):
        gctx.complete_dev(gsrc, gdst, params, stream=gs.cuda_stream)          # warm up outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=gs):
        gctx.complete_dev(gsrc, gdst, params, stream=torch.cuda.current_stream().cuda_stream)
    for _ in range(8):
        graph.replay()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n1):
        graph.replay()
    torch.cuda.synchronize()
    extra["batch1_graph_replay_frames_per_s"] = n1 / (time.perf_counter() - t1)
    gctx.close()
    return extra


def load_profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=30,
                    help="untimed steps first (a freshly started process needs ~20 steps of the 1024-frame step to reach its steady step time: tools/time_warmup.py)")
    ap.add_argument("--total-frames", type=int, default=1024, help="frames per step over ALL ranks, sharded per frame (BASELINE configs[4]: 1024)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: every rank owns --batch frames per step")
    ap.add_argument("--batch", type=int, default=1024, help="--weak: frames per GPU per step")
    ap.add_argument("--unique", type=int, default=32, help="distinct synthetic frames per GPU (tiled to the rank's frame count)")
    ap.add_argument("--parts", type=int, default=1,
                    help="contexts (one stream each) a rank's frames are spread over, never joined inside the timed region (default 1: one call "
                         "per step; configs['4_in_flight'] reports four)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the sub-results for BASELINE configs [1] [2] [3]")
    ap.add_argument("--batch1", action="store_true", help="also time the PCIe-inclusive forms of configs[1]")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the multi-rank protocol (gloo, no GPU work, value is meaningless): tests only")
    ap.add_argument("--dry-run-fail-rank", type=int, default=None, help="tests only: that rank reports a failed step")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not args.weak and args.total_frames < world:
        raise SystemExit("--total-frames must give every rank at least one frame")
    if args.dry_run:
        return dry_run(args, rank, world)

    import numpy as np
    import torch
    from depth_completion_mt_amd import Context, make_params, synth
    from depth_completion_mt_amd import _lib as L
    if not torch.cuda.is_available() or L.lib().dcmt_device_count() < 1:
        raise SystemExit("bench.py needs a gfx950 GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    B = frames_of_rank(args, rank, world)                # frames this rank owns per step
    total = args.batch * world if args.weak else args.total_frames
    uniq = min(args.unique, B)
    host = synth.synth_batch(uniq, ROWS, COLS, seed0=rank * 100003)
    d_uniq = torch.from_numpy(host).cuda()
    reps = (B + uniq - 1) // uniq
    d_src = d_uniq.repeat(reps, 1, 1)[:B].contiguous()
    d_dst = torch.empty_like(d_src)
    params = make_params()                              # the reference's literals; 1 speculative loop application
    stream = torch.cuda.current_stream()
    # --parts P: frames are independent, so a rank's frames may go to P contexts, one stream each, with nothing joining them inside
    # the timed region: k_pre (bound by memory) of one part then runs beside k_fp_s (bound by VALU issue) of another -- what a caller
    # streaming batches through the library has with several calls in flight.  Worth +5..13 % at 1024 frames in parts of 256 once
    # the GPU has been busy for some tens of milliseconds, nothing in a 20-step run from a cold start (tools/time_two_streams.py,
    # DESIGN.md section 7), so the headline stays one call per step and configs["4_in_flight"] reports the other.
    from depth_completion_mt_amd.sharding import shard_range
    parts = max(1, min(args.parts, B))
    bounds = [shard_range(B, k, parts) for k in range(parts)]
    ctxs = [Context(local_rank, ROWS, COLS, e - b) for b, e in bounds]
    streams = [torch.cuda.Stream() for _ in range(parts)] if parts > 1 else [stream]

    def step():
        for c, (b, e), s_ in zip(ctxs, bounds, streams):
            c.complete_dev(d_src[b:e], d_dst[b:e], params, stream=s_.cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()

    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    t0 = time.perf_counter()
    for e_, s_ in zip(ev0, streams):
        e_.record(s_)
    for _ in range(args.steps):
        step()
    for e_, s_ in zip(ev1, streams):
        e_.record(s_)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # HIP events on the launch streams: from the first stream's start to the last stream's end
    gpu_ms_per_step = max(a.elapsed_time(b) for a in ev0 for b in ev1) / args.steps

    # every frame must have converged inside the timed configuration (no skipped work).  The status goes through the same
    # all-reduce as the times, so that a failing rank cannot leave the others waiting in a collective.
    iters, st = [], L.OK
    for c, (b, e) in zip(ctxs, bounds):
        it_k, st_k = c.last_fill_iters(e - b)
        iters += list(it_k)
        st = st if st_k == L.OK else st_k
    # ... and what was timed must be RIGHT: frames of the timed output buffer (first, middle, last of this rank's shard) against the oracle
    check = sorted({0, B // 2, B - 1})
    mismatch = verify_against_oracle(np, [d_dst[i].cpu().numpy() for i in check], [host[i % uniq] for i in check], f"rank {rank}")
    failed, wrong, elapsed, gpu_ms_per_step = all_reduce_max(dist, [0.0 if st == L.OK else 1.0, 0.0 if mismatch is None else 1.0, elapsed, gpu_ms_per_step], device="cuda")
    if failed or wrong:
        if dist is not None:
            dist.destroy_process_group()
        raise SystemExit(mismatch or ("a rank's timed output differs from the oracle: result invalid" if wrong else
                                      "a frame needed more hole-closure applications than were enqueued: result invalid"))

    if rank == 0:
        # the same frames through ONE context on one stream (what a single call gives), and its live per-kernel split: untimed
        # steps with the library's own events around its kernel groups
        ctx = Context(local_rank, ROWS, COLS, B)
        one_ms = timed(torch, lambda: ctx.complete_dev(d_src, d_dst, params, stream=stream.cuda_stream), args.steps, stream)
        ctx.set_kernel_timing(True)
        kt = [0.0, 0.0, 0.0, 0.0]
        nk = 5
        for _ in range(nk):
            ctx.complete_dev(d_src, d_dst, params, stream=stream.cuda_stream)
            t = ctx.last_kernel_times()
            kt = [a + b / nk for a, b in zip(kt, (t["front"], t["k_pre"], t["k_fp_s"], t["behind"]))]
        ctx.set_kernel_timing(False)
        path = ctx.last_path()                      # e.g. "k_pre_p<Q16OUT> + k_fp_q"
        k_pre_name, k_fp_name = (path.split(" + ") + ["?"])[:2] if " + " in path else (path, "?")
        ctx.close()
        roof = hbm_roofline(B * BYTES_PER_FRAME, gpu_ms_per_step)
        roof["kernel"] = (f"whole cascade per step = {path} (on frames that are multiples of 1/256 m, the KITTI depth format, X6 crosses HBM as 16-bit codes; "
                          "+ the gated f32 rerun and 3 redo launches, which return at once) of every part, HIP events on the launch streams "
                          "around the K timed steps (first start to last end); with more than one part the kernels of different parts overlap, so the per-kernel "
                          "durations of a profile add up to more than the step; per_kernel and single_context: the same frames through ONE context on one stream "
                          "(the library's own events, dcmt_set_kernel_timing, 5 extra steps); rocprofv3 averages of that single-context step in profiles/")
        roof["single_context"] = dict(hbm_roofline(B * BYTES_PER_FRAME, one_ms), value=B * 1e3 / one_ms, unit="frames/s",
                                      note="one dcmt_complete_f32_dev call per step on one stream: the kernels run one after the other")
        roof["kernels"] = path
        roof["per_kernel"] = {k_pre_name: dict(hbm_roofline(B * ROWS * COLS * 4 * 2, kt[1]), ms=kt[1], note="H2..H6: reads the sparse frame, writes X6 (algorithmic bytes of this kernel alone: 8 B/px)"),
                              k_fp_name: dict(hbm_roofline(B * ROWS * COLS * 4 * 2, kt[2]), ms=kt[2], note="H7..H11 (and the gated rerun launches behind it, which return at once): reads X6, writes the dense frame"),
                              "redo_launches_ms": kt[3]}
        tr = load_profile_json("traffic_latest.json")
        roof["traffic"] = tr.get("hbm_bytes_per_step") if tr else None
        if tr and B == 1024:
            roof["traffic_over_algorithmic"] = tr["hbm_bytes_per_step"] / (B * BYTES_PER_FRAME)
        roof["traffic_source"] = "profiles/traffic_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the 1024-frame step on the builder's box (tools/collect_profiles.sh), not measured in this run"
        va = load_profile_json("valu_latest.json")
        if va and B == 1024:
            # instruction-issue side: what binds both kernels.  SQ_ACTIVE_INST_VALU counts quad-cycles (x4 = SIMD cycles issuing VALU)
            issue_ms = {k: v["active_inst_valu"] * 4.0 / (N_SIMD * CLOCK_HZ) * 1e3 for k, v in va["kernels"].items()}
            tot_issue = sum(issue_ms.values())
            roof["valu"] = {"insts_per_step": sum(v["insts_valu"] for v in va["kernels"].values()),
                            "insts_per_step_by_kernel": {k: v["insts_valu"] for k, v in va["kernels"].items()},
                            "cycles_per_inst": sum(v["active_inst_valu"] for v in va["kernels"].values()) * 4.0 / sum(v["insts_valu"] for v in va["kernels"].values()),
                            "issue_ms_per_step": tot_issue, "issue_ms_by_kernel": issue_ms,
                            "frac_of_issue": tot_issue / gpu_ms_per_step,
                            "peak": f"{N_SIMD} SIMDs x {CLOCK_HZ / 1e9:.1f} GHz; min/max/med3/cmp/cndmask/DPP issue at ~4.4 cycles per wave64 instruction (tools/issue_probe.hip)",
                            "source": "profiles/valu_latest.json: rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU pass of this step on the builder's box"}
            roof["valu_frac"] = roof["valu"]["frac_of_issue"]     # share of the step's GPU time the SIMDs spend issuing VALU instructions
            roof["limiter"] = "valu-issue" if roof["valu_frac"] > 0.6 else "mixed"
            roof["bound_note"] = ("bound = the roofline the PATH is priced against (stencils at 8 algorithmic B/px: HBM; the contract's enum); "
                                  "limiter = what the kernels are limited by TODAY (valu_frac of the step is VALU issue time)")
        line = {
            "metric": METRIC, "value": total * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None, "dtype": "f32",
            "data": f"synthetic KITTI-like sparse depth (depth_completion_mt_amd/synth.py: depths are multiples of 1/256 m like a KITTI uint16 PNG / 256), {uniq} distinct frames per GPU tiled to {B}",
            "config": {"workload": (f"DC_lidar_only img_completion, {COLS}x{ROWS} f32, {total} device-resident frames per step "
                                    + (f"= {args.batch} per GPU (weak scaling)" if args.weak else
                                       f"sharded per frame over {world} GPU(s) = {B} per GPU (BASELINE configs[4]; strong scaling)")),
                       "total_frames_per_step": total, "frames_per_gpu_per_step": B, "rows": ROWS, "cols": COLS, "k0": "as_compiled", "blur": "gaussian",
                       "sharding": "per-frame, no collective",
                       "contexts_per_gpu": parts, "frames_per_context": [e - b for b, e in bounds],
                       "in_flight": "one call per context and step, one stream per context, joined only at the ends of the timed region"},
            "roofline": roof,
            "verified": True,
            "verified_how": f"frames {check} of the timed output buffer (every rank: first, middle, last of its shard) bit-identical to the CPU oracle on the same inputs",
            "fill_iters_max": max(iters),
        }
        if world == 1 and not args.no_configs:
            line["configs"] = other_configs(torch, local_rank, d_src, d_dst, params, stream, args.steps)
        if world == 1 and not args.no_configs:
            line["next_rows"] = next_rows(torch, np, local_rank, d_src, d_dst, stream)
        if args.batch1:
            line.update(batch1_extras(torch, np, L, local_rank, host, uniq, d_src, d_dst, params))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(min(16, len(os.sched_getaffinity(0))))   # the box's CPU share for one GPU is 16
        print(json.dumps(line))
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
