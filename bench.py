#!/usr/bin/env python3
"""bench.py -- frames/s of the img_completion cascade on device-resident batches.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A step = one pass of the hot path (dcmt_complete_f32_dev, all kernels of the cascade) over
one device-resident batch of `--batch` synthetic 352x1216 sparse-depth frames per GPU.
Frames are independent, so ranks shard the frames with NO collective on the data path
(torch.distributed is used for the barriers and the max-over-ranks of the elapsed time
only); scaling is weak: every rank processes its own `--batch` frames per step.

Prints ONE JSON line (rank 0).  value = frames/s over all ranks.  roofline.achieved =
algorithmic bytes (8 B/px: sparse f32 in + dense f32 out, intermediates count zero) per
step / mean GPU time per step measured with HIP events on the launch stream.
cpu_baseline = the CPU oracle (a port: OpenCV is not installed, the reference cannot be
built) on this box's host cores, frame-parallel, bounded sample.
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


def dry_run(args, rank, world):
    """The rank protocol of the real run -- barrier, K timed steps, barrier, max over ranks, one JSON line from
    rank 0 -- over gloo with the GPU step replaced by a sleep.  Exercised by tests/test_sharding.py on CPU."""
    import torch
    import torch.distributed as dist
    from depth_completion_mt_amd import sharding
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B = args.batch
    b, e = sharding.shard_range(B * world, rank, world)            # weak scaling: every rank owns B frames
    assert e - b == B
    for _ in range(args.warmup):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (rank + 1))
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    if rank == 0:
        print(json.dumps({"metric": "depth frames/sec at 1216x352 (KITTI); achieved HBM GB/s vs peak", "value": B * world * args.steps / elapsed,
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "dry run (no GPU work)", "config": {"workload": "dry run", "frames_per_gpu_per_step": B}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="frames per GPU per step")
    ap.add_argument("--unique", type=int, default=32, help="distinct synthetic frames per GPU (tiled to --batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch1", action="store_true", help="also time batch=1 streamed launches (configs[1])")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the multi-rank protocol (gloo, no GPU work, value is meaningless): tests only")
    args = ap.parse_args()

    import numpy as np
    import torch
    from depth_completion_mt_amd import Context, make_params, synth
    from depth_completion_mt_amd import _lib as L

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available() or L.lib().dcmt_device_count() < 1:
        raise SystemExit("bench.py needs a gfx950 GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    B = args.batch
    uniq = min(args.unique, B)
    host = synth.synth_batch(uniq, ROWS, COLS, seed0=rank * 100003)
    d_uniq = torch.from_numpy(host).cuda()
    reps = (B + uniq - 1) // uniq
    d_src = d_uniq.repeat(reps, 1, 1)[:B].contiguous()
    d_dst = torch.empty_like(d_src)
    ctx = Context(local_rank, ROWS, COLS, B)
    params = make_params()                              # the reference's literals; 1 speculative loop application
    stream = torch.cuda.current_stream()

    def step():
        ctx.complete_dev(d_src, d_dst, params, stream=stream.cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gpu_ms_per_step = ev0.elapsed_time(ev1) / args.steps       # HIP events on the launch stream

    # every frame must have converged inside the timed configuration (no skipped work)
    iters, st = ctx.last_fill_iters(B)
    if st != L.OK:
        raise SystemExit("a frame needed more hole-closure applications than were enqueued: result invalid")

    if dist is not None:
        t = torch.tensor([elapsed, gpu_ms_per_step], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, gpu_ms_per_step = float(t[0]), float(t[1])

    extra = {}
    if rank == 0:
        # per-kernel split, live: the same batch with stop_after = column extension runs k_pre_s alone (same kernel, same
        # reads and writes; its output goes to d_dst instead of the scratch plane); k_fp_s = the step minus that.
        # Outside the timed region; cross-checks the rocprofv3 per-kernel averages in profiles/.
        pre_params = make_params(stop_after=L.STAGE_EXTEND)
        for _ in range(2):
            ctx.complete_dev(d_src, d_dst, pre_params, stream=stream.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.steps):
            ctx.complete_dev(d_src, d_dst, pre_params, stream=stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        pre_ms = e0.elapsed_time(e1) / args.steps
        extra["per_kernel_ms"] = {"k_pre_s": pre_ms, "k_fp_s_by_difference": gpu_ms_per_step - pre_ms}
    if args.batch1 and rank == 0:
        # BASELINE configs[1]: frames arrive one at a time (batch = 1 per call).  One context + stream per
        # in-flight frame; independent frames overlap on the GPU, each call is still a whole cascade on one frame.
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
        # BASELINE configs[1] as SURVEY.md section 8d words it: host-pinned input and output, frames arriving one at a time --
        # per frame an H2D copy, the cascade on that one frame, a D2H copy, all stream-ordered; four streams keep the PCIe
        # copies of one frame under the kernels of another
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
        # and the drop-in itself: dcmt_complete_f32 on pageable host arrays, one synchronous call per frame -- what the
        # cv::Mat shim does for the reference's own mains
        import ctypes
        hctx = Context(local_rank, ROWS, COLS, 1)
        hp = make_params()
        h_dst = np.empty((ROWS, COLS), np.float32)                 # the C entry point itself, on preallocated arrays
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
        # the same single-frame call captured once into a HIP graph and replayed: the entry point never synchronises or
        # allocates, so its memsets and kernel launches are capturable as they are; the replay removes the per-launch host cost
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

    if rank == 0:
        frames_total = B * world * args.steps
        value = frames_total / elapsed
        achieved = B * BYTES_PER_FRAME / (gpu_ms_per_step * 1e-3) / 1e9      # GB/s per GPU, whole chain
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")       # written by tools/pmc_traffic.py
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_step")
            except Exception:
                traffic = None
        line = {
            "metric": "depth frames/sec at 1216x352 (KITTI); achieved HBM GB/s vs peak",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": f"synthetic KITTI-like sparse depth (depth_completion_mt_amd/synth.py), {uniq} distinct frames per GPU tiled to {B}",
            "config": {"workload": f"DC_lidar_only img_completion, {COLS}x{ROWS} f32, device-resident batch of {B} frames per GPU per step "
                                   f"(BASELINE configs[4] sharding; configs[1] batch=1 is latency-bound, see --batch1)",
                       "frames_per_gpu_per_step": B, "rows": ROWS, "cols": COLS, "k0": "as_compiled", "blur": "gaussian",
                       "sharding": "per-frame, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "frac_of_measured_copy_ceiling": achieved / HBM_COPY_CEILING_GBS,
                         "gpu_ms_per_step": gpu_ms_per_step, "algorithmic_bytes_per_step": B * BYTES_PER_FRAME,
                         "kernel": "whole cascade per step = k_pre_s + k_fp_s (+ 3 skipped redo launches), HIP events around each step on the launch stream; live per-kernel split in per_kernel_ms, rocprofv3 averages in profiles/"},
            "fill_iters_max": max(iters),
        }
        line.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(min(16, len(os.sched_getaffinity(0))))   # the box's CPU share for one GPU is 16
        print(json.dumps(line))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
