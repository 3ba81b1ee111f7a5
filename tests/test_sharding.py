"""Multi-GPU path on CPU: the per-frame partition and the bench's barrier / max-over-ranks
protocol, with world_size 2 over gloo (no GPU, no data-path collective to test: there is none)."""
import os
import socket
import subprocess
import time
import sys
import textwrap

import pytest

from conftest import ROOT
from depth_completion_mt_amd import sharding


def test_shard_ranges_cover_every_frame_once():
    for n in (0, 1, 7, 8, 1023, 1024, 1025):
        for w in (1, 2, 3, 4, 8):
            sh = sharding.all_shards(n, w)
            assert sh[0][0] == 0 and sh[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
            sizes = [e - b for b, e in sh]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.all_shards(1024, 8) == [(i * 128, (i + 1) * 128) for i in range(8)]   # BASELINE configs[4]
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)


def test_job_throughput_uses_the_slowest_rank():
    assert sharding.job_throughput([128, 128], [1.0, 2.0]) == 128.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_protocol(tmp_path):
    """Two processes: each takes its shard of 11 frames, 'processes' it (a checksum of the frame
    indices stands in for the GPU work, which needs a GPU), then the ranks agree on the slowest
    time with an all_reduce(MAX) and rank 0 checks coverage with an all_gather of the ranges --
    the same collectives bench.py uses around (never inside) the data path."""
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, time
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from depth_completion_mt_amd import sharding
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        b, e = sharding.shard_range(11, rank, world)
        dist.barrier()
        t0 = time.perf_counter()
        work = sum(range(b, e))
        time.sleep(0.05 * (rank + 1))
        dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        ranges = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(ranges, torch.tensor([b, e, work]))
        if rank == 0:
            assert [int(r[0]) for r in ranges] == [0, 6] and [int(r[1]) for r in ranges] == [6, 11]
            assert sum(int(r[2]) for r in ranges) == sum(range(11))
            assert float(el[0]) >= 0.1
            print("OK", float(el[0]))
        dist.destroy_process_group()
    """))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0][0]


def _bench(nproc, *flags, timeout=300):
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")]
    return subprocess.run(cmd + list(flags), capture_output=True, text=True, timeout=timeout)


def test_bench_rank_protocol_two_ranks_gloo():
    """bench.py's own multi-rank plumbing (env parsing, sharding of --total-frames, barriers, max over ranks, ONE JSON line from
    rank 0) launched exactly as the driver launches it, with --dry-run replacing the GPU step."""
    import json
    # the default: BASELINE configs[4] as written -- 1024 frames in total, sharded per frame (strong scaling)
    r = _bench(2, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "strong"
    assert d["config"]["total_frames_per_step"] == 1024 and d["config"]["frames_per_gpu_per_step"] == 512
    assert d["ms_per_step"] >= 4.0          # the slower rank (2 x 2 ms per step) sets the time
    assert abs(d["value"] - 1024 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]      # whole-job frames over the slowest rank's time
    # an uneven shard: 1001 frames on 2 ranks -> rank 0 owns 501
    d = json.loads([l for l in _bench(2, "--gpus", "2", "--steps", "2", "--warmup", "1", "--total-frames", "1001", "--dry-run").stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["frames_per_gpu_per_step"] == 501 and d["config"]["total_frames_per_step"] == 1001
    # weak scaling on request, and labelled as such
    r = _bench(2, "--gpus", "2", "--steps", "3", "--warmup", "1", "--weak", "--batch", "8", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "weak" and d["config"]["frames_per_gpu_per_step"] == 8 and d["config"]["total_frames_per_step"] == 16
    # a mismatch between --gpus and the launched world size is an error, not a silent single-GPU run
    assert _bench(1, "--gpus", "2", "--dry-run", timeout=120).returncode != 0
    # more ranks than frames is refused
    assert _bench(1, "--gpus", "1", "--total-frames", "0", "--dry-run", timeout=120).returncode != 0


def test_bench_failing_rank_ends_every_rank():
    """A rank whose step failed (a frame that did not converge) must not leave the others waiting in a collective: the status
    travels in the same all-reduce as the times, every rank exits non-zero, and no JSON line is printed."""
    t0 = time.time()
    r = _bench(2, "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", "--dry-run-fail-rank", "1", timeout=240)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert time.time() - t0 < 120
