"""One-rank-per-GPU rendezvous exactly as bench.py does it for N > 1 (nccl = RCCL): init, barrier, all_reduce MAX.
Run under torch.distributed.run; with --nproc-per-node 1 it checks that the RCCL path of a box works at all."""
import os, torch, torch.distributed as dist
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
dist.barrier()
t = torch.tensor([float(rank + 1)], device="cuda", dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print(f"rank {rank}/{world}: all_reduce MAX = {t.item()}")
dist.destroy_process_group()
