"""RCCL smoke test on one GPU: a one-rank process group with backend "nccl" (= RCCL on ROCm) and the collectives bench.py
issues at N > 1 — reduce(sum) of a framebuffer, barrier, all-reduce(max) of the step time."""
import os, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
frame = torch.full((512, 512, 3), 2.0, device="cuda")
dist.reduce(frame, dst=0, op=dist.ReduceOp.SUM); dist.barrier(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    dist.reduce(frame, dst=0, op=dist.ReduceOp.SUM)
dist.barrier(); torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("backend", dist.get_backend(), "| 10 x reduce(3 MB) + barrier: %.2f ms" % ms, "| frame mean", float(frame.mean()), "| max", float(t.item()))
dist.destroy_process_group()
