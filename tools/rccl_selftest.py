"""RCCL smoke test on one GPU: the calls bench.py makes at N > 1 (init, reduce(sum) of a framebuffer, barrier, max over ranks),
with a one-rank group."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch
from lajolla_public_amd import dist as ljdist
torch.cuda.set_device(0)
ljdist.init_process_group("nccl")
frame = torch.full((512, 512, 3), 2.0, device="cuda")
t0 = time.perf_counter()
for _ in range(10):
    ljdist.reduce_framebuffer(frame, dst=0)
ljdist.barrier(); torch.cuda.synchronize()
print("reduce x10 + barrier: %.2f ms" % ((time.perf_counter() - t0) * 1e3), "frame", float(frame.mean()), "max_over_ranks", ljdist.max_over_ranks(1.5, device=torch.device("cuda", 0)))
