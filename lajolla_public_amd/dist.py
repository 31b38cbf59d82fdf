"""Multi-GPU layer: one process per GPU, image tiles sharded across ranks, one RCCL reduce of the framebuffer.

The reference's only parallelism is its thread pool over 16x16 tiles (render.cpp:75-98, parallel.cpp:183-237); tiles are
independent (per-tile RNG stream, disjoint pixel writes).  Here rank r renders the tiles t = ty*ntx + tx with
t % world_size == r at full spp into a zero-initialised full frame; every pixel then has exactly one non-zero
contributor, so a sum-reduce over ranks is exact and order independent (bit-identical to a 1-GPU render).
PyTorch is used for the process group only (backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests).
"""
import os

import numpy as np

TILE = 16  # render.cpp:75


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend):
    import torch.distributed as dist
    rank, world, _ = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def tile_owner_mask(width, height, rank, world_size):
    """Boolean (h, w) mask of the pixels rank `rank` renders: tiles t = ty*ntx+tx with t % world_size == rank."""
    ntx = (width + TILE - 1) // TILE
    ys, xs = np.mgrid[0:height, 0:width]
    t = (ys // TILE) * ntx + (xs // TILE)
    return (t % world_size) == rank


def reduce_framebuffer(frame, dst=0):
    """Sum-reduce a (h, w, 3) tensor over ranks onto `dst` (in place on dst).  No-op for a single process."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM)
    return frame


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device=None):
    """All-reduce MAX of a python float (the bench's per-step time)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
