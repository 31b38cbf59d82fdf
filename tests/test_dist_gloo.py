"""world_size-2 test of the multi-GPU layer on CPU (gloo): tile sharding + framebuffer sum-reduce + max-over-ranks timing,
i.e. exactly the torch.distributed calls bench.py makes, with the per-rank framebuffers produced by the CPU oracle
(no GPU here).  The reduced image on rank 0 must be bit-identical to a single-rank render."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import lajolla_public_amd as lj
from lajolla_public_amd import dist as ljdist
from helpers import Oracle, scene_path
rank, world = ljdist.init_process_group("gloo")
hs = lj.parse_scene(scene_path("cbox"))
o = Oracle(hs)
crop = (160, 160, 288, 240)
_, part, _, _ = o.render(spp=2, rng_mode=0, crop=crop, rank=rank, world_size=world, threads=2)
frame = torch.from_numpy(part.astype(np.float32))
ljdist.barrier()
ljdist.reduce_framebuffer(frame, dst=0)
t = ljdist.max_over_ranks(1.0 + rank)
assert t == float(world), t
if rank == 0:
    _, full, _, _ = o.render(spp=2, rng_mode=0, crop=crop, threads=2)
    assert np.array_equal(frame.numpy(), full.astype(np.float32)), "reduced framebuffer differs from the single-rank image"
    print("GLOO_OK", float(frame.sum()))
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_tile_sharding_and_reduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=300)
        outs.append(out)
        assert p.returncode == 0, out
    assert "GLOO_OK" in outs[0]
