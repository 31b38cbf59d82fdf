"""Wall-clock per bench step of rank 0's share at world sizes 1..8, on one GPU (no reduce): what bench.py's timed loop sees."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lajolla_public_amd as lj
hs = lj.parse_scene(os.path.join(ROOT, "scenes/cbox/cbox.xml"))
ctx = lj.Context(0); sc = lj.Scene(ctx, hs)
frame = torch.zeros((hs.height, hs.width, 3), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream()
for world in (1, 2, 4, 8):
    for _ in range(2):
        lj.render_device(sc, frame.data_ptr(), stream=stream.cuda_stream, spp=256, rank=0, world_size=world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        lj.render_device(sc, frame.data_ptr(), stream=stream.cuda_stream, spp=256, rank=0, world_size=world)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    print(f"world {world}: wall {wall:.2f} ms/step, device {sc.stats().render_ms:.2f} ms, host overhead {wall - sc.stats().render_ms:.2f} ms", flush=True)
