"""Per-rank render time of the cbox bench workload for world sizes 1..8, measured on one GPU (rank 0's share)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lajolla_public_amd as lj
hs = lj.parse_scene(os.path.join(ROOT, "scenes/cbox/cbox.xml"))
ctx = lj.Context(0); sc = lj.Scene(ctx, hs)
full = None
for world in (1, 2, 4, 8):
    for rep in range(3):
        lj.render(sc, spp=256, rank=0, world_size=world)
    s = sc.stats()
    full = full or s.render_ms
    print(f"world {world}: device {s.render_ms:.2f} ms, steps {s.wavefront_steps}, samples {s.samples}, ideal {full/world:.2f} ms", flush=True)
