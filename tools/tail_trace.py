"""One cbox render of rank 0's share at world size 8 (run under rocprofv3 --kernel-trace to see the per-step launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lajolla_public_amd as lj
hs = lj.parse_scene(os.path.join(ROOT, "scenes/cbox/cbox.xml"))
ctx = lj.Context(0); sc = lj.Scene(ctx, hs)
lj.render(sc, spp=256, rank=0, world_size=8)
lj.render(sc, spp=256, rank=0, world_size=8)
print(sc.stats().render_ms)
