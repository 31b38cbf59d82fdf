"""Per-rank device time of one scene at world size N (each rank's share rendered alone on this GPU): load balance of the
tile -> rank assignment (tile t goes to rank t mod N)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lajolla_public_amd as lj
scene = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "scenes/sponza/sponza.xml")
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
hs = lj.parse_scene(scene)
ctx = lj.Context(0); sc = lj.Scene(ctx, hs)
lj.render(sc, spp=spp, rank=0, world_size=world)
ms = []
for r in range(world):
    lj.render(sc, spp=spp, rank=r, world_size=world)
    ms.append(sc.stats().render_ms)
lj.render(sc, spp=spp)
whole = sc.stats().render_ms
print(f"{os.path.basename(scene)} spp={spp} world={world}: per-rank ms " + " ".join(f"{m:.1f}" for m in ms) + f" | max {max(ms):.1f} mean {sum(ms)/world:.1f} whole/N {whole/world:.1f} -> efficiency {whole/world/max(ms):.2f}", flush=True)
