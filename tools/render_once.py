"""Render one scene N times on cuda:0 and print stats (used under rocprofv3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lajolla_public_amd as lj
scene = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "scenes/cbox/cbox.xml")
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
pool = int(sys.argv[5]) if len(sys.argv) > 5 else 0
hs = lj.parse_scene(scene)
ctx = lj.Context(0)
sc = lj.Scene(ctx, hs)
for i in range(reps):
    img = lj.render(sc, spp=spp, flags=flags, pool_paths=pool)
    s = sc.stats()
    print(f"spp={spp} device {s.render_ms:.2f} ms {s.samples/s.render_ms/1e3:.1f} Msamples/s K={s.bounce_iterations/s.samples:.3f} steps={s.wavefront_steps} "
          f"extend {s.extend_ms:.2f} ms shade {s.shade_ms:.2f} ms", flush=True)
