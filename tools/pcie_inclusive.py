"""Wall-clock of lj.render() — the host-framebuffer entry point lj_render: render + resolve + device-to-host copy of the frame —
beside the device time of the same renders (what bench.py's `value` is computed from).  cbox 512x512 @ 256 spp."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lajolla_public_amd as lj
hs = lj.parse_scene(os.path.join(ROOT, "scenes/cbox/cbox.xml"))
ctx = lj.Context(0); sc = lj.Scene(ctx, hs)
for _ in range(2):
    lj.render(sc, spp=256)
n, dev = 10, 0.0
t0 = time.perf_counter()
for _ in range(n):
    img = lj.render(sc, spp=256)
    dev += sc.stats().render_ms
wall = (time.perf_counter() - t0) * 1e3 / n
samples = hs.width * hs.height * 256
print(f"lj_render (host framebuffer, {img.nbytes / 1e6:.1f} MB over PCIe): {wall:.2f} ms per call = {samples / wall / 1e3:.0f} Msamples/s; device time of the same renders {dev / n:.2f} ms")
