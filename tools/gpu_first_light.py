"""First-light check on a real MI355X: traversal parity, per-sample parity, a timed cbox render."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lajolla_public_amd as lj  # noqa: E402
from helpers import Oracle, random_rays, scene_path  # noqa: E402


def main():
    hs = lj.parse_scene(scene_path("cbox"))
    ctx = lj.Context(0)
    sc = lj.Scene(ctx, hs)
    print("scene info: tris", sc.info.n_triangles, "nodes", sc.info.n_bvh_nodes, "eps", sc.info.shadow_epsilon, flush=True)
    o = Oracle(hs)
    rays = random_rays(hs, 200000, 1, o)
    hg = lj.intersect(sc, rays["org"], rays["dir"], 0.0, np.inf)
    ho = o.intersect(rays)
    for f in ("t", "u", "v", "shape_id", "prim_id"):
        a, b = hg[f], ho[f]
        mism = (a.view(np.uint32) != b.view(np.uint32)).sum()
        print("intersect", f, "bit mismatches:", int(mism), flush=True)
    crop = (200, 200, 232, 232)
    spp = 16
    rc, rgb, ps, st = o.render(spp=spp, rng_mode=0, crop=crop, per_sample=True)
    pg = lj.render_samples(sc, crop, spp=spp)
    diff = np.abs(pg - ps).max(axis=-1)
    scale = np.maximum(np.abs(ps).max(axis=-1), 1e-3)
    rel = diff / scale
    print("per-sample: frac rel>1e-3", (rel > 1e-3).mean(), "median", np.median(rel), "mean ratio", pg.mean() / ps.mean(), flush=True)
    for spp in (16, 64, 256):
        for it in range(2):
            t = time.time()
            img = lj.render(sc, spp=spp)
            dt = time.time() - t
            s = sc.stats()
            print(f"render spp={spp}: wall {dt*1e3:.1f} ms, device {s.render_ms:.1f} ms, {s.samples/ s.render_ms/1e3:.1f} Msamples/s, "
                  f"K={s.bounce_iterations/s.samples:.3f}, steps={s.wavefront_steps}, mean={img.mean(axis=(0,1))}", flush=True)
    img2 = lj.render(sc, spp=256)
    print("deterministic:", bool((img == img2).all()), flush=True)
    np.save(os.path.join(ROOT, "gpurun_out", "cbox_256.npy"), img)


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    main()
