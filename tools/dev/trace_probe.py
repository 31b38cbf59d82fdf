"""Does a batch of probe rays through lj_intersect rank the BVH4 and BVH8 traversal kernels the way a render does?  (device ms of the query kernel)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lajolla_public_amd as lj
for scene in sys.argv[1:]:
    hs = lj.parse_scene(scene)
    ctx = lj.Context(0)
    res = {}
    for wide in ("0", "1"):
        os.environ["LJ_TUNE_BVH8"] = wide
        sc = lj.Scene(ctx, hs)
        c, r = np.array(sc.info.bounds_center), sc.info.bounds_radius
        rng = np.random.default_rng(1)
        for n in (1 << 16, 1 << 18, 1 << 20):
            org = (c + (rng.random((n, 3)) - 0.5) * r).astype(np.float32)
            d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
            best = 1e9
            for rep in range(3):
                lj.intersect(sc, org, d.astype(np.float32), 1e-4, np.inf)
                best = min(best, sc.stats().render_ms)
            res[(wide, n)] = best
    print(os.path.basename(scene), {f"bvh8={k[0]} n={k[1]}": round(v, 3) for k, v in res.items()})
