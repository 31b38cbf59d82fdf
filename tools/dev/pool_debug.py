"""developer probe: where does the pooled trace differ from the oracle?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import lajolla_public_amd as lj
from helpers import Oracle, random_rays, scene_path
hs = lj.parse_scene(scene_path(sys.argv[1] if len(sys.argv) > 1 else "sponza"))
ctx = lj.Context(0); sc = lj.Scene(ctx, hs); o = Oracle(hs)
rays = random_rays(hs, 1 << 20, 11, o)
hg = lj.intersect(sc, rays["org"], rays["dir"], 0.0, np.inf); ho = o.intersect(rays)
bad = np.zeros(len(rays), bool)
for f in ("t", "u", "v", "shape_id", "prim_id"):
    d = hg[f].view(np.uint32) != ho[f].view(np.uint32)
    print(f, int(d.sum()))
    bad |= d
idx = np.nonzero(bad)[0]
print("bad", len(idx), "first", idx[:20], "lane", idx[:20] % 64)
for i in idx[:12]:
    print(i, "gpu", [hg[f][i] for f in ("t", "u", "v", "shape_id", "prim_id")], "oracle", [ho[f][i] for f in ("t", "u", "v", "shape_id", "prim_id")])
hg2 = lj.intersect(sc, rays["org"], rays["dir"], 0.0, np.inf)
print("repeatable:", all(np.array_equal(hg[f].view(np.uint32), hg2[f].view(np.uint32)) for f in ("t", "u", "v", "shape_id", "prim_id")))
