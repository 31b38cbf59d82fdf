"""The BVH8 extend path (extend8.hip: k_extend8 / k_trace_rays8, selected with LJ_TUNE_BVH8=1 for trees beyond the extend kernel's LDS
image) against the oracle and against the default BVH4 kernels: the closest hit is the minimum of (t, primitive id) over everything a
ray tests, so hit records — and with them every per-sample radiance — are bit-identical whatever the tree."""
import numpy as np
import pytest

import lajolla_public_amd as lj
from helpers import Oracle, random_rays, scene_path

pytestmark = pytest.mark.gpu

CROPS = {"disney_bsdf": (300, 200, 332, 232), "sponza": (300, 300, 332, 332)}


@pytest.fixture(scope="module")
def ctx():
    return lj.Context(0)


@pytest.mark.parametrize("name", ["disney_bsdf", "sponza"])
def test_bvh8_kernels_match_oracle_and_bvh4(name, ctx, monkeypatch):
    hs = lj.parse_scene(scene_path(name))
    o = Oracle(hs)
    monkeypatch.setenv("LJ_TUNE_BVH8", "0")   # read at upload (the default is by scene: BVH8 under an environment map, BVH4 otherwise)
    sc4 = lj.Scene(ctx, hs)
    monkeypatch.setenv("LJ_TUNE_BVH8", "1")   # this scene object walks the DNode8 tree
    sc8 = lj.Scene(ctx, hs)
    monkeypatch.delenv("LJ_TUNE_BVH8")
    rays = random_rays(hs, 1 << 19, 21, o)
    h8, ho = lj.intersect(sc8, rays["org"], rays["dir"], 0.0, np.inf), o.intersect(rays)
    assert (ho["shape_id"] >= 0).mean() > 0.2
    for f in ("t", "u", "v", "shape_id", "prim_id"):
        assert np.array_equal(h8[f].view(np.uint32), ho[f].view(np.uint32)), f
    r2 = random_rays(hs, 200000, 22, o)
    r2["tnear"] = np.float32(sc8.info.shadow_epsilon)
    r2["tfar"] = (np.random.default_rng(6).random(len(r2)) * sc8.info.bounds_radius).astype(np.float32)
    assert np.array_equal(lj.occluded(sc8, r2["org"], r2["dir"], r2["tnear"], r2["tfar"]), o.occluded(r2))
    # the render loop: per-sample radiance through k_extend8 equals the BVH4 kernel's bit for bit, and so do the ray counts
    p8 = lj.render_samples(sc8, CROPS[name], spp=8)
    s8 = sc8.stats()
    p4 = lj.render_samples(sc4, CROPS[name], spp=8)
    s4 = sc4.stats()
    assert np.array_equal(p8.view(np.uint32), p4.view(np.uint32))
    assert (s8.rays_closest, s8.rays_shadow, s8.bounce_iterations) == (s4.rays_closest, s4.rays_shadow, s4.bounce_iterations)
