"""CPU-side parity of the *device* shading code (device/dshade.h + device/dtrace.h compiled for the host, float)
against the double-precision oracle, path by path under the same per-(pixel, sample) pcg32 streams.

Tolerances (also DESIGN.md §6): float shading follows the double oracle until a discrete decision flips (a light /
triangle choice, Russian roulette, a shadow ray at a grazing self-intersection guard).  So
  * the MEDIAN per-sample relative difference must be at float round-off level (< 2e-6),
  * at most 2 % of the samples may differ by more than 1e-3 (diverged paths — still valid samples of the same
    estimator),
  * per-crop mean radiance must agree to 2e-4 relative,
  * the per-pixel relative L2 over a crop at 16 spp must be < 1e-2 (it falls as 1/sqrt(spp))."""
import numpy as np
import pytest

import lajolla_public_amd as lj
from helpers import Oracle, Twin, scene_path

CASES = [("cbox", (200, 200, 232, 232), None), ("cbox", (0, 0, 48, 32), None), ("cbox", (300, 40, 332, 72), 3),
         ("veach_mi", (300, 200, 348, 232), None), ("veach_mi", (100, 380, 132, 412), None)]


@pytest.mark.parametrize("name,crop,max_depth", CASES)
def test_per_sample_parity(name, crop, max_depth):
    hs = lj.parse_scene(scene_path(name))
    o, tw = Oracle(hs), Twin(hs)
    spp = 16
    rc, _, ps, st = o.render(spp=spp, rng_mode=0, crop=crop, per_sample=True, max_depth=max_depth)
    assert rc == 0
    pt, bounces = tw.render_samples(crop, spp, max_depth=max_depth)
    assert np.isfinite(pt).all() and (pt >= 0).all()
    diff = np.abs(pt - ps).max(axis=-1)
    scale = np.maximum(np.abs(ps).max(axis=-1), 1e-3)
    rel = diff / scale
    assert np.median(rel) < 2e-6
    assert (rel > 1e-3).mean() < 0.02
    assert abs(pt.mean() / ps.mean() - 1) < 2e-4
    pix_o, pix_t = ps.mean(axis=2), pt.mean(axis=2)
    assert np.linalg.norm(pix_o - pix_t) / np.linalg.norm(pix_o) < 1e-2
    # the bounce-iteration count K agrees to within the diverged paths
    assert abs(bounces / st.bounces - 1) < 5e-3


def test_max_depth_semantics():
    """path_tracing.h:66 — `num_vertices <= max_depth + 1`; max_depth 1 leaves only directly visible emission."""
    hs = lj.parse_scene(scene_path("cbox"))
    o, tw = Oracle(hs), Twin(hs)
    crop = (224, 40, 288, 72)  # the luminaire
    _, _, ps, _ = o.render(spp=4, rng_mode=0, crop=crop, per_sample=True, max_depth=1)
    pt, b = tw.render_samples(crop, 4, max_depth=1)
    assert b == 0
    assert np.allclose(pt, ps, rtol=1e-5, atol=1e-7)
    assert ps.max() > 1.0  # emitter seen directly
