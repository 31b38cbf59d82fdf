"""Edge cases of the render entry points on the GPU: degenerate crops, odd sample counts, tiny and huge path pools, rank
counts that do not divide the tiles (or exceed them), depth limits — every result against the oracle or against an
equivalent render, bit for bit where the quantity is deterministic."""
import numpy as np
import pytest

import lajolla_public_amd as lj
from lajolla_public_amd import _abi
from helpers import Oracle, scene_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cbox():
    hs = lj.parse_scene(scene_path("cbox"))
    return hs, lj.Scene(lj.Context(0), hs), Oracle(hs)


def test_single_pixel_and_border_crops(cbox):
    hs, sc, o = cbox
    full = lj.render(sc, spp=3)
    for crop in [(0, 0, 1, 1), (511, 511, 512, 512), (0, 511, 512, 512), (255, 0, 256, 512), (17, 33, 18, 34)]:
        x0, y0, x1, y1 = crop
        c = lj.render(sc, spp=3, crop=crop)
        same = bool(np.array_equal(c[y0:y1, x0:x1], full[y0:y1, x0:x1]))
        outside = c.copy(); outside[y0:y1, x0:x1] = 0
        assert same and not outside.any(), crop
        ps = lj.render_samples(sc, crop, spp=3)
        assert ps.shape == (y1 - y0, x1 - x0, 3, 3)
        rc, _, ref, _ = o.render(spp=3, crop=crop, per_sample=True)
        rel = np.abs(ps - ref).max(axis=-1) / np.maximum(np.abs(ref).max(axis=-1), 1e-3)
        assert np.median(rel) < 2e-5   # (a single pixel has three samples: not much of a median)


def test_pool_sizes_and_sample_counts_do_not_change_a_bit(cbox):
    hs, sc, o = cbox
    crop = (100, 100, 228, 164)
    for spp in (1, 2, 7, 33):
        a = lj.render(sc, spp=spp, crop=crop)
        for pool in (4096, 5000, 1 << 16, 1 << 20, 1 << 25):
            same = bool(np.array_equal(a, lj.render(sc, spp=spp, crop=crop, pool_paths=pool)))
            assert same, (spp, pool)


def test_rank_counts_that_do_not_divide_the_tiles(cbox):
    hs, sc, o = cbox
    a = lj.render(sc, spp=2)
    for world in (3, 7):
        acc = np.zeros_like(a)
        for r in range(world):
            acc += lj.render(sc, spp=2, rank=r, world_size=world)
        same = bool(np.array_equal(acc, a))
        assert same, world
    # more ranks than tiles: the ranks beyond the last tile render nothing
    assert not lj.render(sc, spp=2, rank=1500, world_size=2000).any()
    assert sc.stats().samples == 0
    with pytest.raises(lj.LajollaError) as e:
        lj.render(sc, spp=2, rank=2, world_size=2)
    assert e.value.code == _abi.LJ_ERR_INVALID_ARG


def test_depth_limits(cbox):
    hs, sc, o = cbox
    crop = (200, 20, 312, 84)   # the luminaire and the ceiling around it
    for md in (0, 1, 2, 5):
        ps = lj.render_samples(sc, crop, spp=2, max_depth=md)
        rc, _, ref, _ = o.render(spp=2, crop=crop, per_sample=True, max_depth=md)
        rel = np.abs(ps - ref).max(axis=-1) / np.maximum(np.abs(ref).max(axis=-1), 1e-3)
        assert np.isfinite(ps).all() and np.median(rel) < 2e-6 and (rel > 1e-3).mean() < 0.03, md
    # path_tracing.h:58-66: with max_depth 0 or 1 the bounce loop never runs — only directly visible emission is left
    d0, d1 = lj.render_samples(sc, crop, spp=2, max_depth=0), lj.render_samples(sc, crop, spp=2, max_depth=1)
    same = bool(np.array_equal(d0, d1))
    assert same and d0.max() > 1.0 and (d0 == 0).mean() > 0.5


def test_bad_arguments_are_refused(cbox):
    hs, sc, o = cbox
    for kw in (dict(crop=(-1, 0, 8, 8)), dict(crop=(0, 0, 513, 8)), dict(rank=-1, world_size=2)):
        with pytest.raises(lj.LajollaError) as e:
            lj.render(sc, **{"spp": 1, **kw})
        assert e.value.code == _abi.LJ_ERR_INVALID_ARG


def test_driver_loop_writes_what_render_returns(cbox, tmp_path, capsys):
    """main.cpp:12-51 — parse, render, imwrite to `-o`, the same messages."""
    from lajolla_public_amd.__main__ import main
    hs, sc, o = cbox
    out = tmp_path / "cbox.pfm"
    assert main(["-t", "4", "-o", str(out), "--spp", "2", scene_path("cbox")]) == 0
    said = capsys.readouterr().out
    assert "Parsing and constructing scene" in said and "Rendering..." in said and f"Image written to {out}" in said
    raw = out.read_bytes()
    head = b"PF\n512 512\n-1\n"
    same = raw[:len(head)] == head and raw[len(head):] == lj.render(sc, spp=2).tobytes()
    assert same


def test_fused_tail_and_lanes_do_not_change_a_bit(cbox, monkeypatch):
    """The end of a render runs fused (k_tail) and a render is cut into lanes (four from 4 M paths in flight, two from 1 M);
    neither may change a sample."""
    hs, sc, o = cbox
    spp = 17                                       # 4.46 M samples: four lanes by default
    a = lj.render(sc, spp=spp)
    assert sc.stats().samples == 512 * 512 * spp
    monkeypatch.setenv("LJ_TUNE_TAIL", "0")
    b = lj.render(sc, spp=spp)
    monkeypatch.setenv("LJ_TUNE_LANES", "1")
    c = lj.render(sc, spp=spp)
    monkeypatch.delenv("LJ_TUNE_TAIL")
    d = lj.render(sc, spp=spp)
    monkeypatch.setenv("LJ_TUNE_LANES", "3")
    e = lj.render(sc, spp=spp)
    monkeypatch.delenv("LJ_TUNE_LANES")
    monkeypatch.setenv("LJ_TUNE_TAIL_FRAC", "16")   # fuse from the moment every camera sample has been started
    f = lj.render(sc, spp=spp)
    g = lj.render(sc, spp=spp, pool_paths=1 << 21)  # two lanes
    same = [bool(np.array_equal(a, x)) for x in (b, c, d, e, f, g)]
    assert all(same), same


def test_scene_whose_light_tables_do_not_fit_in_lds(tmp_path):
    """An emitter of 1800 triangles: the per-workgroup LDS copy of the light tables is skipped (> 24 KiB) and the shade kernel
    reads them from global memory — same parity bars as everywhere."""
    n = 30
    v, f = [], []
    for j in range(n + 1):
        for i in range(n + 1):
            v.append("v %.6f 1.98 %.6f" % (-0.3 + 0.6 * i / n, -0.3 + 0.6 * j / n))
    for j in range(n):
        for i in range(n):
            a, b, c, d = j * (n + 1) + i + 1, j * (n + 1) + i + 2, (j + 1) * (n + 1) + i + 2, (j + 1) * (n + 1) + i + 1
            f += ["f %d %d %d" % (a, b, c), "f %d %d %d" % (a, c, d)]   # facing -y
    (tmp_path / "light.obj").write_text("\n".join(v + f) + "\n")
    (tmp_path / "floor.obj").write_text("v -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\nf 1 3 2\nf 1 4 3\n")
    xml = tmp_path / "biglight.xml"
    xml.write_text("""<scene version="0.6.0"><integrator type="path"><integer name="maxDepth" value="3"/></integrator>
      <sensor type="perspective"><float name="fov" value="60"/><transform name="toWorld"><lookat origin="0, 1, 3.2" target="0, 0.8, 0" up="0, 1, 0"/></transform>
        <sampler type="independent"><integer name="sampleCount" value="4"/></sampler>
        <film type="hdrfilm"><integer name="width" value="64"/><integer name="height" value="48"/></film></sensor>
      <shape type="obj"><string name="filename" value="floor.obj"/><bsdf type="diffuse"><rgb name="reflectance" value="0.6, 0.5, 0.4"/></bsdf></shape>
      <shape type="obj"><string name="filename" value="light.obj"/><bsdf type="diffuse"><rgb name="reflectance" value="0, 0, 0"/></bsdf>
        <emitter type="area"><rgb name="radiance" value="6, 5, 4"/></emitter></shape></scene>""")
    hs = lj.parse_scene(str(xml))
    sc, o = lj.Scene(lj.Context(0), hs), Oracle(hs)
    crop = (0, 0, 64, 48)
    ps = lj.render_samples(sc, crop, spp=4)
    rc, _, ref, _ = o.render(spp=4, crop=crop, per_sample=True)
    assert rc == 0 and ref.max() > 1.0
    rel = np.abs(ps - ref).max(axis=-1) / np.maximum(np.abs(ref).max(axis=-1), 1e-3)
    assert np.isfinite(ps).all() and np.median(rel) < 2e-6 and (rel > 1e-3).mean() < 0.02
    assert abs(ps.mean() / ref.mean() - 1) < 2e-4


def test_render_cut_into_passes(cbox):
    """More than 2^27 samples are rendered in several passes over disjoint pixel ranges; the two rank shares of the same
    render fit one pass each, so their sum pins the multi-pass frame bit for bit."""
    hs, sc, o = cbox
    spp = 520                                       # 136 M samples > 2^27
    whole = lj.render(sc, spp=spp)
    assert sc.stats().samples == 512 * 512 * spp
    parts = lj.render(sc, spp=spp, rank=0, world_size=2) + lj.render(sc, spp=spp, rank=1, world_size=2)
    same = bool(np.array_equal(whole, parts))
    assert same and bool(np.isfinite(whole).all())
