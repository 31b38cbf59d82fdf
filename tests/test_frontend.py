"""Front-end parity: our Mitsuba-XML / OBJ / .serialized loaders against what the reference's own
parse_scene.cpp / parse_obj.cpp / load_serialized.cpp produced for the same files (tests/golden/scene_*.json).
Tolerance 1e-12 relative — in practice the values are bit-identical because the number semantics (std::stof,
reciprocal-multiply division, transform composition order) are reproduced."""
import os

import numpy as np
import pytest

import lajolla_public_amd as lj
from lajolla_public_amd import _abi
from helpers import golden, scene_path, ROOT


def close(a, b, rel=1e-12, abs_=1e-300):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return a.shape == b.shape and np.all(np.abs(a - b) <= rel * np.maximum(np.abs(a), np.abs(b)) + abs_)


TEX_KIND = {"constant": 0, "image": 1, "checkerboard": 2}


def check_texture(t, g, spectrum):
    assert t.kind == TEX_KIND[g["kind"]]
    if g["kind"] == "constant":
        v = g["value"] if spectrum else [g["value"]] * 3
        assert close(list(t.value), v)
    elif g["kind"] == "image":
        assert t.texture_id == g["texture_id"]
        assert close([t.uscale, t.vscale, t.uoffset, t.voffset], [g["uscale"], g["vscale"], g["uoffset"], g["voffset"]])
    else:
        assert close(list(t.value), g["color0"]) and close(list(t.color1), g["color1"])
        assert close([t.uscale, t.vscale, t.uoffset, t.voffset], [g["uscale"], g["vscale"], g["uoffset"], g["voffset"]])


def check_scene(name, image_scenes=False):
    g = golden("scene_" + name)
    hs = lj.parse_scene(scene_path(name))
    d = hs.desc
    # RenderOptions (parse_scene.cpp:265-309) and Camera (camera.cpp:7-21, parse_scene.cpp:459-556)
    o = g["options"]
    assert (d.options.integrator, d.options.samples_per_pixel, d.options.max_depth, d.options.rr_depth) == \
        (o["integrator"], o["samples_per_pixel"], o["max_depth"], o["rr_depth"])
    c = g["camera"]
    assert (d.camera.width, d.camera.height) == (c["width"], c["height"])
    for k in ("cam_to_world", "world_to_cam", "sample_to_cam", "cam_to_sample"):
        assert close(list(getattr(d.camera, k)), c[k], abs_=1e-18), k
    assert d.camera.filter_kind == {"box": 0, "tent": 1, "gaussian": 2}[c["filter"]] and close(d.camera.filter_param, c["filter_param"])
    # materials, in parse order (parse_scene.cpp:558-809)
    assert d.n_materials == len(g["materials"])
    for i, gm in enumerate(g["materials"]):
        m = d.materials[i]
        assert _abi.MATERIAL_KINDS[m.kind] == gm["kind"]
        slots = _abi.MATERIAL_SLOTS[gm["kind"]]
        assert m.n_tex == len(slots)
        for s, sname in enumerate(slots):
            check_texture(m.tex[s], gm[sname], sname in _abi.SPECTRUM_SLOTS)
        if "eta" in gm:
            assert close(m.eta, gm["eta"])
    # shapes (parse_scene.cpp:811-970, parse_obj.cpp, load_serialized.cpp)
    assert d.n_shapes == len(g["shapes"])
    P, N, UV, I = hs.positions(), hs.normals(), hs.uvs(), hs.indices()
    for i, gs in enumerate(g["shapes"]):
        s = d.shapes[i]
        assert (s.material_id, s.area_light_id) == (gs["material_id"], gs["area_light_id"])
        if gs["kind"] == "sphere":
            assert s.kind == _abi.LJ_SHAPE_SPHERE and close(list(s.position), gs["position"]) and close(s.radius, gs["radius"])
            continue
        assert s.kind == _abi.LJ_SHAPE_TRIMESH
        assert (s.n_vertices, s.n_triangles) == (gs["n_positions"], gs["n_indices"])
        assert bool(s.has_normals) == (gs["n_normals"] > 0) and bool(s.has_uvs) == (gs["n_uvs"] > 0)
        p = P[s.first_vertex:s.first_vertex + s.n_vertices]
        idx = I[s.first_triangle:s.first_triangle + s.n_triangles]
        assert close(p.sum(axis=0), gs["sum_positions"], rel=1e-10)
        vs, ts = gs["vertex_stride"], gs["index_stride"]
        assert close(p[::vs], gs["positions"])
        assert idx[::ts].tolist() == gs["indices"]
        w = (np.arange(len(idx)) % 7 + 1)[:, None] * idx.astype(np.int64) * np.array([1, 2, 3])
        assert int(w.sum()) == gs["index_checksum"]
        if gs["n_normals"]:
            n = N[s.first_vertex:s.first_vertex + s.n_vertices]
            assert close(n[::vs], gs["normals"], abs_=1e-15) and close(n.sum(axis=0), gs["sum_normals"], rel=1e-9, abs_=1e-9)
        if gs["n_uvs"]:
            uv = UV[s.first_vertex:s.first_vertex + s.n_vertices]
            assert close(uv[::vs], gs["uvs"]) and close(uv.sum(axis=0), gs["sum_uvs"], rel=1e-9)
    # lights, in parse order (parse_scene.cpp:935-966, 1084-1111)
    assert d.n_lights == len(g["lights"]) and d.envmap_light_id == g["envmap_light_id"]
    for i, gl in enumerate(g["lights"]):
        l = d.lights[i]
        if gl["kind"] == "area":
            assert l.kind == _abi.LJ_LIGHT_AREA and l.shape_id == gl["shape_id"] and close(list(l.intensity), gl["intensity"])
        else:
            assert l.kind == _abi.LJ_LIGHT_ENVMAP and close(list(l.to_world), gl["to_world"], abs_=1e-18) and close(list(l.to_local), gl["to_local"], abs_=1e-18)
            assert close(l.scale, gl["scale"])
    return hs, g


def test_cbox_matches_reference_parser():
    hs, g = check_scene("cbox")
    assert hs.desc.n_triangles == 38 and hs.desc.n_shapes == 8


def test_veach_mi_matches_reference_parser():
    hs, g = check_scene("veach_mi")
    assert hs.desc.options.max_depth == 2  # integrator type="direct" (parse_scene.cpp:292-294)
    assert sum(1 for i in range(hs.desc.n_shapes) if hs.desc.shapes[i].kind == _abi.LJ_SHAPE_SPHERE) == 5


def test_disney_bsdf_scene_matches_reference_parser():
    """Mitsuba .serialized v3 meshes (load_serialized.cpp:174-256), DisneyBSDF + checkerboard, rotated envmap."""
    hs, g = check_scene("disney_bsdf")
    assert hs.desc.n_triangles == 61600 and hs.desc.envmap_light_id == 0
    im = hs.desc.images3[0]   # envmap level 0: OpenEXR, HALF channels, PIZ compression (exr_decode.cpp)
    arr = np.ctypeslib.as_array(im.data, shape=(im.height, im.width, 3)).astype(np.float64)
    assert (im.width, im.height) == tuple(g["image3s"][0]["dims"][0])
    assert close(arr.sum(axis=(0, 1)), g["image3s"][0]["level_sums"][0], rel=1e-12)
    # ... and texel for texel against the PFM the reference's own loader (tinyexr via imread3) wrote of the same file
    # (oracle/convert_assets.cpp): HALF -> float is exact, so the two must be bit-identical
    with open(os.path.join(ROOT, "tests", "golden", "matpreview_envmap_reference_decode.pfm"), "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        ref = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, 3)[::-1]   # PFM is bottom-up
    got = np.ctypeslib.as_array(im.data, shape=(im.height, im.width, 3))
    assert (w, h) == (im.width, im.height)
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(ref, dtype=np.float32).view(np.uint32))


def test_sponza_matches_reference_parser_and_jpeg_decoder():
    """37 sub-meshes of one .serialized file + ten baseline JPEGs: our decoder must reproduce stb_image's texels
    (then pow(v/255, 2.2), image.cpp:96 -> stb_image.h:1849) bit for bit."""
    hs, g = check_scene("sponza")
    assert hs.desc.n_triangles == 66445 and hs.desc.n_images3 == 10
    for i, gi in enumerate(g["image3s"]):
        im = hs.desc.images3[i]
        arr = np.ctypeslib.as_array(im.data, shape=(im.height, im.width, 3)).astype(np.float64)
        assert [im.width, im.height] == gi["dims"][0]
        assert np.array_equal(arr.sum(axis=(0, 1)), np.array(gi["level_sums"][0]))
        st = max(1, (im.width * im.height) // 32)
        assert np.array_equal(arr.reshape(-1, 3)[::st], np.array(gi["texels0"]))


def test_error_behaviour(tmp_path):
    """The reference throws fl_exception via Error() (flexception.h:8-24); the C ABI turns each site into a code."""
    with pytest.raises(lj.LajollaError) as e:
        lj.parse_scene(str(tmp_path / "missing.xml"))
    assert e.value.code == _abi.LJ_ERR_IO
    bad = tmp_path / "bad.xml"
    bad.write_text("<scene><integrator type='path'></scene>")
    with pytest.raises(lj.LajollaError) as e:
        lj.parse_scene(str(bad))
    assert e.value.code == _abi.LJ_ERR_PARSE
    for body, frag in [("<integrator type='bdpt'/>", "Unsupported integrator"),
                       ("<bsdf type='phong' id='x'/>", "Unknown BSDF"),
                       ("<shape type='cube'><bsdf type='diffuse'/></shape>", "Unknown shape"),
                       ("<shape type='obj'><string name='filename' value='nope.obj'/><bsdf type='diffuse'/></shape>", "obj"),
                       ("<shape type='sphere'><ref id='nomat'/></shape>", "not found"),
                       ("<emitter type='point'/>", "Unknown emitter"),
                       ("<bsdf type='diffuse' id='a'><rgb name='reflectance' value='1 2'/></bsdf>", "parse_vector3")]:
        f = tmp_path / "s.xml"
        f.write_text(f"<?xml version='1.0'?><scene version='0.4.0'>{body}</scene>")
        with pytest.raises(lj.LajollaError) as e:
            lj.parse_scene(str(f))
        assert frag in str(e.value), (body, str(e.value))


def test_parser_semantics(tmp_path):
    """Quirks that change pixels: std::stof rounding, `new * accumulated` transform order, integrator-after-sensor
    resetting spp, anonymous top-level bsdf dropped, one-entry spectrum white for bsdfs / whitepoint for emitters."""
    f = tmp_path / "q.xml"
    f.write_text("""<scene version="0.4.0">
      <sensor type="perspective"><float name="fov" value="0.1"/>
        <transform name="toWorld"><translate x="1" y="2" z="3"/><scale x="2"/><rotate y="1" angle="90"/></transform>
        <sampler type="independent"><integer name="sampleCount" value="77"/></sampler>
        <film type="hdrfilm"><integer name="width" value="30"/><integer name="height" value="20"/></film></sensor>
      <integrator type="path"><integer name="maxDepth" value="7"/></integrator>
      <bsdf type="diffuse"><rgb name="reflectance" value="0.1"/></bsdf>
      <bsdf type="diffuse" id="m"><spectrum name="reflectance" value="0.3"/></bsdf>
      <shape type="sphere"><point name="center" x="0.1" y="0" z="0"/><float name="radius" value="0.7"/><ref id="m"/>
        <emitter type="area"><spectrum name="radiance" value="2"/></emitter></shape>
    </scene>""")
    hs = lj.parse_scene(str(f))
    d = hs.desc
    assert d.options.samples_per_pixel == 4 and d.options.max_depth == 7   # integrator parsed after the sensor
    assert d.n_materials == 1                                              # the anonymous bsdf is dropped
    assert list(d.materials[0].tex[0].value) == [1.0, 1.0, 1.0]            # spectrum "0.3" -> white
    assert d.shapes[0].position[0] == float(np.float32(0.1)) and d.shapes[0].radius == float(np.float32(0.7))
    xyz = np.array([0.9505, 1.0, 1.0888]) * 2.0
    rgb = [3.240479 * xyz[0] - 1.537150 * xyz[1] - 0.498535 * xyz[2], -0.969256 * xyz[0] + 1.875991 * xyz[1] + 0.041556 * xyz[2],
           0.055648 * xyz[0] - 0.204043 * xyz[1] + 1.057311 * xyz[2]]
    assert close(list(d.lights[0].intensity), rgb)
    m = np.array(list(d.camera.cam_to_world)).reshape(4, 4)
    # rotate * scale * translate applied to the origin: translate first, then scale x, then rotate about y by 90 deg
    p = m @ np.array([0, 0, 0, 1.0])
    assert np.allclose(p[:3], [3.0, 2.0, -2.0], atol=1e-6)
    assert d.camera.width == 30 and d.camera.height == 20


def test_obj_loader_quads_and_synthesised_normals(tmp_path):
    """parse_obj.cpp:137-234: quad fan (0,1,2),(0,2,3); vt stored as (s, 1-t); angle-weighted normals when absent."""
    (tmp_path / "q.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 0.25\nf 1/1 2/2 3/3 4/4\n")
    (tmp_path / "s.xml").write_text("<scene version='0.4.0'><shape type='obj'><string name='filename' value='q.obj'/><bsdf type='diffuse'/></shape></scene>")
    hs = lj.parse_scene(str(tmp_path / "s.xml"))
    assert hs.indices().tolist() == [[0, 1, 2], [0, 2, 3]]
    assert np.allclose(hs.uvs(), [[0, 1], [1, 1], [1, 0], [0, 0.75]])
    assert np.allclose(hs.normals(), [[0, 0, 1]] * 4)
    assert hs.desc.shapes[0].has_normals == 1 and hs.desc.shapes[0].has_uvs == 1


def test_image_writers_round_trip(tmp_path):
    """imwrite (image.cpp:135-173).  PFM: the reference's header and top-down float rows, byte for byte.  EXR: HALF
    scan-line file that our own reader (and any OpenEXR reader) takes back; values are the nearest halfs."""
    rng = np.random.default_rng(3)
    img = (rng.random((37, 53, 3)) * 4).astype(np.float32)
    img[0, 0] = [0.0, 1e-7, 70000.0]   # zero, a half subnormal, overflow to +inf
    pfm = tmp_path / "out.pfm"
    lj.write_image(str(pfm), img)
    raw = pfm.read_bytes()
    head = b"PF\n53 37\n-1\n"
    assert raw[:len(head)] == head and raw[len(head):] == img.tobytes()
    exr = tmp_path / "env.exr"
    lj.write_image(str(exr), img)
    scene = tmp_path / "s.xml"
    scene.write_text(f"""<scene version="0.6.0"><integrator type="path"/><sensor type="perspective"><film type="hdrfilm">
        <integer name="width" value="8"/><integer name="height" value="8"/></film></sensor>
        <emitter type="envmap"><string name="filename" value="{exr}"/></emitter>
        <shape type="sphere"><bsdf type="diffuse"/></shape></scene>""")
    hs = lj.parse_scene(str(scene))
    im = hs.desc.images3[0]
    got = np.ctypeslib.as_array(im.data, shape=(im.height, im.width, 3))
    assert (im.width, im.height) == (53, 37)
    with np.errstate(over="ignore"):
        assert np.array_equal(got, img.astype(np.float16).astype(np.float32))
    with pytest.raises(lj.LajollaError) as e:
        lj.write_image(str(tmp_path / "out.png"), img)
    assert e.value.code == _abi.LJ_ERR_UNSUPPORTED


@pytest.mark.parametrize("name", ["hetvol", "hetvol_colored", "vol_cbox_teapot", "volpath_test6", "volpath_test5", "volpath_test1"])
def test_media_match_reference_parser(name):
    """parse_medium / parse_phase_function / parse_volume_spectrum / load_volume (parse_scene.cpp:359-457, volume.cpp:6-104):
    media, their order, the shapes' interior / exterior references and the grid volumes' voxels, against the reference's
    own parser (tests/golden/media.json)."""
    g = golden("media")["scenes"][name]
    hs = lj.parse_scene(os.path.join(ROOT, "scenes", "volpath_test", name + ".xml"))
    d = hs.desc
    assert d.camera.medium_id == g["camera_medium_id"]
    assert (d.options.integrator, d.options.vol_path_version, d.options.max_null_collisions, d.options.max_depth, d.options.rr_depth) == \
           (g["integrator"], g["vol_path_version"], g["max_null_collisions"], g["max_depth"], g["rr_depth"])
    assert [[d.shapes[i].material_id, d.shapes[i].interior_medium_id, d.shapes[i].exterior_medium_id] for i in range(d.n_shapes)] == g["shape_media"]
    assert d.n_media == len(g["media"])
    for i, gm in enumerate(g["media"]):
        m = d.media[i]
        assert m.kind == (_abi.LJ_MEDIUM_HOMOGENEOUS if gm["kind"] == "homogeneous" else _abi.LJ_MEDIUM_HETEROGENEOUS)
        assert m.phase_kind == (_abi.LJ_PHASE_HG if gm["phase"] == "hg" else _abi.LJ_PHASE_ISOTROPIC) and m.g == gm["g"]
        if gm["kind"] == "homogeneous":
            assert list(m.sigma_a) == gm["sigma_a"] and list(m.sigma_s) == gm["sigma_s"]
            continue
        for vol, gv in ((m.albedo, gm["albedo"]), (m.density, gm["density"])):
            if gv["kind"] == "constant":
                assert vol.kind == _abi.LJ_VOLUME_CONSTANT and list(vol.value) == gv["value"]
                continue
            assert vol.kind == _abi.LJ_VOLUME_GRID and list(vol.resolution) == gv["resolution"]
            assert list(vol.p_min) == gv["p_min"] and list(vol.p_max) == gv["p_max"] and list(vol.max_data) == gv["max_data"] and vol.scale == gv["scale"]
            n = vol.resolution[0] * vol.resolution[1] * vol.resolution[2]
            vox = np.ctypeslib.as_array(vol.data, shape=(n, 3)).astype(np.float64)
            assert close(vox.sum(axis=0), gv["data_sum"], rel=1e-12)
            st = max(1, n // 17)
            assert np.array_equal(vox[::st], np.array(gv["samples"]))


def _fnv1a64(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a, "<f4").tobytes():
        h = ((h ^ b) * 1099511628211) & 0xffffffffffffffff
    return "%016x" % h


def test_every_image_format_the_reference_reads_decodes_to_the_same_texels():
    """imread3 / imread1 hand .jpg, .png, .hdr, .tga, .bmp, .psd, .gif and .pic files to stb_image and .exr files to tinyexr (image.cpp:28-133).  tests/assets/images/ holds small files: baseline and
    progressive JPEG (4:2:0 / 4:2:2 / 4:4:4 / grey, restart intervals, optimised tables, libjpeg's full progressive scan script, Adobe CMYK; one channel
    of a YCbCr file is its Y plane), every PNG colour type / bit depth / filter type, Adam7, palettes with tRNS, Radiance HDR in its three encodings, every TGA image type
    (colour-mapped, true colour, grey; raw and run-length) x pixel / palette depth x origin, and BMP files of every header size, palette
    depth, 16 / 24 / 32-bit pixel layout and bit-field mask the reference's loader accepts, PSD files (raw / PackBits planes, 8 / 16 bits, an alpha plane that un-mattes the colours), GIF files (first frame: global / local palettes,
    interlace, a transparent index, a frame inside a larger screen with a background index, a code table that fills up), Softimage PIC files (raw / pure / mixed run-length
    packets, channels split over packets), and tiled OpenEXR files (edge tiles, HALF /
    FLOAT channels, ZIP, a mip-mapped file: tinyexr's LoadEXR assembles level 0) (written by oracle/make_image_assets.py);
    tests/golden/image_decode.json is what the reference's own loaders return for them (oracle/decode_with_reference.cpp).  Our decoders
    (jpeg_decode.cpp, png_decode.cpp, tga_bmp_decode.cpp, exr_decode.cpp) must return the same floats, bit for bit."""
    g = golden("image_decode")["files"]
    assert len(g) >= 84 and {"base_cmyk.jpg", "prog_cmyk.jpg","base_420.jpg", "base_444_restart.jpg", "base_grey_optimized.jpg", "prog_420.jpg", "prog_422.jpg", "prog_444.jpg", "prog_grey.jpg",
                             "prog_420_restart.jpg", "prog_one_block.jpg","rgb_mixed.pic", "rgba_raw_alpha_rle.pic", "split_channels.pic", "wide_long_runs.pic","pal256.gif", "pal16_interlaced_transparent.gif", "local32_inset_bg.gif", "pal2_inset_interlaced.gif", "big_table_reset.gif","rgb8_raw.psd", "rgb8_rle.psd", "rgba8_rle.psd", "rgba16_raw_5ch.psd", "grey_pair_rle.psd","tiled_half_none.exr", "tiled_float_zip.exr", "tiled_grey_zip.exr", "tiled_mipmap_zip.exr", "gray1.png", "rgb16_interlaced.png", "pal4.png", "graya16.png", "rle.hdr", "flat_wide.hdr", "narrow.hdr",
                             "rgb24_rle.tga", "rgb15_rle.tga", "graya16.tga", "pal16_16.tga", "pal8_15_start.tga", "gray8_rle.tga",
                             "rgb24_core.bmp", "rgb32_fields_v3.bmp", "rgb16_odd_fields.bmp", "pal1.bmp", "pal4_gap.bmp", "rgb16_v4_4444.bmp"} <= set(g)
    for name, e in sorted(g.items()):
        for ch, key in ((3, "imread3"), (1, "imread1")):
            img = lj.read_image(os.path.join(ROOT, "tests", "assets", "images", name), ch)
            assert img.shape == (e[key]["height"], e[key]["width"], ch), (name, key)
            n = len(e[key]["first_texels"])
            assert np.array_equal(img.reshape(-1)[:n], np.array(e[key]["first_texels"], np.float32)), (name, key)
            assert _fnv1a64(img.reshape(-1)) == e[key]["fnv1a64"], (name, key)


def test_unsupported_and_broken_images_fail_loudly(tmp_path):
    bmp_rle = b"BM" + (54 + 8).to_bytes(4, "little") + b"\0" * 4 + (54).to_bytes(4, "little") + (40).to_bytes(4, "little") + (2).to_bytes(4, "little") * 2 + b"\1\0\x08\0" + (1).to_bytes(4, "little") + b"\0" * 28
    for name, payload in (("x.tga", b"\0" * 64), ("x.gif", b"GIF89a" + b"\0" * 32), ("x.pic", b"\x53\x80\xf6\x34" + b"\0" * 100), ("x.tif", b"II*\0" + b"\0" * 100), ("x.bmp", bmp_rle), ("y.bmp", b"BM" + b"\0" * 60),
                          ("x.png", b"\x89PNG\r\n\x1a\n" + b"\0" * 40), ("x.hdr", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 9\n\2\2\0")):
        p = tmp_path / name
        p.write_bytes(payload)
        with pytest.raises(lj.LajollaError) as e:
            lj.read_image(str(p), 3)
        assert e.value.code in (_abi.LJ_ERR_UNSUPPORTED, _abi.LJ_ERR_PARSE)
    with pytest.raises(lj.LajollaError) as e:
        lj.read_image(str(tmp_path / "missing.png"), 3)
    assert e.value.code == _abi.LJ_ERR_IO


def test_jpeg_segments_shorter_than_their_payload_are_refused(tmp_path):
    """A DQT / DHT / SOF / DRI / SOS segment whose declared length is shorter than what its parser reads, at the end of the file, must be a
    parse error, not a read past the buffer (round-2 advisor: FF D8 FF DB 00 03 00 read 64 bytes beyond a 7-byte file)."""
    soi = b"\xff\xd8"
    cases = {"dqt7": soi + b"\xff\xdb\x00\x03\x00", "dqt16": soi + b"\xff\xdb\x00\x05\x10\x01\x02", "dht": soi + b"\xff\xc4\x00\x04\x00\x01",
             "dht_syms": soi + b"\xff\xc4\x00\x14\x00" + b"\x10" * 16 + b"\x00", "sof6": soi + b"\xff\xc0\x00\x04\x08\x00",
             "sof_comps": soi + b"\xff\xc0\x00\x08\x08\x00\x08\x00\x08\x03", "dri": soi + b"\xff\xdd\x00\x02",
             "sos": soi + b"\xff\xc0\x00\x0b\x08\x00\x08\x00\x08\x01\x01\x11\x00" + b"\xff\xda\x00\x02"}
    for name, payload in cases.items():
        p = tmp_path / (name + ".jpg")
        p.write_bytes(payload)
        with pytest.raises(lj.LajollaError) as e:
            lj.read_image(str(p), 3)
        assert e.value.code in (_abi.LJ_ERR_PARSE, _abi.LJ_ERR_UNSUPPORTED), name


def test_image_headers_that_do_not_fit_their_file_are_refused(tmp_path):
    """A header may claim any size; nothing is allocated for one the file could not possibly hold (found by tools/fuzz_decoders.sh: a damaged
    BMP asked for 1.5 TB)."""
    src = open(os.path.join(ROOT, "tests", "assets", "images", "rgb24.bmp"), "rb").read()
    big = bytearray(src)
    big[18:22] = (1 << 23).to_bytes(4, "little")     # width
    big[22:26] = (1 << 23).to_bytes(4, "little")     # height
    p = tmp_path / "huge.bmp"
    p.write_bytes(bytes(big))
    with pytest.raises(lj.LajollaError) as e:
        lj.read_image(str(p), 3)
    assert e.value.code == _abi.LJ_ERR_PARSE
    tga = bytearray(open(os.path.join(ROOT, "tests", "assets", "images", "rgb24_rle.tga"), "rb").read())
    tga[12:16] = b"\xff\xff\xff\xff"                  # 65535 x 65535 run-length TGA in a 2 KB file
    p = tmp_path / "huge.tga"
    p.write_bytes(bytes(tga))
    with pytest.raises(lj.LajollaError) as e:
        lj.read_image(str(p), 3)
    assert e.value.code == _abi.LJ_ERR_PARSE


def test_damaged_serialized_mesh_and_volume_headers_are_refused(tmp_path):
    """Found by tools/fuzz_scenes.sh: a sub-mesh dictionary that points outside the file, and a grid volume whose header asks for more voxels
    than the file could hold, have to be refused — not followed / allocated."""
    import shutil
    d = tmp_path / "matpreview"
    shutil.copytree(os.path.join(ROOT, "scenes", "matpreview"), d)
    ser = bytearray((d / "matpreview.serialized").read_bytes())
    ser[-4:] = (0x7fffffff).to_bytes(4, "little")          # sub-mesh count: the dictionary would start far before the file
    (d / "matpreview.serialized").write_bytes(bytes(ser))
    with pytest.raises(lj.LajollaError) as e:
        lj.parse_scene(str(d / "matpreview.xml"))
    assert e.value.code == _abi.LJ_ERR_PARSE
    v = tmp_path / "vol"
    v.mkdir()
    for f in ("hetvol.xml", "smoke.vol", "bounds.obj"):
        if os.path.exists(os.path.join(ROOT, "scenes", "volpath_test", f)):
            shutil.copy(os.path.join(ROOT, "scenes", "volpath_test", f), v / f)
    vol = bytearray((v / "smoke.vol").read_bytes())
    vol[8:20] = (30000).to_bytes(4, "little") * 3         # 2.7e13 voxels
    (v / "smoke.vol").write_bytes(bytes(vol))
    with pytest.raises(lj.LajollaError) as e:
        lj.parse_scene(str(v / "hetvol.xml"))
    assert e.value.code == _abi.LJ_ERR_PARSE
