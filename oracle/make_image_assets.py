#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — writes the small PNG / Radiance HDR files the texture-decoder tests read (tests/assets/images/), with its own
encoders (zlib from the standard library), so that every colour type, bit depth, filter type and the Adam7 interlace occur:

    python oracle/make_image_assets.py        # deterministic: same bytes every time

The expected decode of each file is produced by the REFERENCE's own loaders (imread3 / imread1 -> stb_image) through
oracle/decode_with_reference.cpp (built by oracle/ref_build.sh) and committed as tests/golden/image_decode.json."""
import os
import struct
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "assets", "images")
W, H = 37, 23


def chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def filter_rows(rows, bpp, rng):
    """rows: list of bytes (unfiltered scanlines); a filter type per row, cycling 0..4 from a random start"""
    out, prev = bytearray(), bytes(len(rows[0])) if rows else b""
    ft = int(rng.integers(0, 5))
    for row in rows:
        f = bytearray(len(row))
        for i in range(len(row)):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) >> 1, paeth(a, b, c)][ft]
            f[i] = (row[i] - pred) & 0xff
        out.append(ft)
        out += f
        prev, ft = row, (ft + 1) % 5
    return bytes(out)


def pack_rows(px, depth):
    """px: (h, w, ch) integer samples -> list of scanlines as bytes"""
    h, w, ch = px.shape
    rows = []
    for y in range(h):
        if depth == 16:
            rows.append(px[y].astype(">u2").tobytes())
        elif depth == 8:
            rows.append(px[y].astype(np.uint8).tobytes())
        else:
            bits = "".join(format(int(v), "0%db" % depth) for v in px[y].reshape(-1))
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows


def write_png(name, px, ctype, depth, rng, palette=None, trns=None, interlace=False):
    h, w, ch = px.shape
    bpp = max(1, ch * depth // 8)
    if not interlace:
        raw = filter_rows(pack_rows(px, depth), bpp, rng)
    else:
        raw = b""
        for x0, y0, dx, dy in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
            sub = px[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += filter_rows(pack_rows(sub, depth), bpp, rng)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    data += chunk(b"gAMA", struct.pack(">I", 45455))            # ignored by stb_image: must not change the result
    if palette is not None:
        data += chunk(b"PLTE", palette.astype(np.uint8).tobytes())
    if trns is not None:
        data += chunk(b"tRNS", trns)
    comp = zlib.compress(raw, 9)
    cut = len(comp) // 3
    data += chunk(b"IDAT", comp[:cut]) + chunk(b"IDAT", comp[cut:]) + chunk(b"IEND", b"")   # (two IDAT chunks)
    open(os.path.join(OUT, name), "wb").write(data)


def write_hdr(name, rgbe, rle):
    h, w, _ = rgbe.shape
    out = bytearray(b"#?RADIANCE\n# made by oracle/make_image_assets.py\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y %d +X %d\n" % (h, w))
    for y in range(h):
        if not rle:
            out += rgbe[y].tobytes()
            continue
        out += bytes([2, 2, (w >> 8) & 0xff, w & 0xff])
        for k in range(4):
            row, i = rgbe[y, :, k], 0
            while i < w:
                run = 1
                while i + run < w and run < 127 and row[i + run] == row[i]:
                    run += 1
                if run >= 3:
                    out += bytes([128 + run, int(row[i])])
                    i += run
                else:
                    j = i
                    while j < w and j - i < 128 and not (j + 2 < w and row[j] == row[j + 1] == row[j + 2]):
                        j += 1
                    j = max(j, i + 1)
                    out += bytes([j - i]) + row[i:j].tobytes()
                    i = j
    open(os.path.join(OUT, name), "wb").write(bytes(out))


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261004)
    yy, xx = np.mgrid[0:H, 0:W]
    smooth = lambda m: ((xx * 7 + yy * 13) % (m + 1)).astype(np.int64)
    for depth in (1, 2, 4, 8, 16):
        m = (1 << depth) - 1
        g = (smooth(m) ^ rng.integers(0, m + 1, (H, W))) & m if depth >= 8 else smooth(m)
        write_png(f"gray{depth}.png", g[..., None], 0, depth, rng)
    write_png("gray4_interlaced.png", smooth(15)[..., None], 0, 4, rng, interlace=True)
    write_png("gray8_trns.png", smooth(255)[..., None], 0, 8, rng, trns=struct.pack(">H", 17))
    for depth in (8, 16):
        m = (1 << depth) - 1
        rgb = rng.integers(0, m + 1, (H, W, 3))
        rgb[: H // 2] = np.stack([smooth(m), smooth(m)[::-1], smooth(m)[:, ::-1]], -1)[: H // 2]
        write_png(f"rgb{depth}.png", rgb, 2, depth, rng)
        write_png(f"rgb{depth}_interlaced.png", rgb, 2, depth, rng, interlace=True)
        write_png(f"rgba{depth}.png", np.concatenate([rgb, rng.integers(0, m + 1, (H, W, 1))], -1), 6, depth, rng)
        write_png(f"graya{depth}.png", np.stack([smooth(m), rng.integers(0, m + 1, (H, W))], -1), 4, depth, rng)
    write_png("rgba16_interlaced.png", rng.integers(0, 65536, (H, W, 4)), 6, 16, rng, interlace=True)
    for depth in (1, 2, 4, 8):
        n = 1 << depth
        pal = rng.integers(0, 256, (n, 3))
        write_png(f"pal{depth}.png", rng.integers(0, n, (H, W, 1)), 3, depth, rng, palette=pal, trns=bytes(rng.integers(0, 256, n // 2 + 1).tolist()) if depth == 4 else None)
    # Radiance HDR: smooth + random mantissas, exponents around 128, some e == 0 pixels, long runs
    rgbe = np.zeros((H, W, 4), np.uint8)
    rgbe[..., :3] = rng.integers(0, 256, (H, W, 3))
    rgbe[..., 3] = 120 + (xx // 5 + yy // 4) % 16
    rgbe[3:6, 4:30, :] = np.array([200, 100, 50, 130], np.uint8)
    rgbe[10, :, 3] = 0
    write_hdr("rle.hdr", rgbe, True)
    write_hdr("flat_wide.hdr", rgbe, False)        # w >= 8 without scanline headers: stb's "not run-length encoded" path
    write_hdr("narrow.hdr", rgbe[:, :5].copy(), False)   # w < 8: always flat
    print("wrote", len(os.listdir(OUT)), "files to", OUT)


if __name__ == "__main__":
    main()
