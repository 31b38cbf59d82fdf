#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — writes the small PNG / Radiance HDR / TGA / BMP files the texture-decoder tests read (tests/assets/images/), with
its own encoders (zlib from the standard library), so that every PNG colour type, bit depth, filter type and the Adam7 interlace, every
TGA image type / pixel depth / origin and every BMP header / pixel depth / mask layout the reference's loader accepts occur:

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


def write_tga(name, px, image_type, bpp, rng, palette=None, pal_bits=24, top_down=False, id_len=0, pal_start=0):
    """px: (h, w) palette indices / 16-bit words / grey, or (h, w, c) bytes in FILE order (blue first); rows are written in file order
    (bottom-up unless top_down).  image_type 1 / 2 / 3, +8: run-length packets (raw and repeat packets mixed)."""
    h, w = px.shape[:2]
    rle = image_type >= 8
    pal_len = 0 if palette is None else len(palette)
    hdr = struct.pack("<BBBHHBHHHHBB", id_len, 1 if palette is not None else 0, image_type, pal_start, pal_len, pal_bits if palette is not None else 0,
                      0, 0, w, h, bpp, (0x20 if top_down else 0) | (8 if bpp == 32 else 0))
    out = bytearray(hdr) + bytes(rng.integers(0, 256, id_len).tolist())
    if palette is not None:
        out += bytes(pal_start)           # (the reference's loader skips `first entry index` BYTES before the colour map)
        for e in palette:
            out += struct.pack("<H", int(e)) if pal_bits in (15, 16) else bytes(int(v) for v in e)
    nb = (bpp + 7) // 8
    def pixel(y, x):
        v = px[y, x]
        if np.ndim(v) == 0:
            return int(v).to_bytes(nb, "little")
        return bytes(int(c) for c in v)
    rows = range(h) if top_down else range(h - 1, -1, -1)
    stream = [pixel(y, x) for y in rows for x in range(w)]
    if not rle:
        out += b"".join(stream)
    else:
        i = 0
        while i < len(stream):
            run = 1
            while i + run < len(stream) and run < 128 and stream[i + run] == stream[i]:
                run += 1
            if run >= 2:
                out += bytes([0x80 | (run - 1)]) + stream[i]
                i += run
            else:
                n = min(int(rng.integers(1, 9)), len(stream) - i)
                out += bytes([n - 1]) + b"".join(stream[i:i + n])
                i += n
    open(os.path.join(OUT, name), "wb").write(bytes(out))


def write_bmp(name, rows, bpp, hsz=40, palette=None, masks=None, top_down=False, compress=0, gap=0):
    """rows: list of h packed scanlines (bytes, unpadded), TOP row first; written bottom-up unless top_down."""
    h = len(rows)
    w = rows_width[0]
    body = b""
    for r in (rows if top_down else rows[::-1]):
        body += r + bytes((-len(r)) & 3)
    pal = b""
    if palette is not None:
        for e in palette:
            pal += bytes([int(e[2]), int(e[1]), int(e[0])]) + (b"" if hsz == 12 else b"\0")
    if hsz == 12:
        info = struct.pack("<IHHHH", 12, w, h, 1, bpp)
    else:
        info = struct.pack("<IiiHHIIiiII", hsz, w, -h if top_down else h, 1, bpp, compress, len(body), 2835, 2835, 0, 0)
        if hsz == 56:
            info += struct.pack("<IIII", 0x11, 0x22, 0x33, 0x44)     # (skipped by the reference's loader, which then reads the masks that follow)
        if hsz in (40, 56) and compress == 3:
            info += struct.pack("<III", *masks[:3])
        if hsz in (108, 124):
            m = list(masks) if masks is not None else [0, 0, 0, 0]
            info += struct.pack("<IIII", *m) + struct.pack("<I", 0x73524742) + bytes(48)
            if hsz == 124:
                info += bytes(16)
    offset = 14 + len(info) + len(pal) + gap
    data = b"BM" + struct.pack("<IHHI", offset + len(body), 0, 0, offset) + info + pal + bytes(gap) + body
    open(os.path.join(OUT, name), "wb").write(data)


rows_width = [0]


def bmp_rows(px, bpp):
    """px: (h, w) integers (indices or packed pixel words) or (h, w, 3) RGB bytes -> packed scanlines"""
    h, w = px.shape[:2]
    rows_width[0] = w
    rows = []
    for y in range(h):
        if bpp == 24:
            rows.append(bytes(int(c) for p_ in px[y] for c in (p_[2], p_[1], p_[0])))
        elif bpp in (16, 32):
            rows.append(b"".join(int(v).to_bytes(bpp // 8, "little") for v in px[y]))
        elif bpp == 8:
            rows.append(bytes(int(v) for v in px[y]))
        else:
            bits = "".join(format(int(v), "0%db" % bpp) for v in px[y])
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows


def exr_zip(raw):
    """OpenEXR ZIP block: bytes split into even / odd halves, delta-coded, deflated (stored raw when that is not smaller)"""
    n = len(raw)
    t = bytearray(raw[0::2] + raw[1::2])
    prev = t[0]
    for i in range(1, n):
        cur = t[i]
        t[i] = (cur - prev + 128) & 0xff
        prev = cur
    z = zlib.compress(bytes(t), 6)
    return z if len(z) < n else raw


def write_exr_tiled(name, chans, tile, compression, mipmap=False):
    """chans: list of (name, 'half' | 'float', (h, w) array), written in alphabetical order; tile = (xs, ys); compression 0 (none) or 3
    (zip: one block per tile).  mipmap: the lower levels are present too (box-filtered), as a MIPMAP_LEVELS / ROUND_DOWN file holds them."""
    chans = sorted(chans, key=lambda c: c[0])
    h, w = chans[0][2].shape
    xs, ys = tile
    def attr(n, t, v):
        return n.encode() + b"\0" + t.encode() + b"\0" + struct.pack("<I", len(v)) + v
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", 1 if t == "half" else 2, 0, 0, 0, 0, 1, 1) for n, t, _ in chans) + b"\0"
    hdr = struct.pack("<II", 20000630, 2 | 0x200)
    hdr += attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([compression]))
    hdr += attr("dataWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1)) + attr("displayWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1))
    hdr += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    hdr += attr("tiles", "tiledesc", struct.pack("<IIB", xs, ys, 1 if mipmap else 0)) + b"\0"
    levels = [[c[2] for c in chans]]
    if mipmap:
        lw, lh = w, h
        while max(lw, lh) > 1:
            lw, lh = max(1, lw // 2), max(1, lh // 2)
            levels.append([np.ascontiguousarray(a[: lh * (a.shape[0] // lh) : a.shape[0] // lh, : lw * (a.shape[1] // lw) : a.shape[1] // lw][:lh, :lw]) for a in levels[0]])
    chunks = []
    for l, arrs in enumerate(levels):
        lh, lw = arrs[0].shape
        for ty in range((lh + ys - 1) // ys):
            for tx in range((lw + xs - 1) // xs):
                raw = b""
                for y in range(ty * ys, min(lh, (ty + 1) * ys)):
                    for (n, t, _), a in zip(chans, arrs):
                        raw += a[y, tx * xs:min(lw, (tx + 1) * xs)].astype("<f2" if t == "half" else "<f4").tobytes()
                data = exr_zip(raw) if compression == 3 else raw
                chunks.append(struct.pack("<iiiii", tx, ty, l, l, len(data)) + data)
    pos = len(hdr) + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", pos)
        pos += len(c)
    open(os.path.join(OUT, name), "wb").write(hdr + table + b"".join(chunks))


def write_psd(name, planes, depth, rle, rng):
    """planes: list of (h, w) integer arrays (R, G, B[, A[, extra]]); depth 8 or 16; rle: PackBits rows (8-bit only)"""
    h, w = planes[0].shape
    out = bytearray(b"8BPS" + struct.pack(">H6xHIIHH", 1, len(planes), h, w, depth, 3))
    out += struct.pack(">I", 0) + struct.pack(">I", 6) + b"\x01\x02\x03\x04\x05\x06" + struct.pack(">I", 0)   # mode data | image resources | layers
    out += struct.pack(">H", 1 if rle else 0)
    if not rle:
        for p_ in planes:
            out += p_.astype(">u2" if depth == 16 else np.uint8).tobytes()
    else:
        rows, counts = [], []
        for p_ in planes:
            for y in range(h):
                row, enc, i = [int(v) for v in p_[y]], bytearray(), 0
                while i < w:
                    run = 1
                    while i + run < w and run < 128 and row[i + run] == row[i]:
                        run += 1
                    if run >= 2:
                        enc += bytes([257 - run, row[i]]); i += run
                    else:
                        n = min(int(rng.integers(1, 6)), w - i)
                        enc += bytes([n - 1] + row[i:i + n]); i += n
                    if rng.integers(0, 9) == 0:
                        enc += bytes([128])          # a no-op packet
                rows.append(bytes(enc)); counts.append(len(enc))
        out += b"".join(struct.pack(">H", c) for c in counts) + b"".join(rows)
    open(os.path.join(OUT, name), "wb").write(bytes(out))


def gif_lzw(indices, mcs):
    """GIF's variable-width LZW (clear code first, again whenever the table is full), packed into data sub-blocks"""
    clear, eoi = 1 << mcs, (1 << mcs) + 1
    bits, nbits, data = 0, 0, bytearray()
    def emit(code, size):
        nonlocal bits, nbits
        bits |= code << nbits
        nbits += size
        while nbits >= 8:
            data.append(bits & 0xff); bits >>= 8; nbits -= 8
    table = {(i,): i for i in range(clear)}
    size, nxt = mcs + 1, eoi + 1
    emit(clear, size)
    w = ()
    for k in indices:
        k = int(k)
        if w + (k,) in table:
            w = w + (k,)
            continue
        emit(table[w], size)
        table[w + (k,)] = nxt
        nxt += 1
        if nxt > (1 << size) and size < 12:
            size += 1
        if nxt == 4096:
            emit(clear, size)
            table = {(i,): i for i in range(clear)}
            size, nxt = mcs + 1, eoi + 1
        w = (k,)
    if w:
        emit(table[w], size)
    emit(eoi, size)
    if nbits:
        data.append(bits & 0xff)
    out = bytearray()
    for i in range(0, len(data), 255):
        blk = data[i:i + 255]
        out += bytes([len(blk)]) + blk
    return bytes(out) + b"\0"


def write_gif(name, screen, frame, idx, gpal=None, lpal=None, bgindex=0, transparent=None, interlace=False, comment=False, version=b"89a"):
    """screen (W, H); frame (x0, y0); idx: (h, w) palette indices; palettes: (2^n, 3) arrays"""
    W_, H_ = screen
    h, w = idx.shape
    def pal_bits(p_):
        return int(np.log2(len(p_))) - 1
    out = bytearray(b"GIF" + version + struct.pack("<HHBBB", W_, H_, (0x80 | pal_bits(gpal)) if gpal is not None else 0, bgindex, 0))
    if gpal is not None:
        out += gpal.astype(np.uint8).tobytes()
    if comment:
        out += b"\x21\xfe\x05hello\x00"
    if transparent is not None:
        out += b"\x21\xf9\x04" + struct.pack("<BHB", 0x01, 7, transparent) + b"\0"
    out += b"\x2c" + struct.pack("<HHHHB", frame[0], frame[1], w, h, (0x40 if interlace else 0) | ((0x80 | pal_bits(lpal)) if lpal is not None else 0))
    if lpal is not None:
        out += lpal.astype(np.uint8).tobytes()
    rows = list(range(h))
    if interlace:
        rows = [y for start, step in ((0, 8), (4, 8), (2, 4), (1, 2)) for y in range(start, h, step)]
    n_colours = len(lpal if lpal is not None else gpal)
    mcs = max(2, int(np.log2(n_colours)))
    out += bytes([mcs]) + gif_lzw(np.concatenate([idx[y] for y in rows]), mcs) + b"\x3b"
    open(os.path.join(OUT, name), "wb").write(bytes(out))


def write_pic(name, rgba, packets, rng):
    """rgba: (h, w, 4) bytes; packets: list of (type, channel mask) — type 0 raw, 1 pure run-length, 2 mixed run-length; mask 0x80 R ... 0x10 A"""
    h, w, _ = rgba.shape
    out = bytearray(b"\x53\x80\xf6\x34" + struct.pack(">f", 0.0) + b"made by oracle/make_image_assets.py".ljust(80, b"\0") + b"PICT" + struct.pack(">HHfHH", w, h, 1.0, 3, 0))
    for i, (t, ch) in enumerate(packets):
        out += bytes([1 if i + 1 < len(packets) else 0, 8, t, ch])
    for y in range(h):
        for t, ch in packets:
            sel = [k for k in range(4) if ch & (0x80 >> k)]
            vals = [bytes(int(rgba[y, x, k]) for k in sel) for x in range(w)]
            if t == 0:
                out += b"".join(vals)
                continue
            i = 0
            while i < w:
                run = 1
                while i + run < w and run < (255 if t == 1 else 300) and vals[i + run] == vals[i]:
                    run += 1
                if t == 1:
                    out += bytes([run]) + vals[i]; i += run
                elif run >= 2:
                    out += (bytes([128]) + struct.pack(">H", run) if run > 128 or rng.integers(0, 4) == 0 else bytes([run + 127])) + vals[i]; i += run
                else:
                    n = min(int(rng.integers(1, 7)), w - i)
                    out += bytes([n - 1]) + b"".join(vals[i:i + n]); i += n
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
    # ---- TGA: every image type x pixel depth the reference's loader accepts, both origins, run-length packets, an image-id field
    runs = (xx // 6 + yy // 3)          # long runs for the run-length files
    bgr = rng.integers(0, 256, (H, W, 3)); bgr[: H // 2] = np.stack([runs * 9 % 256, runs * 5 % 256, runs * 3 % 256], -1)[: H // 2]
    bgra = np.concatenate([bgr, rng.integers(0, 256, (H, W, 1))], -1)
    w16 = rng.integers(0, 65536, (H, W)); w16[: H // 2] = (runs * 1057 % 65536)[: H // 2]
    g8 = rng.integers(0, 256, (H, W)); g8[: H // 2] = (runs * 11 % 256)[: H // 2]
    write_tga("rgb24.tga", bgr, 2, 24, rng)
    write_tga("rgb24_top.tga", bgr, 2, 24, rng, top_down=True, id_len=7)
    write_tga("rgb24_rle.tga", bgr, 10, 24, rng)
    write_tga("rgba32.tga", bgra, 2, 32, rng)
    write_tga("rgba32_rle_top.tga", bgra, 10, 32, rng, top_down=True)
    write_tga("rgb16.tga", w16, 2, 16, rng)
    write_tga("rgb15_rle.tga", w16 & 0x7fff, 10, 15, rng)
    write_tga("gray8.tga", g8, 3, 8, rng)
    write_tga("gray8_rle.tga", g8, 11, 8, rng, top_down=True)
    write_tga("graya16.tga", w16, 3, 16, rng)                       # 16-bit grey = grey + alpha
    pal24 = rng.integers(0, 256, (200, 3)); pal32 = rng.integers(0, 256, (64, 4)); pal16 = rng.integers(0, 65536, 300); pal8 = rng.integers(0, 256, (40, 1))
    write_tga("pal8_24.tga", rng.integers(0, 210, (H, W)), 1, 8, rng, palette=pal24, pal_bits=24)           # (indices >= 200 read entry 0)
    write_tga("pal8_32_rle.tga", runs % 64, 9, 8, rng, palette=pal32, pal_bits=32, top_down=True)
    write_tga("pal16_16.tga", rng.integers(0, 300, (H, W)), 1, 16, rng, palette=pal16, pal_bits=16)
    write_tga("pal8_15_start.tga", rng.integers(0, 30, (H, W)), 1, 8, rng, palette=pal16[:30] & 0x7fff, pal_bits=15, pal_start=5)
    write_tga("pal8_8.tga", rng.integers(0, 40, (H, W)), 1, 8, rng, palette=pal8, pal_bits=8)
    # ---- BMP: header variants x pixel depths x mask layouts, both row orders, a gap before the pixel data
    rgb = bgr[..., ::-1]
    write_bmp("rgb24.bmp", bmp_rows(rgb, 24), 24)
    write_bmp("rgb24_top.bmp", bmp_rows(rgb, 24), 24, top_down=True)
    write_bmp("rgb24_core.bmp", bmp_rows(rgb, 24), 24, hsz=12)
    write_bmp("rgb24_v5.bmp", bmp_rows(rgb, 24), 24, hsz=124)
    w32 = rng.integers(0, 1 << 32, (H, W), dtype=np.uint64)
    write_bmp("rgb32.bmp", bmp_rows(w32, 32), 32)                                                            # default masks: B, G, R, A bytes
    write_bmp("rgb32_v4_alpha.bmp", bmp_rows(w32, 32), 32, hsz=108, masks=(0xff0000, 0xff00, 0xff, 0xff000000), compress=3)
    write_bmp("rgb32_fields.bmp", bmp_rows(w32, 32), 32, masks=(0x0ff00000, 0x0003f000, 0x00000ff0), compress=3)
    write_bmp("rgb32_fields_v3.bmp", bmp_rows(w32, 32), 32, hsz=56, masks=(0x000000ff, 0x0000ff00, 0x00ff0000), compress=3)
    write_bmp("rgb16_555.bmp", bmp_rows(w16, 16), 16)
    write_bmp("rgb16_565.bmp", bmp_rows(w16, 16), 16, masks=(0xf800, 0x07e0, 0x001f), compress=3)
    write_bmp("rgb16_v4_4444.bmp", bmp_rows(w16, 16), 16, hsz=108, masks=(0x0f00, 0x00f0, 0x000f, 0xf000), compress=3)
    write_bmp("rgb16_odd_fields.bmp", bmp_rows(w16, 16), 16, masks=(0xe000, 0x1800, 0x00fe), compress=3)      # 3-, 2- and 7-bit channels
    write_bmp("pal8.bmp", bmp_rows(rng.integers(0, 200, (H, W)), 8), 8, palette=rng.integers(0, 256, (200, 3)))
    write_bmp("pal8_top.bmp", bmp_rows(rng.integers(0, 256, (H, W)), 8), 8, palette=rng.integers(0, 256, (256, 3)), top_down=True)
    write_bmp("pal4_gap.bmp", bmp_rows(rng.integers(0, 16, (H, W)), 4), 4, palette=rng.integers(0, 256, (16, 3)), gap=6)
    write_bmp("pal1.bmp", bmp_rows((xx * 3 + yy * 5) // 4 % 2, 1), 1, palette=rng.integers(0, 256, (2, 3)))
    write_bmp("pal4_v5.bmp", bmp_rows(rng.integers(0, 9, (H, W)), 4), 4, hsz=124, palette=rng.integers(0, 256, (9, 3)))
    # ---- tiled OpenEXR (single part): edge tiles, HALF and FLOAT channels, an alpha channel to skip, one grey channel, a mip-mapped file
    fr, fg, fb = (rng.random((H, W)) * 4 - 1 for _ in range(3))
    write_exr_tiled("tiled_half_none.exr", [("R", "half", fr), ("G", "half", fg), ("B", "half", fb)], (16, 8), 0)
    write_exr_tiled("tiled_float_zip.exr", [("R", "float", fr), ("G", "float", fg), ("B", "float", fb), ("A", "half", fr * 0 + 1)], (10, 10), 3)
    write_exr_tiled("tiled_grey_zip.exr", [("Y", "half", np.round(fg * 8) / 8)], (64, 64), 3)
    write_exr_tiled("tiled_mipmap_zip.exr", [("R", "half", fb), ("G", "float", fr), ("B", "half", fg)], (8, 8), 3, mipmap=True)
    # ---- PSD: raw and PackBits planes, 8 and 16 bits, with / without an alpha plane (un-matted from white), a fifth channel that is skipped
    pr, pg, pb = (rng.integers(0, 256, (H, W)) for _ in range(3))
    pr[: H // 2] = (runs * 7 % 256)[: H // 2]; pg[: H // 2] = (runs * 3 % 256)[: H // 2]
    pa = rng.integers(0, 256, (H, W)); pa[:, : W // 3] = 255; pa[:, W // 3: W // 2] = 0
    write_psd("rgb8_raw.psd", [pr, pg, pb], 8, False, rng)
    write_psd("rgb8_rle.psd", [pr, pg, pb], 8, True, rng)
    write_psd("rgba8_rle.psd", [pr, pg, pb, pa], 8, True, rng)
    write_psd("rgba16_raw_5ch.psd", [pr * 257, pg * 255 + 9, pb * 256 + 255, pa * 257, pr], 16, False, rng)
    write_psd("grey_pair_rle.psd", [pg, pb], 8, True, rng)          # two channels: blue reads 0
    # ---- GIF (first frame): global / local palettes, interlace, transparency, a frame inside a larger screen with a background index,
    # a picture large and random enough to fill the code table (clear code in mid-stream)
    gp256, gp16, gp2, lp32 = rng.integers(0, 256, (256, 3)), rng.integers(0, 256, (16, 3)), rng.integers(0, 256, (2, 3)), rng.integers(0, 256, (32, 3))
    write_gif("pal256.gif", (W, H), (0, 0), rng.integers(0, 256, (H, W)), gpal=gp256, version=b"87a")
    write_gif("pal16_interlaced_transparent.gif", (W, H), (0, 0), (runs + rng.integers(0, 2, (H, W))) % 16, gpal=gp16, transparent=3, interlace=True, comment=True)
    write_gif("local32_inset_bg.gif", (W + 9, H + 6), (4, 3), rng.integers(0, 32, (H, W)), gpal=gp16, lpal=lp32, bgindex=5, transparent=7)
    write_gif("pal2_inset_interlaced.gif", (W, H), (1, 2), (xx[: H - 5, : W - 3] // 3 + yy[: H - 5, : W - 3]) % 2, gpal=gp2, bgindex=1, interlace=True)
    write_gif("big_table_reset.gif", (160, 120), (0, 0), rng.integers(0, 256, (120, 160)), gpal=gp256)
    # ---- Softimage PIC: raw / pure run-length / mixed run-length packets, channels split over packets, with and without alpha
    prgba = np.stack([pr, pg, pb, pa], -1)
    write_pic("rgb_mixed.pic", prgba, [(2, 0xE0)], rng)
    write_pic("rgba_raw_alpha_rle.pic", prgba, [(0, 0xE0), (1, 0x10)], rng)
    write_pic("split_channels.pic", prgba, [(1, 0x80), (2, 0x60)], rng)
    wide = np.zeros((5, 400, 4), np.int64); wide[..., 0] = 7; wide[2, 100:, 1] = 200; wide[..., 2] = (np.arange(400) // 150) * 90
    write_pic("wide_long_runs.pic", wide, [(2, 0xE0)], rng)
    # ---- JPEG: baseline and progressive files written by Pillow's libjpeg (this script runs in the authoring container only; the files
    # are committed): chroma subsampling 4:2:0 / 4:2:2 / 4:4:4, grey, restart intervals, optimised Huffman tables; progressive files
    # carry libjpeg's default scan script (interleaved DC, per-component AC bands, successive-approximation refinement scans)
    from PIL import Image
    def photo(w, h):
        y, x = np.mgrid[0:h, 0:w]
        base = np.stack([128 + 100 * np.sin(x / 5.0) * np.cos(y / 7.0), 128 + 90 * np.cos((x + y) / 9.0), 40 + 2.5 * x + 1.5 * y], -1)
        return np.clip(base + rng.normal(0, 18, (h, w, 3)), 0, 255).astype(np.uint8)
    small, large = photo(W, H), photo(83, 61)
    Image.fromarray(small).save(os.path.join(OUT, "base_420.jpg"), quality=75)
    Image.fromarray(large).save(os.path.join(OUT, "base_444_restart.jpg"), quality=90, subsampling=0, restart_marker_blocks=3)
    Image.fromarray(small[..., 1]).save(os.path.join(OUT, "base_grey_optimized.jpg"), quality=60, optimize=True)
    Image.fromarray(small).save(os.path.join(OUT, "prog_420.jpg"), quality=75, progressive=True)
    Image.fromarray(small).save(os.path.join(OUT, "prog_422.jpg"), quality=85, subsampling=1, progressive=True)
    Image.fromarray(large).save(os.path.join(OUT, "prog_444.jpg"), quality=92, subsampling=0, progressive=True)
    Image.fromarray(large[..., 0]).save(os.path.join(OUT, "prog_grey.jpg"), quality=70, progressive=True)
    Image.fromarray(large).save(os.path.join(OUT, "prog_420_restart.jpg"), quality=80, progressive=True, restart_marker_blocks=2)
    Image.fromarray(photo(8, 8)).save(os.path.join(OUT, "prog_one_block.jpg"), quality=95, progressive=True)
    cmyk = np.concatenate([255 - small, rng.integers(120, 256, (H, W, 1)).astype(np.uint8)], -1)
    Image.fromarray(cmyk, "CMYK").save(os.path.join(OUT, "base_cmyk.jpg"), quality=85)
    Image.fromarray(cmyk, "CMYK").save(os.path.join(OUT, "prog_cmyk.jpg"), quality=85, progressive=True)
    print("wrote", len(os.listdir(OUT)), "files to", OUT)


if __name__ == "__main__":
    main()
