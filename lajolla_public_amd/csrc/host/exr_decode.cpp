// OpenEXR input for the texture front end (image.cpp:54-73,109-129 in the reference, which delegates to tinyexr's
// LoadEXR): single-part scan-line and tiled files (level 0 of a mip- / rip-mapped one, as LoadEXR assembles it) with HALF / FLOAT /
// UINT channels, compression NONE, ZIPS, ZIP or PIZ —
// written from the OpenEXR file-format and PIZ documentation (Huffman coding + 2-D Haar wavelet + value LUT), no
// third-party code.  Output: R, G, B as float, row-major, y = 0 at the top; a file with a single channel is replicated
// to all three (tinyexr's behaviour for LoadEXR).  HALF -> float is exact, so texels are bit-identical to what the
// reference holds (tests/test_frontend.py compares with the PFM the reference's own loader produced).
#include "host_scene.h"
#include <cstring>
#include <fstream>
#include <zlib.h>

namespace lj {

namespace {

struct Reader {
    const uint8_t *p, *end; const std::string &name;
    void need(size_t n) const { if ((size_t)(end - p) < n) throw LjError(LJ_ERR_PARSE, "truncated EXR: " + name); }
    uint8_t u8() { need(1); return *p++; }
    uint16_t u16() { need(2); uint16_t v = (uint16_t)(p[0] | (p[1] << 8)); p += 2; return v; }
    uint32_t u32() { need(4); uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); p += 4; return v; }
    int32_t i32() { return (int32_t)u32(); }
    uint64_t u64() { uint64_t lo = u32(), hi = u32(); return lo | (hi << 32); }
    std::string str() { const uint8_t *s = p; while (p < end && *p) p++; need(1); std::string r((const char *)s, (size_t)(p - s)); p++; return r; }
};

float half_to_float(uint16_t h) {
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = s;
        else {  // subnormal half: renormalise
            int sh = 0; uint32_t mm = m;
            while (!(mm & 1024u)) { mm <<= 1; sh++; }
            bits = s | ((uint32_t)(113 - sh) << 23) | ((mm & 1023u) << 13);
        }
    } else if (e == 31) bits = s | 0x7f800000u | (m << 13);
    else bits = s | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

// ---------------------------------------------------------------- PIZ: Huffman decoder
constexpr int kEncBits = 16, kDecBits = 14, kEncSize = (1 << kEncBits) + 1, kDecSize = 1 << kDecBits, kDecMask = kDecSize - 1;
constexpr int kShortZeroRun = 59, kLongZeroRun = 63, kShortestLongRun = 2 + kLongZeroRun - kShortZeroRun;

struct HufDec { int len = 0; int lit = 0; std::vector<int> longs; };

inline int huf_length(uint64_t code) { return (int)(code & 63); }
inline uint64_t huf_code(uint64_t code) { return code >> 6; }

void huf_uncompress(const uint8_t *in, size_t n_in, uint16_t *out, size_t n_out, const std::string &name) {
    auto fail = [&](const char *what) -> void { throw LjError(LJ_ERR_PARSE, std::string("EXR PIZ Huffman data: ") + what + ": " + name); };
    if (n_in < 20) { if (n_out) fail("truncated header"); return; }
    auto rd32 = [&](size_t o) { return (uint32_t)in[o] | ((uint32_t)in[o + 1] << 8) | ((uint32_t)in[o + 2] << 16) | ((uint32_t)in[o + 3] << 24); };
    const int im = (int)rd32(0), iM = (int)rd32(4);
    const uint32_t n_bits = rd32(12);
    if (im < 0 || im >= kEncSize || iM < 0 || iM >= kEncSize) fail("symbol range");
    const uint8_t *p = in + 20, *end = in + n_in;
    // ---- code lengths, 6 bits each, with run-length codes for zeros
    std::vector<uint64_t> hcode(kEncSize, 0);
    {
        uint64_t c = 0; int lc = 0;
        auto get_bits = [&](int nb) { while (lc < nb) { if (p >= end) fail("truncated code table"); c = (c << 8) | *p++; lc += 8; } lc -= nb; return (uint32_t)((c >> lc) & ((1u << nb) - 1)); };
        for (int i = im; i <= iM; i++) {
            const uint32_t l = get_bits(6);
            hcode[i] = l;
            if (l == (uint32_t)kLongZeroRun) {
                int zerun = (int)get_bits(8) + kShortestLongRun;
                if (i + zerun > iM + 1) fail("zero run past the table");
                while (zerun--) hcode[i++] = 0;
                i--;
            } else if (l >= (uint32_t)kShortZeroRun) {
                int zerun = (int)l - kShortZeroRun + 2;
                if (i + zerun > iM + 1) fail("zero run past the table");
                while (zerun--) hcode[i++] = 0;
                i--;
            }
        }
    }
    // ---- canonical codes from the lengths
    {
        uint64_t n[59] = {0};
        for (int i = 0; i < kEncSize; i++) n[hcode[i]] += 1;
        uint64_t c = 0;
        for (int i = 58; i > 0; --i) { const uint64_t nc = (c + n[i]) >> 1; n[i] = c; c = nc; }
        for (int i = 0; i < kEncSize; i++) { const int l = (int)hcode[i]; if (l > 0) hcode[i] = (uint64_t)l | (n[l]++ << 6); }
    }
    // ---- decoding table: 14-bit direct lookup, lists for longer codes
    std::vector<HufDec> dec(kDecSize);
    for (int i = im; i <= iM; i++) {
        const uint64_t c = huf_code(hcode[i]); const int l = huf_length(hcode[i]);
        if (l == 0) continue;
        if (c >> l) fail("invalid code table");
        if (l > kDecBits) { HufDec &pl = dec[(size_t)(c >> (l - kDecBits))]; if (pl.len) fail("invalid code table"); pl.longs.push_back(i); }
        else {
            HufDec *pl = &dec[(size_t)(c << (kDecBits - l))];
            for (uint64_t k = (uint64_t)1 << (kDecBits - l); k > 0; k--, pl++) { if (pl->len || !pl->longs.empty()) fail("invalid code table"); pl->len = l; pl->lit = i; }
        }
    }
    // ---- decode; symbol iM is the run-length code: the next 8 bits repeat the previous output value
    if ((size_t)(end - p) < (n_bits + 7) / 8) fail("truncated bit stream");
    const uint8_t *ie = p + (n_bits + 7) / 8;
    uint64_t c = 0; int lc = 0;
    uint16_t *o = out, *oe = out + n_out;
    auto emit = [&](int sym) {
        if (sym == iM) {
            if (lc < 8) { if (p >= ie) fail("truncated run"); c = (c << 8) | *p++; lc += 8; }
            lc -= 8;
            int cs = (int)((c >> lc) & 0xff);
            if (o == out || o + cs > oe) fail("run past the output");
            const uint16_t s = o[-1];
            while (cs-- > 0) *o++ = s;
        } else { if (o >= oe) fail("too many symbols"); *o++ = (uint16_t)sym; }
    };
    while (p < ie) {
        c = (c << 8) | *p++; lc += 8;
        while (lc >= kDecBits) {
            const HufDec &pl = dec[(size_t)((c >> (lc - kDecBits)) & kDecMask)];
            if (pl.len) { lc -= pl.len; emit(pl.lit); }
            else {
                if (pl.longs.empty()) fail("invalid code");
                size_t j = 0;
                for (; j < pl.longs.size(); j++) {
                    const int l = huf_length(hcode[pl.longs[j]]);
                    while (lc < l && p < ie) { c = (c << 8) | *p++; lc += 8; }
                    if (lc >= l && huf_code(hcode[pl.longs[j]]) == ((c >> (lc - l)) & (((uint64_t)1 << l) - 1))) { lc -= l; emit(pl.longs[j]); break; }
                }
                if (j == pl.longs.size()) fail("invalid code");
            }
        }
    }
    const int pad = (int)((8 - n_bits) & 7);
    c >>= pad; lc -= pad;
    while (lc > 0) {
        const HufDec &pl = dec[(size_t)((c << (kDecBits - lc)) & kDecMask)];
        if (!pl.len) fail("invalid code");
        lc -= pl.len; emit(pl.lit);
    }
    if (o != oe) fail("too few symbols");
}

// ---------------------------------------------------------------- PIZ: inverse 2-D Haar wavelet on 16-bit values
inline void wdec14(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    const int ls = (int16_t)l, hs = (int16_t)h;
    const int ai = ls + (hs & 1) + (hs >> 1);
    a = (uint16_t)(int16_t)ai; b = (uint16_t)(int16_t)(ai - hs);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xffff;
    const int aa = (d + bb - 0x8000) & 0xffff;
    b = (uint16_t)bb; a = (uint16_t)aa;
}
void wav2_decode(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1;
    while (p <= n) p <<= 1;
    p >>= 1;
    int p2 = p;
    p >>= 1;
    auto wdec = [&](uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) { if (w14) wdec14(l, h, a, b); else wdec16(l, h, a, b); };
    while (p >= 1) {
        uint16_t *py = in, *ey = in + (ptrdiff_t)oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t *px = py, *ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                wdec(*px, *p10, i00, i10); wdec(*p01, *p11, i01, i11);
                wdec(i00, i01, *px, *p01); wdec(i10, i11, *p10, *p11);
            }
            if (nx & p) { uint16_t *p10 = px + oy1; wdec(*px, *p10, i00, *p10); *px = i00; }
        }
        if (ny & p) {
            uint16_t *px = py, *ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) { uint16_t *p01 = px + ox1; wdec(*px, *p01, i00, *p01); *px = i00; }
        }
        p2 = p; p >>= 1;
    }
}

struct Channel { std::string name; int type; int words; };  // type 0 UINT, 1 HALF, 2 FLOAT; words = 16-bit words per pixel

// one PIZ block -> `lines` scan lines, each line = every channel's row in channel-list order (native little endian)
void piz_decode(const uint8_t *in, size_t n_in, std::vector<uint8_t> &out, int nx, int lines, const std::vector<Channel> &chans, const std::string &name) {
    Reader r{in, in + n_in, name};
    std::vector<uint8_t> bitmap(8192, 0);
    const uint16_t min_nz = r.u16(), max_nz = r.u16();
    if (max_nz >= 8192) throw LjError(LJ_ERR_PARSE, "EXR PIZ bitmap range: " + name);
    if (min_nz <= max_nz) { r.need((size_t)(max_nz - min_nz + 1)); memcpy(&bitmap[min_nz], r.p, (size_t)(max_nz - min_nz + 1)); r.p += max_nz - min_nz + 1; }
    std::vector<uint16_t> lut(65536, 0);
    int k = 0;
    for (int i = 0; i < 65536; i++) if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t max_value = (uint16_t)(k - 1);
    const int32_t length = r.i32();
    if (length < 0) throw LjError(LJ_ERR_PARSE, "EXR PIZ block length: " + name);
    r.need((size_t)length);
    size_t total = 0;
    for (const Channel &c : chans) total += (size_t)nx * lines * c.words;
    std::vector<uint16_t> tmp(total);
    huf_uncompress(r.p, (size_t)length, tmp.data(), total, name);
    size_t start = 0;
    std::vector<size_t> starts;
    for (const Channel &c : chans) {
        starts.push_back(start);
        for (int j = 0; j < c.words; j++) wav2_decode(tmp.data() + start + j, nx, c.words, lines, nx * c.words, max_value);
        start += (size_t)nx * lines * c.words;
    }
    for (auto &v : tmp) v = lut[v];
    out.resize(total * 2);
    uint8_t *o = out.data();
    std::vector<size_t> cur = starts;
    for (int y = 0; y < lines; y++)
        for (size_t c = 0; c < chans.size(); c++) {
            const size_t n = (size_t)nx * chans[c].words;
            memcpy(o, tmp.data() + cur[c], n * 2);   // host is little endian, as the file format
            o += n * 2; cur[c] += n;
        }
}

// ZIP / ZIPS block: zlib stream, then the byte predictor and the two-halves interleave are undone
void zip_decode(const uint8_t *in, size_t n_in, std::vector<uint8_t> &out, size_t n_out, const std::string &name) {
    std::vector<uint8_t> tmp(n_out);
    uLongf got = (uLongf)n_out;
    if (uncompress(tmp.data(), &got, in, (uLong)n_in) != Z_OK || got != n_out) throw LjError(LJ_ERR_PARSE, "EXR ZIP block does not inflate: " + name);
    for (size_t i = 1; i < n_out; i++) tmp[i] = (uint8_t)(tmp[i - 1] + tmp[i] - 128);
    out.resize(n_out);
    const size_t half = (n_out + 1) / 2;
    for (size_t i = 0, s = 0; s < n_out; i++) { out[s++] = tmp[i]; if (s < n_out) out[s++] = tmp[half + i]; }
}

} // namespace

HostImage decode_exr_rgb(const std::vector<uint8_t> &file, const std::string &name) {
    Reader r{file.data(), file.data() + file.size(), name};
    if (r.u32() != 20000630u) throw LjError(LJ_ERR_PARSE, "not an OpenEXR file: " + name);
    const uint32_t version = r.u32();
    if ((version & 0xff) != 2 || (version & 0x1800)) throw LjError(LJ_ERR_UNSUPPORTED, "only single-part OpenEXR images are read (no deep data or multi-part files): " + name);
    const bool tiled = (version & 0x200) != 0;
    std::vector<Channel> chans;
    int compression = -1, x0 = 0, y0 = 0, x1 = -1, y1 = -1, line_order = 0;
    uint32_t tile_w = 0, tile_h = 0;
    for (;;) {
        const std::string attr = r.str();
        if (attr.empty()) break;
        const std::string type = r.str();
        const uint32_t size = r.u32();
        r.need(size);
        Reader v{r.p, r.p + size, name};
        r.p += size;
        if (attr == "channels") {
            for (;;) {
                Channel c; c.name = v.str();
                if (c.name.empty()) break;
                c.type = v.i32(); v.u8(); v.u8(); v.u8(); v.u8();   // pLinear + 3 reserved
                const int xs = v.i32(), ys = v.i32();
                if (xs != 1 || ys != 1) throw LjError(LJ_ERR_UNSUPPORTED, "subsampled EXR channels are not read: " + name);
                if (c.type < 0 || c.type > 2) throw LjError(LJ_ERR_PARSE, "EXR channel type: " + name);
                c.words = c.type == 1 ? 1 : 2;
                chans.push_back(c);
            }
        } else if (attr == "compression") compression = v.u8();
        else if (attr == "dataWindow") { x0 = v.i32(); y0 = v.i32(); x1 = v.i32(); y1 = v.i32(); }
        else if (attr == "lineOrder") line_order = v.u8();
        else if (attr == "tiles") { tile_w = v.u32(); tile_h = v.u32(); (void)v.u8(); }   // (level mode: only level 0 is read, as LoadEXR does)
    }
    (void)line_order;   // blocks carry their own y coordinate
    if (chans.empty() || x1 < x0 || y1 < y0) throw LjError(LJ_ERR_PARSE, "EXR header lacks channels or a data window: " + name);
    check_image_size((long long)x1 - x0 + 1, (long long)y1 - y0 + 1, file.size(), name);
    const int w = x1 - x0 + 1, h = y1 - y0 + 1;
    int block_lines;
    switch (compression) {
        case 0: case 2: block_lines = 1; break;   // NONE, ZIPS
        case 3: block_lines = 16; break;          // ZIP
        case 4: block_lines = 32; break;          // PIZ
        default: throw LjError(LJ_ERR_UNSUPPORTED, "EXR compression " + std::to_string(compression) + " is not read (NONE, ZIPS, ZIP and PIZ are): " + name);
    }
    size_t line_bytes = 0;
    for (const Channel &c : chans) line_bytes += (size_t)w * c.words * 2;
    // which file channels feed R, G, B (a lone channel feeds all three)
    int src[3] = {-1, -1, -1};
    for (size_t i = 0; i < chans.size(); i++) { if (chans[i].name == "R") src[0] = (int)i; else if (chans[i].name == "G") src[1] = (int)i; else if (chans[i].name == "B") src[2] = (int)i; }
    if (chans.size() == 1) src[0] = src[1] = src[2] = 0;
    if (src[0] < 0 || src[1] < 0 || src[2] < 0) throw LjError(LJ_ERR_UNSUPPORTED, "EXR file has neither R, G, B channels nor a single channel: " + name);
    HostImage img; img.width = w; img.height = h; img.channels = 3;
    img.data.assign((size_t)w * h * 3, 0.0f);
    // `n` texels of every channel, one channel after the other (a scan line, or a tile's part of one) -> R, G, B at dst
    auto unpack_line = [&](const uint8_t *line, int n, float *dst) {
        size_t o = 0;
        std::vector<size_t> off(chans.size());
        for (size_t i = 0; i < chans.size(); i++) { off[i] = o; o += (size_t)n * chans[i].words * 2; }
        for (int k = 0; k < 3; k++) {
            const Channel &c = chans[src[k]];
            const uint8_t *s = line + off[src[k]];
            for (int x = 0; x < n; x++) {
                float v;
                if (c.type == 1) { uint16_t hv; memcpy(&hv, s + 2 * x, 2); v = half_to_float(hv); }
                else if (c.type == 2) memcpy(&v, s + 4 * x, 4);
                else { uint32_t u; memcpy(&u, s + 4 * x, 4); v = (float)u; }
                dst[3 * x + k] = v;
            }
        }
    };
    if (tiled) {
        // Tiles of level (0, 0) — what LoadEXR assembles; their offsets lead the table whatever the level mode.  A tile at the right /
        // lower edge holds only its part inside the data window; each tile is compressed on its own, lines and channels laid out as
        // in a scan-line block.
        if (tile_w == 0 || tile_h == 0 || tile_w > (1u << 20) || tile_h > (1u << 20)) throw LjError(LJ_ERR_PARSE, "tiled EXR without a tile description: " + name);
        const int ntx = (int)((w + tile_w - 1) / tile_w), nty = (int)((h + tile_h - 1) / tile_h);
        std::vector<uint64_t> offsets((size_t)ntx * nty);
        for (auto &o : offsets) o = r.u64();
        std::vector<uint8_t> raw;
        for (uint64_t at : offsets) {
            if (at + 20 > file.size()) throw LjError(LJ_ERR_PARSE, "EXR tile offset outside the file: " + name);
            Reader br{file.data() + at, file.data() + file.size(), name};
            const int tx = br.i32(), ty = br.i32(), lx = br.i32(), ly = br.i32();
            const int32_t size = br.i32();
            if (lx != 0 || ly != 0 || tx < 0 || ty < 0 || tx >= ntx || ty >= nty || size < 0) throw LjError(LJ_ERR_PARSE, "EXR tile header: " + name);
            br.need((size_t)size);
            const int px = tx * (int)tile_w, py = ty * (int)tile_h, tw = std::min((int)tile_w, w - px), th = std::min((int)tile_h, h - py);
            size_t tline = 0;
            for (const Channel &c : chans) tline += (size_t)tw * c.words * 2;
            const size_t expect = tline * th;
            const uint8_t *data = br.p;
            if ((size_t)size == expect) raw.assign(data, data + size);
            else if (compression == 4) piz_decode(data, (size_t)size, raw, tw, th, chans, name);
            else if (compression == 2 || compression == 3) zip_decode(data, (size_t)size, raw, expect, name);
            else throw LjError(LJ_ERR_PARSE, "EXR tile size: " + name);
            if (raw.size() != expect) throw LjError(LJ_ERR_PARSE, "EXR tile decodes to the wrong size: " + name);
            for (int y = 0; y < th; y++) unpack_line(raw.data() + (size_t)y * tline, tw, img.data.data() + ((size_t)(py + y) * w + px) * 3);
        }
        return img;
    }
    const int n_blocks = (h + block_lines - 1) / block_lines;
    std::vector<uint64_t> offsets(n_blocks);
    for (auto &o : offsets) o = r.u64();
    std::vector<uint8_t> raw;
    for (int b = 0; b < n_blocks; b++) {
        if (offsets[b] + 8 > file.size()) throw LjError(LJ_ERR_PARSE, "EXR block offset outside the file: " + name);
        Reader br{file.data() + offsets[b], file.data() + file.size(), name};
        const int by = br.i32() - y0;
        const int32_t size = br.i32();
        if (by < 0 || by >= h || size < 0) throw LjError(LJ_ERR_PARSE, "EXR block header: " + name);
        br.need((size_t)size);
        const int lines = std::min(block_lines, h - by);
        const size_t expect = line_bytes * lines;
        const uint8_t *data = br.p;
        if ((size_t)size == expect) raw.assign(data, data + size);   // stored uncompressed when compression does not help
        else if (compression == 4) piz_decode(data, (size_t)size, raw, w, lines, chans, name);
        else if (compression == 2 || compression == 3) zip_decode(data, (size_t)size, raw, expect, name);
        else throw LjError(LJ_ERR_PARSE, "EXR block size: " + name);
        if (raw.size() != expect) throw LjError(LJ_ERR_PARSE, "EXR block decodes to the wrong size: " + name);
        for (int y = 0; y < lines; y++) unpack_line(raw.data() + (size_t)y * line_bytes, w, img.data.data() + (size_t)(by + y) * w * 3);
    }
    return img;
}

} // namespace lj
