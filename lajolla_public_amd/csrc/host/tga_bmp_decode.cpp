// Truevision TGA, Windows BMP, Photoshop PSD, GIF and Softimage PIC texture decoders of the front end.  The reference's imread3 / imread1
// (image.cpp:28-133) hand .tga, .bmp, .psd, .gif and .pic files to stb_image's float loader; these decoders are written from the file formats and reproduce what that loader returns, texel
// for texel (pinned by tests/golden/image_decode.json, which the reference's own imread3 / imread1 produced for tests/assets/images/*):
//   TGA   colour-mapped (types 1 / 9), true colour (2 / 10) and grey (3 / 11), raw or run-length packets; 8 / 15 / 16 / 24 / 32 bits per
//         pixel or palette entry (15 / 16: five bits per channel, x * 255 / 31, the top bit ignored); rows bottom-up unless bit 5 of the
//         descriptor byte is set (the right-to-left bit 4 is ignored, as the reference's loader does); 16-bit grey = grey + alpha
//   BMP   core (12-byte), info (40 / 56), V4 (108) and V5 (124) headers; 1 / 4 / 8-bit palettes, 16 / 24 / 32-bit pixels with the default
//         or BI_BITFIELDS masks (a channel of n < 8 bits is widened by bit replication); bottom-up unless the height is negative;
//         run-length compressed files are refused (the reference's loader refuses them too)
//   PSD   version 1, RGB mode, 8 or 16 bits per channel (the high byte is kept), raw or PackBits planes; the composite image only; with an
//         alpha plane the colours are un-matted from white where 0 < alpha < 255, in float, as the reference's loader does
//   GIF   87a / 89a, the FIRST frame only: global or local palette, LZW sub-blocks, interlaced rows, a frame smaller than the screen, a
//         transparent index (such pixels stay black); pixels the frame does not cover take the background entry when its index is > 0 —
//         with red and blue exchanged, which is what the reference's loader produces
//   PIC   Softimage: a chain of 8-bit channel packets (any split of R, G, B, A), each raw, pure run-length or mixed run-length per scan line;
//         a channel no packet carries reads 255
// then the conversion every LDR file goes through: luma (77 R + 150 G + 29 B) >> 8 for one channel, and (float) pow(v / 255.0f, 2.2f).
#include <cstdint>
#include "host_scene.h"
#include <cmath>
#include <cstring>

namespace lj {

namespace {

struct Reader {
    const std::vector<uint8_t> &f; const std::string &name; size_t p = 0;
    uint8_t u8() { return p < f.size() ? f[p++] : (uint8_t)0; }   // (reads past the end return 0, as the reference's loader does)
    uint32_t u16() { const uint32_t a = u8(); return a | ((uint32_t)u8() << 8); }
    uint32_t u32() { const uint32_t a = u16(); return a | (u16() << 16); }
    void skip(long long n) { if (n > 0) p = (size_t)std::min<unsigned long long>((unsigned long long)f.size(), (unsigned long long)p + (unsigned long long)n); }
};

// 8-bit samples (1 grey, 2 grey + alpha, 3 RGB, 4 RGBA per texel) -> the float image imread3 (channels 3) / imread1 (channels 1) returns
HostImage finish_ldr(const std::vector<uint8_t> &px, int w, int h, int comp, int channels) {
    HostImage img; img.width = w; img.height = h; img.channels = channels == 1 ? 1 : 3;
    img.data.resize((size_t)w * h * img.channels);
    auto lin = [](uint8_t v) { return (float)std::pow((double)(v / 255.0f), (double)2.2f); };
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint8_t *s = &px[i * comp];
        if (img.channels == 3) {
            for (int c = 0; c < 3; c++) img.data[3 * i + c] = lin(comp >= 3 ? s[c] : s[0]);
        } else {
            img.data[i] = lin(comp >= 3 ? (uint8_t)(((uint32_t)s[0] * 77 + (uint32_t)s[1] * 150 + 29 * (uint32_t)s[2]) >> 8) : s[0]);
        }
    }
    return img;
}

void rgb555(uint32_t v, uint8_t *out) {
    out[0] = (uint8_t)((((v >> 10) & 31u) * 255u) / 31u); out[1] = (uint8_t)((((v >> 5) & 31u) * 255u) / 31u); out[2] = (uint8_t)(((v & 31u) * 255u) / 31u);
}

} // namespace

HostImage read_tga(const std::vector<uint8_t> &file, const std::string &name, int channels) {
    Reader r{file, name};
    if (file.size() < 18) throw LjError(LJ_ERR_PARSE, "truncated TGA header: " + name);
    const int id_len = r.u8(), cmap_type = r.u8();
    int type = r.u8();
    const int pal_start = (int)r.u16(), pal_len = (int)r.u16(), pal_bits = r.u8();
    r.u16(); r.u16();   // x / y origin
    const int w = (int)r.u16(), h = (int)r.u16(), bpp = r.u8(), desc = r.u8();
    // what the reference's loader accepts as a TGA at all (the format has no signature)
    auto depth_ok = [](int b) { return b == 8 || b == 15 || b == 16 || b == 24 || b == 32; };
    bool ok = cmap_type <= 1 && w >= 1 && h >= 1 && depth_ok(bpp);
    if (cmap_type == 1) ok = ok && (type == 1 || type == 9) && depth_ok(pal_bits) && (bpp == 8 || bpp == 16);
    else ok = ok && (type == 2 || type == 3 || type == 10 || type == 11);
    if (!ok) throw LjError(LJ_ERR_PARSE, "not a TGA file this build (or the reference's loader) decodes: " + name);
    check_image_size(w, h, file.size(), name);
    const bool rle = type >= 8;
    if (rle) type -= 8;
    const bool indexed = cmap_type == 1, bottom_up = ((desc >> 5) & 1) == 0;
    const int src_bits = indexed ? pal_bits : bpp;
    const bool packed16 = (src_bits == 15) || (src_bits == 16 && !(type == 3 && !indexed));
    const int comp = src_bits == 8 ? 1 : (packed16 ? 3 : (src_bits == 16 ? 2 : src_bits / 8));
    r.skip(id_len);
    std::vector<uint8_t> palette;
    if (indexed) {
        if (pal_len == 0) throw LjError(LJ_ERR_PARSE, "TGA colour map without entries: " + name);
        r.skip(pal_start);   // (the reference's loader skips `first entry index` bytes here)
        palette.resize((size_t)pal_len * comp);
        if (packed16) for (int i = 0; i < pal_len; i++) rgb555(r.u16(), &palette[(size_t)i * 3]);
        else {
            if (r.p + palette.size() > file.size()) throw LjError(LJ_ERR_PARSE, "truncated TGA colour map: " + name);
            for (auto &b : palette) b = r.u8();
        }
    }
    std::vector<uint8_t> px((size_t)w * h * comp);
    uint8_t cur[4] = {0, 0, 0, 0};
    int run = 0; bool repeating = false;
    for (size_t i = 0; i < (size_t)w * h; i++) {
        bool fetch = true;
        if (rle) {
            if (run == 0) { const int cmd = r.u8(); run = 1 + (cmd & 127); repeating = (cmd >> 7) != 0; }
            else if (repeating) fetch = false;
        }
        if (fetch) {
            if (indexed) {
                int idx = bpp == 8 ? (int)r.u8() : (int)r.u16();
                if (idx >= pal_len) idx = 0;
                memcpy(cur, &palette[(size_t)idx * comp], (size_t)comp);
            } else if (packed16) rgb555(r.u16(), cur);
            else for (int c = 0; c < comp; c++) cur[c] = r.u8();
        }
        memcpy(&px[i * comp], cur, (size_t)comp);
        run--;
    }
    if (bottom_up)
        for (int y = 0; y * 2 < h; y++)
            for (size_t k = 0; k < (size_t)w * comp; k++) std::swap(px[(size_t)y * w * comp + k], px[(size_t)(h - 1 - y) * w * comp + k]);
    if (comp >= 3 && !packed16)   // stored blue first
        for (size_t i = 0; i < (size_t)w * h; i++) std::swap(px[i * comp], px[i * comp + 2]);
    return finish_ldr(px, w, h, comp, channels);
}

HostImage read_bmp(const std::vector<uint8_t> &file, const std::string &name, int channels) {
    Reader r{file, name};
    if (r.u8() != 'B' || r.u8() != 'M') throw LjError(LJ_ERR_PARSE, "not a BMP file: " + name);
    r.u32(); r.u16(); r.u16();
    const long long offset = (int32_t)r.u32();
    const int hsz = (int)r.u32();
    if (offset < 0 || (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124)) throw LjError(LJ_ERR_UNSUPPORTED, "BMP header of this size is not decoded: " + name);
    int w, hs;
    if (hsz == 12) { w = (int)r.u16(); hs = (int)r.u16(); } else { w = (int32_t)r.u32(); hs = (int32_t)r.u32(); }
    if (r.u16() != 1) throw LjError(LJ_ERR_PARSE, "corrupt BMP (planes): " + name);
    const int bpp = (int)r.u16();
    uint32_t mr = 0, mg = 0, mb = 0, ma = 0;
    long long extra = 14;
    auto default_masks = [&]() {
        if (bpp == 16) { mr = 31u << 10; mg = 31u << 5; mb = 31u; }
        else if (bpp == 32) { mr = 0xffu << 16; mg = 0xffu << 8; mb = 0xffu; ma = 0xffu << 24; }
        else { mr = mg = mb = ma = 0; }
    };
    if (hsz != 12) {
        const int compress = (int)r.u32();
        if (compress == 1 || compress == 2) throw LjError(LJ_ERR_UNSUPPORTED, "run-length compressed BMP files are not decoded (nor by the reference's loader): " + name);
        if (compress >= 4) throw LjError(LJ_ERR_UNSUPPORTED, "BMP with embedded JPEG / PNG data: " + name);
        if (compress == 3 && bpp != 16 && bpp != 32) throw LjError(LJ_ERR_PARSE, "corrupt BMP (bit fields need 16 or 32 bits per pixel): " + name);
        for (int k = 0; k < 5; k++) r.u32();
        if (hsz == 40 || hsz == 56) {
            if (hsz == 56) for (int k = 0; k < 4; k++) r.u32();
            if (bpp == 16 || bpp == 32) {
                if (compress == 0) default_masks();
                else { mr = r.u32(); mg = r.u32(); mb = r.u32(); extra += 12; if (mr == mg && mg == mb) throw LjError(LJ_ERR_PARSE, "corrupt BMP (masks): " + name); }
            }
        } else {
            mr = r.u32(); mg = r.u32(); mb = r.u32(); ma = r.u32();
            if (compress != 3) default_masks();
            for (int k = 0; k < 13; k++) r.u32();
            if (hsz == 124) for (int k = 0; k < 4; k++) r.u32();
        }
    }
    const bool bottom_up = hs > 0;
    if (hs == INT32_MIN) throw LjError(LJ_ERR_PARSE, "BMP " + name + ": bad height");   // (-hs would overflow)
    const int h = hs < 0 ? -hs : hs;
    check_image_size(w, h, file.size(), name);
    long long psize = 0;
    if (hsz == 12) { if (bpp < 24) psize = (offset - extra - 24) / 3; }
    else if (bpp < 16) psize = (offset - extra - hsz) >> 2;
    if (psize == 0 && offset != (long long)r.p) throw LjError(LJ_ERR_PARSE, "corrupt BMP (pixel data offset): " + name);
    std::vector<uint8_t> px((size_t)w * h * 3);   // (the alpha channel never reaches a texture: imread3 / imread1 ask for 3 / 1 components)
    size_t z = 0;
    if (bpp < 16) {
        if (psize <= 0 || psize > 256) throw LjError(LJ_ERR_PARSE, "corrupt BMP (palette): " + name);
        uint8_t pal[256][3] = {};
        for (int i = 0; i < (int)psize; i++) { pal[i][2] = r.u8(); pal[i][1] = r.u8(); pal[i][0] = r.u8(); if (hsz != 12) r.u8(); }
        r.skip(offset - extra - hsz - psize * (hsz == 12 ? 3 : 4));
        int row_bytes;
        if (bpp == 1) row_bytes = (w + 7) >> 3; else if (bpp == 4) row_bytes = (w + 1) >> 1; else if (bpp == 8) row_bytes = w;
        else throw LjError(LJ_ERR_PARSE, "corrupt BMP (bits per pixel): " + name);
        const int pad = (-row_bytes) & 3;
        for (int y = 0; y < h; y++) {
            const size_t row0 = r.p;
            for (int x = 0; x < w; x++) {
                int v;
                if (bpp == 8) v = file.size() > row0 + (size_t)x ? file[row0 + x] : 0;
                else if (bpp == 4) { const int b = file.size() > row0 + (size_t)(x >> 1) ? file[row0 + (x >> 1)] : 0; v = (x & 1) ? (b & 15) : (b >> 4); }
                else { const int b = file.size() > row0 + (size_t)(x >> 3) ? file[row0 + (x >> 3)] : 0; v = (b >> (7 - (x & 7))) & 1; }
                px[z++] = pal[v][0]; px[z++] = pal[v][1]; px[z++] = pal[v][2];
            }
            r.skip(row_bytes + pad);
        }
    } else {
        r.skip(offset - extra - hsz);
        int row_bytes = 0;
        if (bpp == 24) row_bytes = 3 * w; else if (bpp == 16) row_bytes = 2 * w; else if (bpp != 32) throw LjError(LJ_ERR_PARSE, "corrupt BMP (bits per pixel): " + name);
        const int pad = (-row_bytes) & 3;
        const bool bytes_bgr = bpp == 24 || (bpp == 32 && mb == 0xffu && mg == 0xff00u && mr == 0x00ff0000u && ma == 0xff000000u);
        auto high_bit = [](uint32_t v) { int n = -1; while (v) { n++; v >>= 1; } return n; };
        auto bit_count = [](uint32_t v) { int n = 0; while (v) { n += (int)(v & 1u); v >>= 1; } return n; };
        int rs = 0, gs = 0, bs = 0, rc = 0, gc = 0, bc = 0;
        if (!bytes_bgr) {
            if (!mr || !mg || !mb) throw LjError(LJ_ERR_PARSE, "corrupt BMP (masks): " + name);
            rs = high_bit(mr) - 7; rc = bit_count(mr); gs = high_bit(mg) - 7; gc = bit_count(mg); bs = high_bit(mb) - 7; bc = bit_count(mb);
            if (rc > 8 || gc > 8 || bc > 8 || bit_count(ma) > 8) throw LjError(LJ_ERR_PARSE, "corrupt BMP (masks wider than 8 bits): " + name);
        }
        // an n-bit field, moved so that its top bit is bit 7, widened to 8 bits by repeating its bit pattern
        auto widen = [](uint32_t v, int shift, int bits) {
            static const uint32_t mul[9] = {0, 0xff, 0x55, 0x49, 0x11, 0x21, 0x41, 0x81, 0x01};
            static const uint32_t shr[9] = {0, 0, 0, 1, 0, 2, 4, 6, 0};
            v = shift < 0 ? v << -shift : v >> shift;
            v >>= (8 - bits);
            return (uint8_t)((v * mul[bits]) >> shr[bits]);
        };
        for (int y = 0; y < h; y++) {
            for (int x = 0; x < w; x++) {
                if (bytes_bgr) { px[z + 2] = r.u8(); px[z + 1] = r.u8(); px[z] = r.u8(); z += 3; if (bpp == 32) r.u8(); }
                else {
                    const uint32_t v = bpp == 16 ? r.u16() : r.u32();
                    px[z++] = widen(v & mr, rs, rc); px[z++] = widen(v & mg, gs, gc); px[z++] = widen(v & mb, bs, bc);
                }
            }
            r.skip(pad);
        }
    }
    if (bottom_up)
        for (int y = 0; y < (h >> 1); y++)
            for (size_t k = 0; k < (size_t)w * 3; k++) std::swap(px[(size_t)y * w * 3 + k], px[(size_t)(h - 1 - y) * w * 3 + k]);
    return finish_ldr(px, w, h, 3, channels);
}

HostImage read_psd(const std::vector<uint8_t> &file, const std::string &name, int channels) {
    Reader r{file, name};
    auto be16 = [&]() { const uint32_t a = r.u8(); return (a << 8) | r.u8(); };
    auto be32 = [&]() { const uint32_t a = be16(); return (a << 16) | be16(); };
    if (be32() != 0x38425053u) throw LjError(LJ_ERR_PARSE, "not a PSD file: " + name);
    if (be16() != 1) throw LjError(LJ_ERR_UNSUPPORTED, "PSD version is not 1: " + name);
    r.skip(6);
    const int n_ch = (int)be16();
    if (n_ch > 16) throw LjError(LJ_ERR_UNSUPPORTED, "PSD with more than 16 channels: " + name);
    const long long h = (int32_t)be32(), w = (int32_t)be32();
    check_image_size(w, h, file.size(), name);
    const int depth = (int)be16();
    if (depth != 8 && depth != 16) throw LjError(LJ_ERR_UNSUPPORTED, "PSD bit depth is not 8 or 16: " + name);
    if (be16() != 3) throw LjError(LJ_ERR_UNSUPPORTED, "PSD is not in RGB colour mode: " + name);
    r.skip(be32()); r.skip(be32()); r.skip(be32());   // mode data, image resources, layers
    const int compression = (int)be16();
    if (compression > 1) throw LjError(LJ_ERR_UNSUPPORTED, "PSD compression other than raw / PackBits: " + name);
    const size_t n = (size_t)w * (size_t)h;
    std::vector<uint8_t> px(n * 4);
    if (compression) r.skip(h * n_ch * 2);   // the per-row byte counts: the planes are decoded as one run of w * h samples each
    for (int c = 0; c < 4; c++) {
        if (c >= n_ch) { for (size_t i = 0; i < n; i++) px[4 * i + c] = c == 3 ? 255 : 0; continue; }
        if (!compression) {
            for (size_t i = 0; i < n; i++) px[4 * i + c] = depth == 16 ? (uint8_t)(be16() >> 8) : r.u8();
            continue;
        }
        size_t count = 0;
        while (count < n) {
            int len = r.u8();
            if (len == 128) { if (r.p >= file.size()) throw LjError(LJ_ERR_PARSE, "truncated PSD plane: " + name); continue; }
            if (len < 128) {
                len++;
                if ((size_t)len > n - count) throw LjError(LJ_ERR_PARSE, "corrupt PSD run: " + name);
                for (int k = 0; k < len; k++) px[4 * (count++) + c] = r.u8();
            } else {
                len = 257 - len;
                if ((size_t)len > n - count) throw LjError(LJ_ERR_PARSE, "corrupt PSD run: " + name);
                const uint8_t v = r.u8();
                for (int k = 0; k < len; k++) px[4 * (count++) + c] = v;
            }
            if (r.p >= file.size() && count < n) throw LjError(LJ_ERR_PARSE, "truncated PSD plane: " + name);
        }
    }
    if (n_ch >= 4)
        for (size_t i = 0; i < n; i++) {
            uint8_t *q = &px[4 * i];
            if (q[3] != 0 && q[3] != 255) {   // colours stored matted against white: c = (c - 255 (1 - a)) / a, in the reference loader's float steps
                const float a = q[3] / 255.0f, ra = 1.0f / a, inv_a = 255.0f * (1 - ra);
                for (int c = 0; c < 3; c++) q[c] = (unsigned char)(q[c] * ra + inv_a);
            }
        }
    return finish_ldr(px, (int)w, (int)h, 4, channels);
}

HostImage read_gif(const std::vector<uint8_t> &file, const std::string &name, int channels) {
    Reader r{file, name};
    if (r.u8() != 'G' || r.u8() != 'I' || r.u8() != 'F' || r.u8() != '8') throw LjError(LJ_ERR_PARSE, "not a GIF file: " + name);
    const int ver = r.u8();
    if ((ver != '7' && ver != '9') || r.u8() != 'a') throw LjError(LJ_ERR_PARSE, "not a GIF file: " + name);
    const int W = (int)r.u16(), H = (int)r.u16(), flags = r.u8(), bgindex = r.u8();
    r.u8();
    check_image_size(W, H, file.size(), name);
    uint8_t gpal[256][4] = {}, lpal[256][4] = {};   // entries as R, G, B, alpha
    auto read_palette = [&](uint8_t pal[256][4], int n, int transparent) {
        for (int i = 0; i < n; i++) { pal[i][0] = r.u8(); pal[i][1] = r.u8(); pal[i][2] = r.u8(); pal[i][3] = transparent == i ? 0 : 255; }
    };
    if (flags & 0x80) read_palette(gpal, 2 << (flags & 7), -1);
    std::vector<uint8_t> px((size_t)W * H * 4, 0), touched((size_t)W * H, 0);
    int eflags = 0, transparent = -1;
    for (;;) {
        if (r.p >= file.size()) throw LjError(LJ_ERR_PARSE, "GIF without an image: " + name);
        const int tag = r.u8();
        if (tag == 0x21) {
            const int ext = r.u8();
            if (ext == 0xF9) {   // graphic control: the transparent index (applies to the global palette at once, to a local one when it is read)
                const int len = r.u8();
                if (len != 4) throw LjError(LJ_ERR_PARSE, "corrupt GIF (graphic control block): " + name);
                eflags = r.u8(); r.u16();
                if (transparent >= 0) gpal[transparent][3] = 255;
                if (eflags & 1) { transparent = r.u8(); gpal[transparent][3] = 0; } else { r.u8(); transparent = -1; }
            }
            for (int len; (len = r.u8()) != 0;) { if (r.p >= file.size()) throw LjError(LJ_ERR_PARSE, "truncated GIF extension: " + name); r.skip(len); }
            continue;
        }
        if (tag == 0x3B) throw LjError(LJ_ERR_PARSE, "GIF without an image: " + name);
        if (tag != 0x2C) throw LjError(LJ_ERR_PARSE, "corrupt GIF (block type): " + name);
        const int x0 = (int)r.u16(), y0 = (int)r.u16(), w = (int)r.u16(), h = (int)r.u16();
        if (x0 + w > W || y0 + h > H) throw LjError(LJ_ERR_PARSE, "corrupt GIF (frame outside the screen): " + name);
        const int lflags = r.u8();
        const uint8_t (*pal)[4];
        if (lflags & 0x80) { read_palette(lpal, 2 << (lflags & 7), (eflags & 1) ? transparent : -1); pal = lpal; }
        else if (flags & 0x80) pal = gpal;
        else throw LjError(LJ_ERR_PARSE, "GIF without a colour table: " + name);
        // rows in file order: top to bottom, or the four interlace passes (every 8th from 0, every 8th from 4, every 4th from 2, odd rows)
        std::vector<int> rows;
        if (lflags & 0x40) { for (int pass = 0; pass < 4; pass++) { const int start[4] = {0, 4, 2, 1}, step[4] = {8, 8, 4, 2}; for (int y = start[pass]; y < h; y += step[pass]) rows.push_back(y); } }
        else for (int y = 0; y < h; y++) rows.push_back(y);
        size_t n_out = 0;
        const size_t n_px = (size_t)w * h;
        auto emit = [&](int index) {
            if (n_out >= n_px) return;
            const int x = x0 + (int)(n_out % (size_t)w), y = y0 + rows[n_out / (size_t)w];
            n_out++;
            const size_t at = (size_t)y * W + x;
            touched[at] = 1;
            if (pal[index][3] > 128) { px[4 * at] = pal[index][0]; px[4 * at + 1] = pal[index][1]; px[4 * at + 2] = pal[index][2]; px[4 * at + 3] = pal[index][3]; }
        };
        // ---- LZW over the data sub-blocks
        const int lzw_cs = r.u8();
        if (lzw_cs > 12) throw LjError(LJ_ERR_PARSE, "corrupt GIF (code size): " + name);
        struct Code { int16_t prefix; uint8_t first, suffix; };
        std::vector<Code> codes(8192);
        const int clear = 1 << lzw_cs;
        for (int i = 0; i < clear; i++) { codes[i].prefix = -1; codes[i].first = codes[i].suffix = (uint8_t)i; }
        int codesize = lzw_cs + 1, codemask = (1 << codesize) - 1, avail = clear + 2, oldcode = -1, valid_bits = 0, len = 0;
        uint32_t bits = 0;
        bool seen_clear = false, done = false;
        std::vector<uint8_t> chain;
        while (!done) {
            if (valid_bits < codesize) {
                if (len == 0) { len = r.u8(); if (len == 0) break; }
                if (r.p >= file.size()) throw LjError(LJ_ERR_PARSE, "truncated GIF image data: " + name);
                len--;
                bits |= (uint32_t)r.u8() << valid_bits; valid_bits += 8;
                continue;
            }
            const int code = (int)(bits & (uint32_t)codemask);
            bits >>= codesize; valid_bits -= codesize;
            if (code == clear) { codesize = lzw_cs + 1; codemask = (1 << codesize) - 1; avail = clear + 2; oldcode = -1; seen_clear = true; }
            else if (code == clear + 1) { r.skip(len); for (int l; (l = r.u8()) > 0;) r.skip(l); done = true; }
            else if (code <= avail) {
                if (!seen_clear) throw LjError(LJ_ERR_PARSE, "corrupt GIF (no clear code): " + name);
                if (oldcode >= 0) {
                    if (avail >= 8192) throw LjError(LJ_ERR_PARSE, "corrupt GIF (too many codes): " + name);
                    Code &c = codes[avail++];
                    c.prefix = (int16_t)oldcode; c.first = codes[oldcode].first; c.suffix = (code == avail) ? c.first : codes[code].first;
                } else if (code == avail) throw LjError(LJ_ERR_PARSE, "corrupt GIF (code before its definition): " + name);
                chain.clear();
                for (int c = code; c >= 0; c = codes[c].prefix) chain.push_back(codes[c].suffix);
                for (size_t k = chain.size(); k-- > 0;) emit(chain[k]);
                if ((avail & codemask) == 0 && avail <= 0x0FFF) { codesize++; codemask = (1 << codesize) - 1; }
                oldcode = code;
            } else throw LjError(LJ_ERR_PARSE, "corrupt GIF (code out of range): " + name);
        }
        if (bgindex > 0)   // what the frame did not cover takes the background entry — copied blue first into an R, G, B pixel by the reference's loader
            for (size_t i = 0; i < (size_t)W * H; i++)
                if (!touched[i]) { px[4 * i] = gpal[bgindex][2]; px[4 * i + 1] = gpal[bgindex][1]; px[4 * i + 2] = gpal[bgindex][0]; px[4 * i + 3] = 255; }
        return finish_ldr(px, W, H, 4, channels);
    }
}

HostImage read_pic(const std::vector<uint8_t> &file, const std::string &name, int channels) {
    static const uint8_t magic[4] = {0x53, 0x80, 0xF6, 0x34};
    if (file.size() < 104 || memcmp(file.data(), magic, 4) != 0 || memcmp(file.data() + 88, "PICT", 4) != 0) throw LjError(LJ_ERR_PARSE, "not a Softimage PIC file: " + name);
    Reader r{file, name};
    r.skip(92);
    auto be16 = [&]() { const uint32_t a = r.u8(); return (int)((a << 8) | r.u8()); };
    const int w = be16(), h = be16();
    r.skip(8);   // ratio, fields, pad
    check_image_size(w, h, file.size(), name);
    struct Packet { int type, channel; };
    std::vector<Packet> packets;
    for (bool chained = true; chained;) {
        if (packets.size() == 10) throw LjError(LJ_ERR_PARSE, "corrupt PIC (too many packets): " + name);
        chained = r.u8() != 0;
        const int size = r.u8();
        Packet pk; pk.type = r.u8(); pk.channel = r.u8();
        if (r.p >= file.size()) throw LjError(LJ_ERR_PARSE, "truncated PIC (packets): " + name);
        if (size != 8) throw LjError(LJ_ERR_UNSUPPORTED, "PIC channels that are not 8 bits wide: " + name);
        if (pk.type > 2) throw LjError(LJ_ERR_UNSUPPORTED, "PIC packet compression type: " + name);
        packets.push_back(pk);
    }
    std::vector<uint8_t> px((size_t)w * h * 4, 0xff);
    auto read_value = [&](int channel, uint8_t *dst) {   // the channels of the mask, most significant bit = red
        for (int i = 0, mask = 0x80; i < 4; i++, mask >>= 1)
            if (channel & mask) { if (r.p >= file.size()) throw LjError(LJ_ERR_PARSE, "truncated PIC (pixel data): " + name); dst[i] = r.u8(); }
    };
    auto copy_value = [](int channel, uint8_t *dst, const uint8_t *src) { for (int i = 0, mask = 0x80; i < 4; i++, mask >>= 1) if (channel & mask) dst[i] = src[i]; };
    for (int y = 0; y < h; y++)
        for (const Packet &pk : packets) {
            uint8_t *dst = &px[(size_t)y * w * 4];
            if (pk.type == 0) { for (int x = 0; x < w; x++, dst += 4) read_value(pk.channel, dst); continue; }
            for (int left = w; left > 0;) {
                int count = r.u8();
                if (r.p >= file.size()) throw LjError(LJ_ERR_PARSE, "truncated PIC (run): " + name);
                uint8_t value[4] = {0, 0, 0, 0};
                if (pk.type == 1) {   // pure run-length: count, value
                    if (count == 0) throw LjError(LJ_ERR_PARSE, "corrupt PIC (empty run): " + name);
                    if (count > left) count = left;
                    read_value(pk.channel, value);
                    for (int i = 0; i < count; i++, dst += 4) copy_value(pk.channel, dst, value);
                } else if (count >= 128) {   // mixed: a repeated value ...
                    count = count == 128 ? be16() : count - 127;
                    if (count > left) throw LjError(LJ_ERR_PARSE, "corrupt PIC (scan line overrun): " + name);
                    read_value(pk.channel, value);
                    for (int i = 0; i < count; i++, dst += 4) copy_value(pk.channel, dst, value);
                } else {   // ... or count + 1 literal values
                    count++;
                    if (count > left) throw LjError(LJ_ERR_PARSE, "corrupt PIC (scan line overrun): " + name);
                    for (int i = 0; i < count; i++, dst += 4) read_value(pk.channel, dst);
                }
                left -= count;
            }
        }
    return finish_ldr(px, w, h, 4, channels);
}

} // namespace lj
