// PNG and Radiance .hdr texture input, written from the format specifications to reproduce what the reference's image loader
// returns for them — imread1 / imread3 hand .png and .hdr files to stb_image's stbi_loadf (image.cpp:28-133), so the
// conventions below are stb's:
//   PNG  colour types 0 / 2 / 3 / 4 / 6, bit depths 1 - 16, Adam7 interlacing; the alpha channel and tRNS are dropped (no
//        premultiplication); 16-bit samples keep their high byte; 1 / 2 / 4-bit grey is scaled to 8 bits (x255, x85, x17); a
//        3-channel read replicates grey, a 1-channel read of colour takes the luma (77 R + 150 G + 29 B) >> 8 — computed on the
//        16-bit samples first when the file has them, as stb does; gAMA / sRGB / iCCP chunks are ignored; then the LDR -> linear
//        conversion (float) pow(v / 255.0f, 2.2f).  Checksums are not verified (stb does not either).
//   HDR  "#?RADIANCE" / "#?RGBE", FORMAT=32-bit_rle_rgbe, "-Y h +X w"; new-style run-length scanlines or flat RGBE; a pixel is
//        mantissa * 2^(e - 136), zero when e == 0; a 1-channel read averages (r + g + b) * scale / 3.
#include "host_scene.h"
#include <cmath>
#include <cstring>
#include <zlib.h>

namespace lj {

namespace {

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// one (sub-)image of `w` x `h` pixels: unfilter `src` (h scanlines of 1 + stride bytes) in place into `out` (h * stride)
void unfilter(const uint8_t *src, size_t avail, int w, int h, int bits_per_pixel, std::vector<uint8_t> &out, const std::string &name) {
    const size_t stride = ((size_t)w * bits_per_pixel + 7) / 8, bpp = (size_t)std::max(1, bits_per_pixel / 8);
    if (avail < (stride + 1) * (size_t)h) throw LjError(LJ_ERR_PARSE, "PNG image data too short: " + name);
    out.assign(stride * (size_t)h, 0);
    for (int y = 0; y < h; y++) {
        const uint8_t *row = src + (stride + 1) * (size_t)y;
        const int ft = row[0];
        uint8_t *cur = out.data() + stride * (size_t)y;
        const uint8_t *up = y ? cur - stride : nullptr;
        if (ft > 4) throw LjError(LJ_ERR_PARSE, "PNG filter type " + std::to_string(ft) + " does not exist: " + name);
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int v = row[1 + i];
            switch (ft) { case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += paeth(a, b, c); break; default: break; }
            cur[i] = (uint8_t)v;
        }
    }
}

} // namespace

// -> samples as 16-bit values (8-bit files: the byte; 16-bit files: the full sample), `channels_out` of them per pixel in
// {1: grey, 3: RGB} as the file holds them (alpha dropped); `sixteen`: the file is 16-bit
void decode_png(const std::vector<uint8_t> &file, const std::string &name, int &width, int &height, int &channels_out, bool &sixteen, std::vector<uint16_t> &samples) {
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) throw LjError(LJ_ERR_PARSE, "not a PNG file: " + name);
    size_t p = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool have_ihdr = false, done = false;
    while (!done && p + 12 <= file.size()) {
        const uint32_t len = be32(&file[p]);
        const std::string type((const char *)&file[p + 4], 4);
        if (p + 12 + (size_t)len > file.size()) throw LjError(LJ_ERR_PARSE, "truncated PNG chunk: " + name);
        const uint8_t *d = &file[p + 8];
        if (type == "IHDR") {
            if (len != 13) throw LjError(LJ_ERR_PARSE, "bad IHDR: " + name);
            width = (int)be32(d); height = (int)be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
            check_image_size(width, height, file.size(), name);
            if (d[10] != 0 || d[11] != 0 || interlace > 1) throw LjError(LJ_ERR_PARSE, "bad PNG compression / filter / interlace method: " + name);
            const bool ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                            ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
            if (!ok) throw LjError(LJ_ERR_PARSE, "bad PNG colour type / bit depth: " + name);
            have_ihdr = true;
        } else if (type == "PLTE") plte.assign(d, d + len);
        else if (type == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (type == "IEND") done = true;
        p += 12 + (size_t)len;
    }
    if (!have_ihdr || idat.empty()) throw LjError(LJ_ERR_PARSE, "PNG without IHDR / IDAT: " + name);
    if (ctype == 3 && (plte.empty() || plte.size() % 3)) throw LjError(LJ_ERR_PARSE, "paletted PNG without a palette: " + name);
    const int file_ch = ctype == 0 ? 1 : (ctype == 2 ? 3 : (ctype == 3 ? 1 : (ctype == 4 ? 2 : 4)));
    const int bpp_bits = file_ch * depth;
    // ---- inflate
    size_t raw_size = 0;
    static const int x0[7] = {0, 4, 0, 2, 0, 1, 0}, y0[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
    auto pass_dim = [&](int ps, int &pw, int &ph) { pw = (width - x0[ps] + dx[ps] - 1) / dx[ps]; ph = (height - y0[ps] + dy[ps] - 1) / dy[ps]; };
    if (!interlace) raw_size = (((size_t)width * bpp_bits + 7) / 8 + 1) * (size_t)height;
    else for (int ps = 0; ps < 7; ps++) { int pw, ph; pass_dim(ps, pw, ph); if (pw > 0 && ph > 0) raw_size += (((size_t)pw * bpp_bits + 7) / 8 + 1) * (size_t)ph; }
    std::vector<uint8_t> raw(raw_size);
    {
        z_stream zs{};
        if (inflateInit(&zs) != Z_OK) throw LjError(LJ_ERR_INTERNAL, "zlib init failed");
        zs.next_in = idat.data(); zs.avail_in = (uInt)idat.size(); zs.next_out = raw.data(); zs.avail_out = (uInt)raw.size();
        const int rc = inflate(&zs, Z_FINISH);
        const size_t got = zs.total_out;
        inflateEnd(&zs);
        if ((rc != Z_STREAM_END && rc != Z_BUF_ERROR && rc != Z_OK) || got < raw_size) throw LjError(LJ_ERR_PARSE, "corrupt PNG image data: " + name);
    }
    // ---- unfilter (+ de-interlace) into one sample array [y][x][file_ch] of 16-bit values
    std::vector<uint16_t> px((size_t)width * height * file_ch);
    auto unpack = [&](const std::vector<uint8_t> &rows, int pw, int ph, int ox, int oy, int sx, int sy) {
        const size_t stride = ((size_t)pw * bpp_bits + 7) / 8;
        for (int y = 0; y < ph; y++) for (int x = 0; x < pw; x++) for (int c = 0; c < file_ch; c++) {
            const uint8_t *row = rows.data() + stride * (size_t)y;
            uint16_t v;
            if (depth == 16) v = (uint16_t)((row[((size_t)x * file_ch + c) * 2] << 8) | row[((size_t)x * file_ch + c) * 2 + 1]);
            else if (depth == 8) v = row[(size_t)x * file_ch + c];
            else { const size_t bit = (size_t)x * depth; v = (uint16_t)((row[bit / 8] >> (8 - depth - (bit % 8))) & ((1 << depth) - 1)); }
            px[(((size_t)(oy + y * sy)) * width + (ox + x * sx)) * file_ch + c] = v;
        }
    };
    std::vector<uint8_t> rows;
    if (!interlace) { unfilter(raw.data(), raw.size(), width, height, bpp_bits, rows, name); unpack(rows, width, height, 0, 0, 1, 1); }
    else {
        size_t off = 0;
        for (int ps = 0; ps < 7; ps++) {
            int pw, ph; pass_dim(ps, pw, ph);
            if (pw <= 0 || ph <= 0) continue;
            unfilter(raw.data() + off, raw.size() - off, pw, ph, bpp_bits, rows, name);
            unpack(rows, pw, ph, x0[ps], y0[ps], dx[ps], dy[ps]);
            off += (((size_t)pw * bpp_bits + 7) / 8 + 1) * (size_t)ph;
        }
    }
    // ---- to grey or RGB samples
    sixteen = depth == 16;
    const size_t n = (size_t)width * height;
    if (ctype == 3) {
        channels_out = 3; samples.resize(n * 3);
        for (size_t i = 0; i < n; i++) {
            const size_t k = px[i];
            if (k * 3 + 2 >= plte.size()) throw LjError(LJ_ERR_PARSE, "PNG palette index out of range: " + name);
            for (int c = 0; c < 3; c++) samples[3 * i + c] = plte[3 * k + c];
        }
    } else if (ctype == 0 || ctype == 4) {
        channels_out = 1; samples.resize(n);
        const int scale = depth == 1 ? 255 : (depth == 2 ? 85 : (depth == 4 ? 17 : 1));
        for (size_t i = 0; i < n; i++) samples[i] = (uint16_t)(px[i * file_ch] * scale);
    } else {
        channels_out = 3; samples.resize(n * 3);
        for (size_t i = 0; i < n; i++) for (int c = 0; c < 3; c++) samples[3 * i + c] = px[i * file_ch + c];
    }
}

HostImage read_png(const std::vector<uint8_t> &file, const std::string &name, int channels) {
    int w = 0, h = 0, ch = 0; bool sixteen = false;
    std::vector<uint16_t> s;
    decode_png(file, name, w, h, ch, sixteen, s);
    HostImage img; img.width = w; img.height = h; img.channels = channels == 1 ? 1 : 3;
    img.data.resize((size_t)w * h * img.channels);
    auto to8 = [&](uint32_t v) { return (uint8_t)(sixteen ? ((v >> 8) & 0xff) : v); };   // stbi__convert_16_to_8
    auto lin = [](uint8_t v) { return (float)std::pow((double)(v / 255.0f), (double)2.2f); };   // stbi__ldr_to_hdr
    for (size_t i = 0; i < (size_t)w * h; i++) {
        if (img.channels == 3) {
            for (int c = 0; c < 3; c++) img.data[3 * i + c] = lin(to8(ch == 3 ? s[3 * i + c] : s[i]));
        } else {
            // stbi__compute_y / stbi__compute_y_16 on the samples as decoded (16-bit files: before the reduction to 8 bits)
            const uint32_t y = ch == 3 ? (((uint32_t)s[3 * i] * 77 + (uint32_t)s[3 * i + 1] * 150 + 29 * (uint32_t)s[3 * i + 2]) >> 8) : s[i];
            img.data[i] = lin(to8(sixteen ? (y & 0xffff) : (y & 0xff)));
        }
    }
    return img;
}

HostImage read_hdr(const std::vector<uint8_t> &file, const std::string &name, int channels) {
    size_t p = 0;
    auto line = [&]() { std::string s; while (p < file.size() && file[p] != '\n') s.push_back((char)file[p++]); if (p < file.size()) p++; return s; };
    const std::string magic = line();
    if (magic != "#?RADIANCE" && magic != "#?RGBE") throw LjError(LJ_ERR_PARSE, "not a Radiance HDR file: " + name);
    bool fmt = false;
    for (;;) {
        if (p >= file.size()) throw LjError(LJ_ERR_PARSE, "truncated HDR header: " + name);
        const std::string l = line();
        if (l.empty()) break;
        if (l == "FORMAT=32-bit_rle_rgbe") fmt = true;
    }
    if (!fmt) throw LjError(LJ_ERR_UNSUPPORTED, "HDR file is not FORMAT=32-bit_rle_rgbe: " + name);
    const std::string res = line();
    int w = 0, h = 0;
    if (sscanf(res.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) throw LjError(LJ_ERR_UNSUPPORTED, "HDR data layout other than '-Y h +X w': " + name);
    check_image_size(w, h, file.size(), name);
    HostImage img; img.width = w; img.height = h; img.channels = channels == 1 ? 1 : 3;
    img.data.assign((size_t)w * h * img.channels, 0.0f);
    auto need = [&](size_t n) { if (p + n > file.size()) throw LjError(LJ_ERR_PARSE, "truncated HDR pixel data: " + name); };
    auto convert = [&](float *out, const uint8_t *in) {   // stbi__hdr_convert
        if (in[3] != 0) {
            const float f1 = (float)std::ldexp(1.0f, (int)in[3] - (128 + 8));
            if (img.channels == 1) out[0] = (in[0] + in[1] + in[2]) * f1 / 3;
            else { out[0] = in[0] * f1; out[1] = in[1] * f1; out[2] = in[2] * f1; }
        } else { for (int c = 0; c < img.channels; c++) out[c] = 0.0f; }
    };
    auto flat_from = [&](size_t first_pixel, const uint8_t *first) {   // the rest of the image as uncompressed RGBE
        size_t i = first_pixel;
        if (first) { convert(&img.data[i * img.channels], first); i++; }
        for (; i < (size_t)w * h; i++) { need(4); convert(&img.data[i * img.channels], &file[p]); p += 4; }
    };
    if (w < 8 || w >= 32768) { flat_from(0, nullptr); return img; }
    std::vector<uint8_t> scan((size_t)w * 4);
    for (int y = 0; y < h; y++) {
        need(4);
        const uint8_t c1 = file[p], c2 = file[p + 1], l1 = file[p + 2], l2 = file[p + 3];
        if (c1 != 2 || c2 != 2 || (l1 & 0x80)) {
            // not run-length encoded: stb takes these four bytes as the first pixel and reads the whole image flat — only legal on the first row
            if (y != 0) throw LjError(LJ_ERR_PARSE, "HDR scanline without a run-length header: " + name);
            const uint8_t first[4] = {c1, c2, l1, l2};
            p += 4;
            flat_from(0, first);
            return img;
        }
        p += 4;
        if (((int)l1 << 8 | l2) != w) throw LjError(LJ_ERR_PARSE, "HDR scanline width mismatch: " + name);
        for (int k = 0; k < 4; k++) {
            int i = 0;
            while (i < w) {
                need(1);
                int count = file[p++];
                if (count > 128) {
                    count -= 128; need(1);
                    const uint8_t v = file[p++];
                    if (count == 0 || i + count > w) throw LjError(LJ_ERR_PARSE, "corrupt HDR run: " + name);
                    for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = v;
                } else {
                    if (count == 0 || i + count > w) throw LjError(LJ_ERR_PARSE, "corrupt HDR run: " + name);
                    need((size_t)count);
                    for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = file[p++];
                }
            }
        }
        for (int x = 0; x < w; x++) convert(&img.data[((size_t)y * w + x) * img.channels], &scan[(size_t)x * 4]);
    }
    return img;
}

} // namespace lj
