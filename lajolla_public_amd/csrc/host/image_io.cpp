// Texture image input of the front end: imread3 / imread1 (image.cpp:28-133 in the reference, which
// delegates to stb_image and tinyexr).  Returned data is float, row-major, y=0 at the top.
//
// Decoders implemented here, from the format specifications (no third-party code):
//   .pfm          portable float map (colour "PF" / grey "Pf"), bottom-up, endianness by the sign of the scale
//   .jpg / .jpeg  baseline and progressive JPEG (jpeg_decode.cpp), then the LDR -> linear conversion stb's stbi_loadf applies:
//                 (float) pow(v / 255.0f, 2.2f)  (stb_image.h:1553,1849) — what the reference's ImageTextures hold.
//   .png / .hdr   PNG (all colour types and bit depths, Adam7) and Radiance RGBE (png_decode.cpp), with stb's conventions
//   .tga / .bmp / .psd / .gif / .pic   Truevision TGA, Windows BMP, Photoshop PSD, GIF (first frame) and Softimage PIC (tga_bmp_decode.cpp), likewise
//   .exr          single-part scan-line or tiled OpenEXR, HALF / FLOAT / UINT channels, NONE / ZIPS / ZIP / PIZ (exr_decode.cpp);
//                 three channels = R, G, B; one channel = their mean (image.cpp:70-72)
// Other formats fail loudly.
#include "host_scene.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace lj {

std::vector<uint8_t> decode_jpeg_rgb8(const std::vector<uint8_t> &file, int &width, int &height, const std::string &name, std::vector<uint8_t> *grey);
HostImage decode_exr_rgb(const std::vector<uint8_t> &file, const std::string &name);
HostImage read_png(const std::vector<uint8_t> &file, const std::string &name, int channels);   // png_decode.cpp
HostImage read_hdr(const std::vector<uint8_t> &file, const std::string &name, int channels);
HostImage read_tga(const std::vector<uint8_t> &file, const std::string &name, int channels);   // tga_bmp_decode.cpp
HostImage read_bmp(const std::vector<uint8_t> &file, const std::string &name, int channels);
HostImage read_psd(const std::vector<uint8_t> &file, const std::string &name, int channels);
HostImage read_gif(const std::vector<uint8_t> &file, const std::string &name, int channels);
HostImage read_pic(const std::vector<uint8_t> &file, const std::string &name, int channels);

namespace {

std::string ext_of(const std::string &f) {
    size_t d = f.find_last_of('.');
    if (d == std::string::npos) return "";
    std::string e = f.substr(d);
    for (auto &c : e) c = (char)std::tolower((unsigned char)c);
    return e;
}

HostImage read_pfm(const std::string &filename) {
    std::ifstream f(filename, std::ios::binary);
    if (!f) throw LjError(LJ_ERR_IO, "cannot open image: " + filename);
    std::string magic; int w = 0, h = 0; double scale = 0;
    f >> magic >> w >> h >> scale;
    f.get();  // the single whitespace byte that ends the header
    int ch = magic == "PF" ? 3 : (magic == "Pf" ? 1 : 0);
    if (!f || ch == 0 || w <= 0 || h <= 0 || scale == 0) throw LjError(LJ_ERR_PARSE, "malformed PFM header: " + filename);
    { const std::streampos at = f.tellg(); f.seekg(0, std::ios::end); const std::streampos end = f.tellg(); f.seekg(at); check_image_size(w, h, (size_t)end, filename); }
    HostImage img; img.width = w; img.height = h; img.channels = ch;
    img.data.resize((size_t)w * h * ch);
    std::vector<float> row((size_t)w * ch);
    bool little = scale < 0;
    for (int y = 0; y < h; y++) {
        f.read((char *)row.data(), (std::streamsize)(row.size() * 4));
        if (!f) throw LjError(LJ_ERR_PARSE, "truncated PFM: " + filename);
        if (!little) for (auto &v : row) { unsigned char *p = (unsigned char *)&v; std::swap(p[0], p[3]); std::swap(p[1], p[2]); }
        // PFM stores the bottom row first
        std::copy(row.begin(), row.end(), img.data.begin() + (size_t)(h - 1 - y) * w * ch);
    }
    return img;
}

HostImage convert_channels(const HostImage &src, int channels) {
    if (src.channels == channels) return src;
    HostImage out; out.width = src.width; out.height = src.height; out.channels = channels;
    size_t n = (size_t)src.width * src.height;
    out.data.resize(n * channels);
    for (size_t i = 0; i < n; i++) {
        if (channels == 3) { float v = src.data[i * src.channels]; out.data[3 * i] = out.data[3 * i + 1] = out.data[3 * i + 2] = v; }
        else out.data[i] = src.data[i * src.channels];  // "the first channel is used" (image.h:44-45)
    }
    return out;
}

} // namespace

HostImage read_image(const std::string &filename, int channels) {
    std::string ext = ext_of(filename);
    if (ext == ".pfm") return convert_channels(read_pfm(filename), channels);
    if (ext == ".exr") {
        std::ifstream f(filename, std::ios::binary);
        if (!f) throw LjError(LJ_ERR_IO, "cannot open image: " + filename);
        std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        bytes.shrink_to_fit();   // capacity == size: a sanitizer build sees any read past the file
        HostImage rgb = decode_exr_rgb(bytes, filename);
        if (channels != 1) return rgb;
        HostImage g; g.width = rgb.width; g.height = rgb.height; g.channels = 1;
        g.data.resize((size_t)rgb.width * rgb.height);
        for (size_t i = 0; i < g.data.size(); i++) {   // image.cpp:70-72: the mean of tinyexr's three FLOATS, formed in float arithmetic
            const float sum = (rgb.data[3 * i] + rgb.data[3 * i + 1]) + rgb.data[3 * i + 2];
            g.data[i] = sum / 3;
        }
        return g;
    }
    if (ext == ".jpg" || ext == ".jpeg") {
        std::ifstream f(filename, std::ios::binary);
        if (!f) throw LjError(LJ_ERR_IO, "cannot open image: " + filename);
        std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        bytes.shrink_to_fit();   // capacity == size: a sanitizer build sees any read past the file
        int w = 0, h = 0;
        std::vector<uint8_t> y8;
        std::vector<uint8_t> rgb = decode_jpeg_rgb8(bytes, w, h, filename, channels == 1 ? &y8 : nullptr);
        HostImage img; img.width = w; img.height = h; img.channels = channels == 1 ? 1 : 3;
        img.data.resize((size_t)w * h * img.channels);
        // stbi_loadf(..., req_comp): 3 -> RGB; 1 -> the Y plane of a YCbCr file (or the luma of an RGB-coded one): jpeg_decode.cpp
        const float gamma = 2.2f;
        for (size_t i = 0; i < (size_t)w * h; i++) {
            if (img.channels == 3) { for (int c = 0; c < 3; c++) img.data[3 * i + c] = (float)std::pow((double)(rgb[3 * i + c] / 255.0f), (double)gamma); }
            else img.data[i] = (float)std::pow((double)(y8[i] / 255.0f), (double)gamma);
        }
        return img;
    }
    if (ext == ".png" || ext == ".hdr" || ext == ".tga" || ext == ".bmp" || ext == ".psd" || ext == ".gif" || ext == ".pic") {
        std::ifstream f(filename, std::ios::binary);
        if (!f) throw LjError(LJ_ERR_IO, "cannot open image: " + filename);
        std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        bytes.shrink_to_fit();   // capacity == size: a sanitizer build sees any read past the file
        if (ext == ".png") return read_png(bytes, filename, channels);
        if (ext == ".hdr") return read_hdr(bytes, filename, channels);
        if (ext == ".psd") return read_psd(bytes, filename, channels);
        if (ext == ".gif") return read_gif(bytes, filename, channels);
        if (ext == ".pic") return read_pic(bytes, filename, channels);
        return ext == ".tga" ? read_tga(bytes, filename, channels) : read_bmp(bytes, filename, channels);
    }
    // (every extension the reference passes to stb_image or tinyexr, image.cpp:31-38,54, is decoded above)
    throw LjError(LJ_ERR_UNSUPPORTED, "image format '" + ext + "' is not decoded by this build (JPEG, PNG, TGA, BMP, PSD, GIF, PIC, Radiance HDR, OpenEXR and PFM are): " + filename);
}

// ------------------------------------------------------------------ imwrite (image.cpp:135-173)
namespace {

uint16_t float_to_half(float f) {   // round to nearest even, overflow to infinity, subnormals kept
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u | (m >> 13) : 0));   // inf / NaN
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const int shift = 14 - e;   // 14..24
        uint32_t h = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (h & 1))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;   // a carry into the exponent is the right answer, up to inf
    return (uint16_t)(sign | h);
}

} // namespace

void write_image(const std::string &filename, int width, int height, const float *rgb) {
    if (width <= 0 || height <= 0 || !rgb) throw LjError(LJ_ERR_INVALID_ARG, "write_image: empty image");
    const std::string ext = ext_of(filename);
    if (ext != ".pfm" && ext != ".exr") throw LjError(LJ_ERR_UNSUPPORTED, "images are written as .pfm or .exr: " + filename);
    std::ofstream f(filename, std::ios::binary);
    if (!f) throw LjError(LJ_ERR_IO, "cannot create image: " + filename);
    const size_t n = (size_t)width * height;
    if (ext == ".pfm") {
        f << "PF\n" << width << " " << height << "\n-1\n";
        f.write((const char *)rgb, (std::streamsize)(n * 3 * sizeof(float)));   // little-endian host, rows top to bottom
    } else {
        auto put32 = [&](uint32_t v) { const unsigned char b[4] = {(unsigned char)v, (unsigned char)(v >> 8), (unsigned char)(v >> 16), (unsigned char)(v >> 24)}; f.write((const char *)b, 4); };
        auto put64 = [&](uint64_t v) { put32((uint32_t)v); put32((uint32_t)(v >> 32)); };
        auto attr = [&](const char *name, const char *type, const std::string &value) {
            f.write(name, (std::streamsize)strlen(name) + 1); f.write(type, (std::streamsize)strlen(type) + 1);
            put32((uint32_t)value.size()); f.write(value.data(), (std::streamsize)value.size());
        };
        auto i32s = [](std::initializer_list<int32_t> v) { std::string s; for (int32_t x : v) { uint32_t u = (uint32_t)x; for (int k = 0; k < 4; k++) s.push_back((char)(u >> (8 * k))); } return s; };
        auto f32s = [](std::initializer_list<float> v) { std::string s; for (float x : v) { uint32_t u; memcpy(&u, &x, 4); for (int k = 0; k < 4; k++) s.push_back((char)(u >> (8 * k))); } return s; };
        std::ostringstream header;
        put32(20000630u); put32(2u);
        std::string chlist;
        for (const char *c : {"B", "G", "R"}) { chlist += c; chlist.push_back('\0'); chlist += i32s({1}); chlist += std::string(4, '\0'); chlist += i32s({1, 1}); }
        chlist.push_back('\0');
        attr("channels", "chlist", chlist);
        attr("compression", "compression", std::string(1, '\0'));
        attr("dataWindow", "box2i", i32s({0, 0, width - 1, height - 1}));
        attr("displayWindow", "box2i", i32s({0, 0, width - 1, height - 1}));
        attr("lineOrder", "lineOrder", std::string(1, '\0'));
        attr("pixelAspectRatio", "float", f32s({1.0f}));
        attr("screenWindowCenter", "v2f", f32s({0.0f, 0.0f}));
        attr("screenWindowWidth", "float", f32s({1.0f}));
        f.put('\0');
        const uint64_t line_bytes = (uint64_t)width * 3 * 2, table = (uint64_t)f.tellp() + (uint64_t)height * 8;
        for (int y = 0; y < height; y++) put64(table + (uint64_t)y * (8 + line_bytes));
        std::vector<uint16_t> line((size_t)width * 3);
        for (int y = 0; y < height; y++) {
            put32((uint32_t)y); put32((uint32_t)line_bytes);
            const float *row = rgb + (size_t)y * width * 3;
            for (int x = 0; x < width; x++) {   // channel rows in alphabetical order: B, G, R
                line[x] = float_to_half(row[3 * x + 2]); line[width + x] = float_to_half(row[3 * x + 1]); line[2 * (size_t)width + x] = float_to_half(row[3 * x]);
            }
            f.write((const char *)line.data(), (std::streamsize)line_bytes);
        }
    }
    if (!f) throw LjError(LJ_ERR_IO, "write failed: " + filename);
}

} // namespace lj
