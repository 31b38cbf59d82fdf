// Texture image input of the front end: imread3 / imread1 (image.cpp:28-133 in the reference, which
// delegates to stb_image and tinyexr).  Returned data is float, row-major, y=0 at the top.
//
// Decoders implemented here, from the format specifications (no third-party code):
//   .pfm          portable float map (colour "PF" / grey "Pf"), bottom-up, endianness by the sign of the scale
//   .jpg / .jpeg  baseline JPEG (jpeg_decode.cpp), then the LDR -> linear conversion stb's stbi_loadf applies:
//                 (float) pow(v / 255.0f, 2.2f)  (stb_image.h:1553,1849) — what the reference's ImageTextures hold.
//   .exr          single-part scan-line OpenEXR, HALF / FLOAT / UINT channels, NONE / ZIPS / ZIP / PIZ (exr_decode.cpp);
//                 three channels = R, G, B; one channel = their mean (image.cpp:70-72)
// Other formats fail loudly.
#include "host_scene.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>

namespace lj {

std::vector<uint8_t> decode_jpeg_rgb8(const std::vector<uint8_t> &file, int &width, int &height, const std::string &name);
HostImage decode_exr_rgb(const std::vector<uint8_t> &file, const std::string &name);

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
        HostImage rgb = decode_exr_rgb(bytes, filename);
        if (channels != 1) return rgb;
        HostImage g; g.width = rgb.width; g.height = rgb.height; g.channels = 1;
        g.data.resize((size_t)rgb.width * rgb.height);
        for (size_t i = 0; i < g.data.size(); i++)   // image.cpp:70-72, in the reference's double arithmetic
            g.data[i] = (float)(((double)rgb.data[3 * i] + (double)rgb.data[3 * i + 1] + (double)rgb.data[3 * i + 2]) / 3);
        return g;
    }
    if (ext == ".jpg" || ext == ".jpeg") {
        std::ifstream f(filename, std::ios::binary);
        if (!f) throw LjError(LJ_ERR_IO, "cannot open image: " + filename);
        std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        int w = 0, h = 0;
        std::vector<uint8_t> rgb = decode_jpeg_rgb8(bytes, w, h, filename);
        HostImage img; img.width = w; img.height = h; img.channels = channels == 1 ? 1 : 3;
        img.data.resize((size_t)w * h * img.channels);
        // stbi_loadf(..., req_comp): 3 -> RGB; 1 -> luma (77 R + 150 G + 29 B) >> 8 of the decoded RGB (stb_image.h stbi__compute_y)
        const float gamma = 2.2f;
        for (size_t i = 0; i < (size_t)w * h; i++) {
            if (img.channels == 3) { for (int c = 0; c < 3; c++) img.data[3 * i + c] = (float)std::pow((double)(rgb[3 * i + c] / 255.0f), (double)gamma); }
            else { uint8_t y = (uint8_t)((rgb[3 * i] * 77 + rgb[3 * i + 1] * 150 + 29 * rgb[3 * i + 2]) >> 8); img.data[i] = (float)std::pow((double)(y / 255.0f), (double)gamma); }
        }
        return img;
    }
    throw LjError(LJ_ERR_UNSUPPORTED, "image format '" + ext + "' is not decoded by this build yet (SURVEY §8f-2): " + filename);
}

} // namespace lj
