// Texture image input of the front end: imread3 / imread1 (image.cpp:28-133 in the reference, which
// delegates to stb_image and tinyexr).  Returned data is float, row-major, y=0 at the top.
//
// Decoders implemented here, from the format specifications (no third-party code):
//   .pfm   portable float map (colour "PF" / grey "Pf"), bottom-up or top-down by the sign of the scale
// LDR formats are gamma-decoded the way stb's stbi_loadf does it: pow(v/255, 2.2) (stb_image.h:1553,1849),
// which is what the reference's ImageTextures hold.
//
// JPEG and OpenEXR (needed by scenes/sponza and scenes/disney_bsdf_test) are the next front-end row
// (SURVEY §8f-2); until they land, loading such a file fails loudly with LJ_ERR_UNSUPPORTED.
#include "host_scene.h"
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <sstream>

namespace lj {

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
    throw LjError(LJ_ERR_UNSUPPORTED, "image format '" + ext + "' is not decoded by this build yet (SURVEY §8f-2): " + filename);
}

} // namespace lj
