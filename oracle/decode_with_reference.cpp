// TEST/ASSET TOOL (authoring container only) — decodes image files with the REFERENCE's own loaders, imread3 and imread1
// (image.cpp:28-133 -> stb_image / tinyexr), and writes what they return as JSON: per file the size, a 64-bit FNV-1a hash of the
// float32 bit patterns of all texels, their sum and the first 48 values.  tests/golden/image_decode.json
// is this tool's output for tests/assets/images/*; the product's decoders (csrc/host/png_decode.cpp, image_io.cpp) are held to it
// bit for bit (tests/test_frontend.py).
//     oracle/_ref/decode_with_reference out.json file...
#include "image.h"
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static uint64_t fnv(const std::vector<float> &v) {
    uint64_t h = 1469598103934665603ull;
    for (float f : v) { uint32_t u; memcpy(&u, &f, 4); for (int k = 0; k < 4; k++) { h ^= (u >> (8 * k)) & 0xff; h *= 1099511628211ull; } }
    return h;
}
static void dump(FILE *o, const char *key, int w, int h, const std::vector<float> &v) {
    fprintf(o, "  \"%s\": {\"width\": %d, \"height\": %d, \"fnv1a64\": \"%016llx\", \"sum\": %.17g", key, w, h, (unsigned long long)fnv(v), [&] { double s = 0; for (float f : v) s += f; return s; }());
    fprintf(o, ", \"first_texels\": ["); for (size_t i = 0; i < v.size() && i < 48; i++) fprintf(o, "%s%.9g", i ? ", " : "", v[i]); fprintf(o, "]");
    fprintf(o, "}");
}
int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: decode_with_reference out.json file...\n"); return 1; }
    FILE *o = fopen(argv[1], "w");
    fprintf(o, "{\n \"generator\": \"oracle/decode_with_reference.cpp: the reference's imread3 / imread1 (image.cpp:28-133, stb_image) on tests/assets/images/*\",\n \"files\": {\n");
    for (int a = 2; a < argc; a++) {
        const std::string path = argv[a], base = path.substr(path.find_last_of('/') + 1);
        Image3 i3 = imread3(path); Image1 i1 = imread1(path);
        std::vector<float> v3, v1;
        for (const auto &p : i3.data) { v3.push_back((float)p.x); v3.push_back((float)p.y); v3.push_back((float)p.z); }
        for (double p : i1.data) v1.push_back((float)p);
        fprintf(o, "%s \"%s\": {\n", a > 2 ? ",\n" : "", base.c_str());
        dump(o, "imread3", i3.width, i3.height, v3); fprintf(o, ",\n"); dump(o, "imread1", i1.width, i1.height, v1);
        fprintf(o, "\n }");
    }
    fprintf(o, "\n }\n}\n");
    fclose(o);
    return 0;
}
