// TEST/ASSET TOOL (authoring container only) — decodes scenes/matpreview/envmap.exr (512x256, HALF, PIZ) with the
// reference's own imread3 (image.cpp:80-133 -> tinyexr) and writes the pixels as a PFM: the golden fixture
// tests/golden/matpreview_envmap_reference_decode.pfm that the product's OpenEXR reader is held to, bit for bit.
// Half -> float is exact, so the PFM holds bit-for-bit what the reference's TexturePool would hold at level 0.
#include "image.h"
#include <cstdio>
int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: convert_assets in.exr out.pfm\n"); return 1; }
    Image3 img = imread3(argv[1]);
    FILE *f = fopen(argv[2], "wb");
    fprintf(f, "PF\n%d %d\n-1.0\n", img.width, img.height);
    for (int y = img.height - 1; y >= 0; y--)  // PFM stores the bottom row first
        for (int x = 0; x < img.width; x++) { float v[3] = {(float)img(x, y)[0], (float)img(x, y)[1], (float)img(x, y)[2]}; fwrite(v, 4, 3, f); }
    fclose(f);
    printf("%d x %d\n", img.width, img.height);
    return 0;
}
