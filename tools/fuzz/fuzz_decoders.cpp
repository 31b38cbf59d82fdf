// Developer harness (not part of the product or the tests): the texture decoders under AddressSanitizer + UBSan on the test assets and on
// thousands of damaged copies of them (random byte flips, truncations, spliced headers).  A decoder may refuse a file (LjError) — it may
// not crash, read out of bounds, overflow or hang.      tools/fuzz_decoders.sh
#include "../../lajolla_public_amd/csrc/host/host_scene.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <unistd.h>

int main(int argc, char **argv) {
    const int rounds = atoi(argv[1]);
    std::mt19937_64 rng(20261004);
    long ok = 0, refused = 0;
    double worst = 0; std::string worst_name;
    for (int a = 2; a < argc; a++) {
        const std::string path = argv[a], ext = path.substr(path.find_last_of('.'));
        std::ifstream f(path, std::ios::binary);
        std::vector<char> orig((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (orig.empty()) continue;
        const std::string tmp = "/tmp/lj_fuzz_" + std::to_string(getpid()) + ext;
        for (int r = 0; r <= rounds; r++) {
            std::vector<char> d = orig;
            if (r > 0) {
                const int mode = (int)(rng() % 4);
                if (mode == 0) d.resize((size_t)(rng() % (d.size() + 1)));                                              // cut short
                else if (mode == 1) { for (int k = 0, n = 1 + (int)(rng() % 8); k < n && !d.empty(); k++) d[rng() % d.size()] = (char)rng(); }   // a few bytes
                else if (mode == 2) { const size_t at = rng() % d.size(), n = std::min<size_t>(d.size() - at, 1 + rng() % 16); for (size_t k = 0; k < n; k++) d[at + k] = (char)(rng() % 3 == 0 ? 0xff : rng()); }
                else { const size_t n = std::min<size_t>(d.size(), 64); for (int k = 0; k < 3; k++) d[rng() % n] = (char)rng(); }   // header only
            }
            { std::ofstream o(tmp, std::ios::binary); o.write(d.data(), (std::streamsize)d.size()); }
            for (int ch : {3, 1}) {
                const auto t0 = std::chrono::steady_clock::now();
                try { lj::HostImage img = lj::read_image(tmp, ch); ok++; if ((size_t)img.width * img.height * img.channels != img.data.size()) { printf("SIZE MISMATCH %s\n", path.c_str()); return 2; } }
                catch (const lj::LjError &) { refused++; }
                catch (const std::bad_alloc &) { refused++; }
                catch (const std::length_error &) { refused++; }
                const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (s > worst) { worst = s; worst_name = path + " round " + std::to_string(r); }
            }
        }
        unlink(tmp.c_str());
    }
    printf("decoded %ld, refused %ld, slowest decode %.3f s (%s)\n", ok, refused, worst, worst_name.c_str());
    return 0;
}
