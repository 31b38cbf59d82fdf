// Developer harness (not part of the product or the tests): the scene front end (Mitsuba XML, OBJ, .serialized, .vol readers, flattening and
// the BVH builder) under AddressSanitizer + UBSan on copies of small shipped scenes whose XML or one of whose resource files is damaged
// (byte flips, cuts, digit edits).  A damaged scene may be refused (LjError) — it may not crash, read out of bounds, overflow or hang.
//     tools/fuzz_scenes.sh
#include "../../lajolla_public_amd/csrc/host/host_scene.h"
#include "../../lajolla_public_amd/csrc/host/flatten.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <memory>
#include <random>
namespace fs = std::filesystem;

static std::vector<char> slurp(const fs::path &p) { std::ifstream f(p, std::ios::binary); return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
static void spit(const fs::path &p, const std::vector<char> &d) { std::ofstream o(p, std::ios::binary); o.write(d.data(), (std::streamsize)d.size()); }

int main(int argc, char **argv) {
    // argv: rounds, work dir (a private copy of the scene directory), scene xml inside it, files to damage (the xml itself and / or resources)
    const int rounds = atoi(argv[1]);
    const fs::path work = argv[2], xml = work / argv[3];
    std::mt19937_64 rng(20261004);
    long ok = 0, refused = 0; double worst = 0;
    for (int a = 4; a < argc; a++) {
        const fs::path victim = work / argv[a];
        const std::vector<char> orig = slurp(victim);
        if (orig.empty()) { printf("cannot read %s\n", victim.c_str()); return 2; }
        const bool text = victim.extension() == ".xml" || victim.extension() == ".obj";
        for (int r = 0; r <= rounds; r++) {
            std::vector<char> d = orig;
            if (r > 0) {
                const int mode = (int)(rng() % (text ? 5 : 3));
                if (mode == 0) d.resize((size_t)(rng() % (d.size() + 1)));
                else if (mode == 1) { for (int k = 0, n = 1 + (int)(rng() % 6); k < n && !d.empty(); k++) d[rng() % d.size()] = (char)rng(); }
                else if (mode == 2) { const size_t n = std::min<size_t>(d.size(), 96); for (int k = 0; k < 3; k++) d[rng() % n] = (char)rng(); }
                else if (mode == 3) { for (int k = 0; k < 4; k++) { size_t at = rng() % d.size(); for (size_t q = 0; q < d.size(); q++, at = (at + 1) % d.size()) if (d[at] >= '0' && d[at] <= '9') { d[at] = (char)('0' + rng() % 10); break; } } }   // edit digits
                else { const size_t at = rng() % d.size(); const char *junk[] = {"-", "1e308", "nan", "\"", "<", "/>", "999999999999", " ", "0"}; const std::string j = junk[rng() % 9]; d.insert(d.begin() + at, j.begin(), j.end()); }
            }
            spit(victim, d);
            const auto t0 = std::chrono::steady_clock::now();
            try {
                std::unique_ptr<lj::HostScene> hs(lj::parse_scene_xml(xml.string()));
                hs->finalize();
                lj::FlatScene F = lj::flatten_scene(hs->desc);
                ok++;
            } catch (const lj::LjError &) { refused++; }
            catch (const std::bad_alloc &) { refused++; }
            catch (const std::length_error &) { refused++; }
            worst = std::max(worst, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        spit(victim, orig);
    }
    printf("%s: loaded %ld, refused %ld, slowest %.2f s\n", argv[3], ok, refused, worst);
    return 0;
}
