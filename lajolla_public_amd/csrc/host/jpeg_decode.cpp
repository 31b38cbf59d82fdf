// JPEG (ITU-T T.81: sequential and progressive DCT, Huffman, 8-bit; interleaved and per-component scans) decoder for texture input.
//
// The reference loads LDR textures through stb_image (image.cpp:44,96), so texel values depend on three choices the
// JPEG standard leaves to the decoder.  We make the same ones so that the decoded texels are bit-identical to what
// the reference's TexturePool holds (checked against tests/golden/scene_sponza.json):
//   * the 8x8 inverse DCT is the 13-bit "slow integer" (Loeffler-Ligtenberg-Moschytz) transform with 12-bit constants,
//     two passes, 2 extra bits kept between them, +128 level shift folded into the final rounding;
//   * 2x2 chroma upsampling uses the triangle filter  (3*near + far) per axis: (3*t_i + t_{i±1} + 8) >> 4;
//   * YCbCr -> RGB in 20-bit fixed point with the BT.601 constants quantised to 12 bits.
// Progressive files collect their coefficients over all scans (DC / AC bands, successive approximation) and go through the same transform.
// Four-component files follow the Adobe marker: CMYK (transform 0), YCCK (2), else the fourth component is ignored.
// Arithmetic-coded / lossless / hierarchical / 12-bit files are rejected with LJ_ERR_UNSUPPORTED (the reference's loader refuses them too).
#include "host_scene.h"
#include <cstring>
#include <fstream>

namespace lj {

namespace {

struct Huff {
    // canonical Huffman table (T.81 Annex C): codes of length l are consecutive, starting at first_code[l]
    uint8_t nsym[17]; uint8_t sym[256];
    int first_code[18], first_index[18];
    void build() {
        int code = 0, idx = 0;
        for (int l = 1; l <= 16; l++) { first_code[l] = code; first_index[l] = idx; code = (code + nsym[l]) << 1; idx += nsym[l]; }
    }
};

struct Component { int id, h, v, tq, td, ta; int x, y; int w2, h2; std::vector<uint8_t> data; std::vector<short> coeff; int dc_pred; };   // x, y: its size in samples; w2, h2: padded to whole MCUs

struct BitReader {
    const uint8_t *p, *end; uint32_t buf = 0; int nbits = 0; bool hit_marker = false;
    void fill() {
        while (nbits <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    int c = p < end ? *p : 0;
                    if (c == 0) p++;            // stuffed zero
                    else { hit_marker = true; p--; b = 0; }  // a marker: feed zeros from here on
                }
            }
            buf |= (uint32_t)b << (24 - nbits); nbits += 8;
        }
    }
    int bit() { if (nbits < 1) fill(); int r = buf >> 31; buf <<= 1; nbits--; return r; }
    int bits(int n) { if (n == 0) return 0; if (nbits < n) fill(); int r = buf >> (32 - n); buf <<= n; nbits -= n; return r; }
    void reset() { buf = 0; nbits = 0; hit_marker = false; }
};

int decode_symbol(BitReader &br, const Huff &h) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br.bit();
        if (h.nsym[l] && code - h.first_code[l] < h.nsym[l] && code >= h.first_code[l]) return h.sym[h.first_index[l] + code - h.first_code[l]];
    }
    throw LjError(LJ_ERR_PARSE, "corrupt JPEG: bad Huffman code");
}
inline int extend(int v, int n) { return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }  // T.81 F.2.2.1

inline uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
inline int f2f(double x) { return (int)(x * 4096 + 0.5); }

// 32-bit two's-complement arithmetic that wraps (a damaged file can hold coefficients no encoder would write; their transform must stay
// defined behaviour — for the values of a valid file nothing ever wraps, so results are unchanged)
inline int wadd(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }
inline int wmul(int a, int b) { return (int)((uint32_t)a * (uint32_t)b); }
struct W {
    int v;
    W() : v(0) {}
    W(int x) : v(x) {}
    W &operator+=(W b) { v = wadd(v, b.v); return *this; }
};
inline W operator+(W a, W b) { return W(wadd(a.v, b.v)); }
inline W operator-(W a, W b) { return W((int)((uint32_t)a.v - (uint32_t)b.v)); }
inline W operator*(W a, W b) { return W(wmul(a.v, b.v)); }
inline int operator>>(W a, int n) { return a.v >> n; }

// one 1-D pass of the slow-integer IDCT on s0..s7; results are x0..x3 (even part) and t0..t3 (odd part), scaled by 2^12
#define LJ_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                                 \
    W t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                         \
    p2 = W(s2); p3 = W(s6);                                                                        \
    p1 = (p2 + p3) * f2f(0.5411961f); t2 = p1 + p3 * f2f(-1.847759065f); t3 = p1 + p2 * f2f(0.765366865f); \
    p2 = W(s0); p3 = W(s4);                                                                        \
    t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;                                                  \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                                        \
    t0 = W(s7); t1 = W(s5); t2 = W(s3); t3 = W(s1);                                                \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;                                        \
    p5 = (p3 + p4) * f2f(1.175875602f);                                                            \
    t0 = t0 * f2f(0.298631336f); t1 = t1 * f2f(2.053119869f); t2 = t2 * f2f(3.072711026f); t3 = t3 * f2f(1.501321110f); \
    p1 = p5 + p1 * f2f(-0.899976223f); p2 = p5 + p2 * f2f(-2.562915447f);                          \
    p3 = p3 * f2f(-1.961570560f); p4 = p4 * f2f(-0.390180644f);                                    \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;

void idct_block(uint8_t *out, int stride, const short d[64]) {
    int val[64];
    for (int i = 0; i < 8; i++) {  // columns
        const short *c = d + i; int *v = val + i;
        LJ_IDCT_1D(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56])
        x0 += 512; x1 += 512; x2 += 512; x3 += 512;  // keep 2 extra bits
        v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10; v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
        v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10; v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
    }
    for (int i = 0; i < 8; i++) {  // rows; 2^12 * 2^2 * 2^3 = 2^17 to remove, round, and level-shift by +128
        const int *v = val + 8 * i; uint8_t *o = out + (size_t)stride * i;
        LJ_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        const W bias(65536 + (128 << 17));
        x0 += bias; x1 += bias; x2 += bias; x3 += bias;
        o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17); o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
        o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17); o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
    }
}

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// upsample one output row of a component to full width
void resample_row(uint8_t *out, const uint8_t *near_, const uint8_t *far_, int w_lores, int hs, int vs) {
    if (hs == 1 && vs == 1) { memcpy(out, near_, w_lores); return; }
    if (hs == 1 && vs == 2) { for (int i = 0; i < w_lores; i++) out[i] = (uint8_t)((3 * near_[i] + far_[i] + 2) >> 2); return; }
    if (hs == 2 && vs == 1) {
        const uint8_t *in = near_;
        if (w_lores == 1) { out[0] = out[1] = in[0]; return; }
        out[0] = in[0]; out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w_lores - 1; i++) { int n = 3 * in[i] + 2; out[i * 2] = (uint8_t)((n + in[i - 1]) >> 2); out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2); }
        out[i * 2] = (uint8_t)((in[w_lores - 2] * 3 + in[w_lores - 1] + 2) >> 2); out[i * 2 + 1] = in[w_lores - 1];
        return;
    }
    if (hs == 2 && vs == 2) {
        if (w_lores == 1) { out[0] = out[1] = (uint8_t)((3 * near_[0] + far_[0] + 2) >> 2); return; }
        int t1 = 3 * near_[0] + far_[0], t0;
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w_lores; i++) {
            t0 = t1; t1 = 3 * near_[i] + far_[i];
            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4); out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[w_lores * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
        return;
    }
    for (int i = 0; i < w_lores; i++) for (int j = 0; j < hs; j++) out[i * hs + j] = near_[i];  // other ratios: replicate
}

inline int fixed12(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }

} // namespace

// returns 8-bit interleaved RGB (or grey replicated), width*height*3.  `grey` (optional): the one-channel image the reference's loader
// returns when one channel is asked for — the Y plane of a YCbCr file (the chroma planes are not even upsampled), the luma
// (77 R + 150 G + 29 B) >> 8 of an RGB-coded file, the samples of a grey file.
std::vector<uint8_t> decode_jpeg_rgb8(const std::vector<uint8_t> &file, int &width, int &height, const std::string &name, std::vector<uint8_t> *grey) {
    auto bad = [&](const char *why) -> LjError { return LjError(LJ_ERR_PARSE, std::string("JPEG ") + name + ": " + why); };
    size_t pos = 0, n = file.size();
    if (n < 4 || file[0] != 0xFF || file[1] != 0xD8) throw bad("not a JPEG (no SOI)");
    pos = 2;
    uint16_t qt[4][64]; bool have_qt[4] = {false, false, false, false};
    Huff hdc[4], hac[4]; bool have_dc[4] = {false}, have_ac[4] = {false};
    std::vector<Component> comp;
    int restart_interval = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, scans = 0;
    bool jfif = false, progressive = false; int app14_transform = -1;
    width = height = 0;
    auto rd16 = [&](size_t p) { return (file[p] << 8) | file[p + 1]; };

    // ---- one scan's entropy-coded data, from `at`; returns where the reader stopped (at the marker that ends the scan)
    // Sequential files: every block is complete after its scan and is transformed at once.  Progressive files (T.81 Annex G): a scan carries
    // the DC coefficients (all components interleaved, or one) or a band Ss..Se of one component's AC coefficients, first at precision Al,
    // then one bit at a time (Ah = Al + 1); coefficients collect in `coeff` and are dequantised and transformed after the last scan.
    auto decode_scan = [&](size_t at, const std::vector<int> &order, int Ss, int Se, int Ah, int Al) -> size_t {
        BitReader br; br.p = &file[at]; br.end = file.data() + n;
        int todo = restart_interval ? restart_interval : 0x7fffffff, eob_run = 0;
        for (auto &c : comp) c.dc_pred = 0;
        short block[64];
        auto refine = [&](short &v, short bit) { if (v != 0 && br.bit() && (v & bit) == 0) v = (short)(v > 0 ? v + bit : v - bit); };
        auto one_block = [&](Component &c, int bx, int by) {   // block (bx, by) of component c, in blocks
            if (!progressive) {
                memset(block, 0, sizeof block);
                int t = decode_symbol(br, hdc[c.td]);
                if (t > 11) throw bad("bad DC magnitude");
                int diff = t ? extend(br.bits(t), t) : 0;
                c.dc_pred = wadd(c.dc_pred, diff);
                block[0] = (short)wmul(c.dc_pred, qt[c.tq][0]);
                for (int k = 1; k < 64;) {
                    int rs = decode_symbol(br, hac[c.ta]);
                    int r = rs >> 4, sz = rs & 15;
                    if (sz == 0) { if (r != 15) break; k += 16; continue; }
                    k += r;
                    if (k > 63) throw bad("AC index out of range");
                    block[kZigzag[k]] = (short)wmul(extend(br.bits(sz), sz), qt[c.tq][kZigzag[k]]);
                    k++;
                }
                idct_block(&c.data[(size_t)by * 8 * c.w2 + (size_t)bx * 8], c.w2, block);
                return;
            }
            short *d = &c.coeff[((size_t)by * (c.w2 / 8) + bx) * 64];
            if (Ss == 0) {   // DC
                if (Se != 0) throw bad("progressive scan mixes DC and AC coefficients");
                if (Ah == 0) {
                    int t = decode_symbol(br, hdc[c.td]);
                    if (t > 15) throw bad("bad DC magnitude");
                    c.dc_pred = wadd(c.dc_pred, t ? extend(br.bits(t), t) : 0);
                    d[0] = (short)wmul(c.dc_pred, 1 << Al);
                } else if (br.bit()) d[0] = (short)(d[0] + (1 << Al));
                return;
            }
            if (Ah == 0) {   // AC band, first pass
                if (eob_run) { eob_run--; return; }
                int k = Ss;
                do {
                    int rs = decode_symbol(br, hac[c.ta]);
                    int r = rs >> 4, sz = rs & 15;
                    if (sz == 0) {
                        if (r < 15) { eob_run = (1 << r); if (r) eob_run += br.bits(r); eob_run--; break; }
                        k += 16;
                    } else {
                        k += r;
                        if (k > 63) throw bad("AC index out of range");
                        d[kZigzag[k++]] = (short)wmul(extend(br.bits(sz), sz), 1 << Al);
                    }
                } while (k <= Se);
                return;
            }
            const short bit = (short)(1 << Al);   // AC band, one more bit
            if (eob_run) { eob_run--; for (int k = Ss; k <= Se; k++) refine(d[kZigzag[k]], bit); return; }
            int k = Ss;
            do {
                int rs = decode_symbol(br, hac[c.ta]);
                int r = rs >> 4, sz = rs & 15;
                if (sz == 0) {
                    if (r < 15) { eob_run = (1 << r) - 1; if (r) eob_run += br.bits(r); r = 64; }   // end of band: the rest is refinement only
                } else {
                    if (sz != 1) throw bad("bad refinement code");
                    sz = br.bit() ? bit : -bit;
                }
                while (k <= Se) {   // pass r still-zero coefficients (refining the non-zero ones on the way), then place the new one
                    short &v = d[kZigzag[k++]];
                    if (v != 0) refine(v, bit);
                    else { if (r == 0) { v = (short)sz; break; } r--; }
                }
            } while (k <= Se);
        };
        // after `restart_interval` MCUs: realign behind the restart marker and reset the predictors; false when what follows is not a restart
        // marker (the scan is over — or cut short: what was decoded stands, as in the reference's loader)
        auto restart = [&]() -> bool {
            br.reset();
            const uint8_t *q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF)) q++;
            if (!(q + 1 < br.end && q[1] >= 0xD0 && q[1] <= 0xD7)) { br.p = q; return false; }
            br.p = q + 2;
            for (auto &c : comp) c.dc_pred = 0;
            eob_run = 0;
            todo = restart_interval;
            return true;
        };
        if (order.size() == 1) {   // one component: its blocks in raster order, as many as its own size needs
            Component &c = comp[order[0]];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int by = 0; by < bh; by++) for (int bx = 0; bx < bw; bx++) { one_block(c, bx, by); if (--todo <= 0 && !restart()) return (size_t)(br.p - file.data()); }
        } else {
            if (progressive && Ss != 0) throw bad("interleaved progressive AC scan");
            for (int my = 0; my < mcuy; my++) for (int mx = 0; mx < mcux; mx++) {
                for (int k : order) { Component &c = comp[k]; for (int y = 0; y < c.v; y++) for (int x = 0; x < c.h; x++) one_block(c, mx * c.h + x, my * c.v + y); }
                if (--todo <= 0 && !restart()) return (size_t)(br.p - file.data());
            }
        }
        return (size_t)(br.p - file.data());
    };

    bool done = false;
    while (!done) {
        while (pos < n && file[pos] != 0xFF) pos++;
        while (pos < n && file[pos] == 0xFF) pos++;
        if (pos >= n) { if (scans) break; throw bad("unexpected end of file"); }
        int m = file[pos++];
        if (m == 0xD9) { if (scans) break; throw bad("EOI before any scan"); }
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (pos + 2 > n) { if (scans) break; throw bad("truncated segment"); }
        int len = rd16(pos);
        if (len < 2 || pos + len > n) { if (scans) break; throw bad("bad segment length"); }
        size_t seg = pos + 2, seg_end = pos + len;
        if (m == 0xDB) {  // DQT
            while (seg < seg_end) {
                int pq = file[seg] >> 4, tq = file[seg] & 15; seg++;
                if (tq > 3) throw bad("bad DQT id");
                if (seg + (pq ? 128u : 64u) > seg_end) throw bad("DQT table runs past its segment");
                for (int i = 0; i < 64; i++) { qt[tq][kZigzag[i]] = pq ? (uint16_t)rd16(seg) : file[seg]; seg += pq ? 2 : 1; }
                have_qt[tq] = true;
            }
        } else if (m == 0xC4) {  // DHT
            while (seg < seg_end) {
                int tc = file[seg] >> 4, th = file[seg] & 15; seg++;
                if (tc > 1 || th > 3) throw bad("bad DHT id");
                if (seg + 16 > seg_end) throw bad("DHT table runs past its segment");
                Huff &h = tc ? hac[th] : hdc[th];
                int total = 0; h.nsym[0] = 0;
                for (int l = 1; l <= 16; l++) { h.nsym[l] = file[seg++]; total += h.nsym[l]; }
                if (total > 256 || seg + total > seg_end) throw bad("bad DHT");
                memcpy(h.sym, &file[seg], total); seg += total;
                h.build(); (tc ? have_ac : have_dc)[th] = true;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {  // SOF0 / SOF1: sequential Huffman; SOF2: progressive Huffman
            if (!comp.empty()) throw bad("second frame header");
            progressive = m == 0xC2;
            if (seg + 6 > seg_end) throw bad("truncated frame header");
            if (file[seg] != 8) throw LjError(LJ_ERR_UNSUPPORTED, "JPEG " + name + ": only 8-bit samples are supported");
            height = rd16(seg + 1); width = rd16(seg + 3);
            int nc = file[seg + 5];
            if (width <= 0 || height <= 0 || (nc != 1 && nc != 3 && nc != 4)) throw LjError(LJ_ERR_UNSUPPORTED, "JPEG " + name + ": unsupported component count");
            if (seg + 6 + 3 * (size_t)nc > seg_end) throw bad("truncated frame header");
            check_image_size(width, height, n, name);
            comp.resize(nc);
            for (int i = 0; i < nc; i++) {
                comp[i].id = file[seg + 6 + 3 * i]; comp[i].h = file[seg + 7 + 3 * i] >> 4; comp[i].v = file[seg + 7 + 3 * i] & 15; comp[i].tq = file[seg + 8 + 3 * i];
                if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) throw bad("bad SOF component");
                hmax = std::max(hmax, comp[i].h); vmax = std::max(vmax, comp[i].v);
            }
            mcux = (width + 8 * hmax - 1) / (8 * hmax); mcuy = (height + 8 * vmax - 1) / (8 * vmax);
            for (auto &c : comp) {
                c.x = (width * c.h + hmax - 1) / hmax; c.y = (height * c.v + vmax - 1) / vmax;
                c.w2 = mcux * c.h * 8; c.h2 = mcuy * c.v * 8; c.data.assign((size_t)c.w2 * c.h2, 0); c.dc_pred = 0;
                if (progressive) c.coeff.assign((size_t)c.w2 * c.h2, 0);
            }
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            throw LjError(LJ_ERR_UNSUPPORTED, "JPEG " + name + ": lossless / hierarchical / arithmetic-coded files are not supported");
        } else if (m == 0xDD) { if (len < 4) throw bad("truncated DRI"); restart_interval = rd16(seg); }
        else if (m == 0xE0) { if (len >= 7 && !memcmp(&file[seg], "JFIF\0", 5)) jfif = true; }
        else if (m == 0xEE) { if (len >= 14 && !memcmp(&file[seg], "Adobe\0", 6)) app14_transform = file[seg + 11]; }
        else if (m == 0xDA) {  // SOS
            if (comp.empty()) throw bad("SOS before SOF");
            if (seg >= seg_end) throw bad("bad scan header");
            int ns = file[seg];
            if (ns < 1 || ns > (int)comp.size() || seg + 1 + 2 * ns + 3 > seg_end) throw bad("bad scan header");
            std::vector<int> order;
            for (int i = 0; i < ns; i++) {
                int id = file[seg + 1 + 2 * i], tt = file[seg + 2 + 2 * i], k = -1;
                for (int c = 0; c < (int)comp.size(); c++) if (comp[c].id == id) k = c;
                if (k < 0) throw bad("SOS references an unknown component");
                comp[k].td = tt >> 4; comp[k].ta = tt & 15;
                if (comp[k].td > 3 || comp[k].ta > 3) throw bad("scan uses an undefined table");
                order.push_back(k);
            }
            const int Ss = file[seg + 1 + 2 * ns], Se = file[seg + 2 + 2 * ns], Ah = file[seg + 3 + 2 * ns] >> 4, Al = file[seg + 3 + 2 * ns] & 15;
            if (progressive) { if (Ss > 63 || Se > 63 || Ss > Se || Ah > 13 || Al > 13) throw bad("bad progressive scan parameters"); }
            else if (Ss != 0 || Ah != 0 || Al != 0) throw bad("bad sequential scan parameters");
            for (int k : order) {   // the tables this scan decodes with
                const bool need_dc = !progressive || Ss == 0, need_ac = !progressive || Ss > 0;
                if ((need_dc && !(progressive && Ah) && !have_dc[comp[k].td]) || (need_ac && !have_ac[comp[k].ta]) || !have_qt[comp[k].tq]) throw bad("scan uses an undefined table");
            }
            pos = decode_scan(seg_end, order, Ss, Se, Ah, Al);
            scans++;
            continue;
        }
        pos = seg_end;
    }
    if (comp.empty() || !scans) throw bad("no image data");
    if (progressive)   // every coefficient is in: dequantise and transform
        for (auto &c : comp)
            for (int by = 0; by < c.h2 / 8; by++) for (int bx = 0; bx < c.w2 / 8; bx++) {
                short *d = &c.coeff[((size_t)by * (c.w2 / 8) + bx) * 64];
                for (int i = 0; i < 64; i++) d[i] = (short)wmul(d[i], qt[c.tq][i]);
                idct_block(&c.data[(size_t)by * 8 * c.w2 + (size_t)bx * 8], c.w2, d);
            }
    // ---- upsample + colour conversion, one output row at a time
    std::vector<uint8_t> out((size_t)width * height * 3);
    if (grey) grey->resize((size_t)width * height);
    const int nc = (int)comp.size();
    const bool is_rgb = nc == 3 && ((comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') || (app14_transform == 0 && !jfif));
    struct Res { int hs, vs, ystep, w_lores, ypos, rows; const uint8_t *line0, *line1; std::vector<uint8_t> buf; } res[4];
    for (int k = 0; k < nc; k++) {
        res[k].hs = hmax / comp[k].h; res[k].vs = vmax / comp[k].v; res[k].ystep = res[k].vs >> 1;
        res[k].w_lores = (width + res[k].hs - 1) / res[k].hs; res[k].ypos = 0; res[k].rows = (height * comp[k].v + vmax - 1) / vmax;
        res[k].line0 = res[k].line1 = comp[k].data.data(); res[k].buf.resize((size_t)width + 8);
    }
    auto ycc = [](int y, int cb_, int cr_, uint8_t *o) {   // BT.601 in 20-bit fixed point, the constants quantised to 12 bits
        int y_fixed = (y << 20) + (1 << 19);
        int cr = cr_ - 128, cb = cb_ - 128;
        int r = y_fixed + cr * fixed12(1.40200f);
        int g = y_fixed + (cr * -fixed12(0.71414f)) + ((cb * -fixed12(0.34414f)) & 0xffff0000);
        int b = y_fixed + cb * fixed12(1.77200f);
        o[0] = clamp8(r >> 20); o[1] = clamp8(g >> 20); o[2] = clamp8(b >> 20);
    };
    auto mul8 = [](int x, int y) { const unsigned t = (unsigned)(x * y + 128); return (uint8_t)((t + (t >> 8)) >> 8); };   // x * y / 255, rounded
    auto luma = [](int r, int g, int b) { return (uint8_t)((r * 77 + g * 150 + 29 * b) >> 8); };
    for (int j = 0; j < height; j++) {
        const uint8_t *row[4];
        for (int k = 0; k < nc; k++) {
            Res &r = res[k];
            bool y_bot = r.ystep >= (r.vs >> 1);
            resample_row(r.buf.data(), y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
            row[k] = r.buf.data();
            if (++r.ystep >= r.vs) { r.ystep = 0; r.line0 = r.line1; if (++r.ypos < r.rows) r.line1 += comp[k].w2; }
        }
        uint8_t *o = &out[(size_t)j * width * 3];
        if (grey) {
            uint8_t *gq = grey->data() + (size_t)j * width;
            if (nc == 3 && is_rgb) for (int i = 0; i < width; i++) gq[i] = luma(row[0][i], row[1][i], row[2][i]);
            else if (nc == 4 && app14_transform == 0) for (int i = 0; i < width; i++) gq[i] = luma(mul8(row[0][i], row[3][i]), mul8(row[1][i], row[3][i]), mul8(row[2][i], row[3][i]));
            else if (nc == 4 && app14_transform == 2) for (int i = 0; i < width; i++) gq[i] = mul8(255 - row[0][i], row[3][i]);
            else memcpy(gq, row[0], (size_t)width);
        }
        if (nc == 1) { for (int i = 0; i < width; i++) { o[3 * i] = o[3 * i + 1] = o[3 * i + 2] = row[0][i]; } }
        else if (is_rgb) { for (int i = 0; i < width; i++) { o[3 * i] = row[0][i]; o[3 * i + 1] = row[1][i]; o[3 * i + 2] = row[2][i]; } }
        else if (nc == 4 && app14_transform == 0) {   // Adobe CMYK (stored inverted): each ink times black
            for (int i = 0; i < width; i++) for (int c = 0; c < 3; c++) o[3 * i + c] = mul8(row[c][i], row[3][i]);
        } else {
            for (int i = 0; i < width; i++) ycc(row[0][i], row[1][i], row[2][i], o + 3 * i);
            if (nc == 4 && app14_transform == 2)   // Adobe YCCK
                for (int i = 0; i < width; i++) for (int c = 0; c < 3; c++) o[3 * i + c] = mul8(255 - o[3 * i + c], row[3][i]);
        }
    }
    return out;
}

} // namespace lj
