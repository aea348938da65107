// Baseline JPEG decoder for the headless host. The reference's albedo texture ("resources/Texture.png") holds JPEG data
// and is decoded by stb_image behind Image::f_load (source/I-O/File.cpp:13-29); texel values feed the path tracer directly
// (NEAREST lookup), so this decoder follows stb_image's arithmetic to the bit: its 12-bit integer inverse DCT (two passes,
// 2 extra bits kept between them), its 3:1 / 9:3:3:1 chroma interpolation and its 20-bit fixed-point YCbCr -> RGB. Decoders
// of the libjpeg family give other bytes for the same file (1.7 % of them, by up to 3).
//
// Handled: 8-bit Huffman sequential frames (SOF0 / SOF1), 1 or 3 components, sampling factors 1..4, interleaved and
// per-component scans, restart intervals, JFIF / Adobe colour tags. Not handled (returns false): progressive and
// arithmetic-coded frames, 12-bit samples, 4-component (CMYK / YCCK) files.
#include "kl_image.hpp"

#include <cstring>

namespace klhost {
namespace {

constexpr int kNatural[64] = {           // zig-zag position -> row-major position in the 8x8 block
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
    bool present = false;
    uint16_t fast[512];                  // 9 leading bits -> (length << 8) | symbol, 0 when the code is longer
    int32_t first_code[18], first_index[18], last_code[18];   // canonical code ranges per length 1..16
    uint8_t symbols[256];

    bool build(const uint8_t counts[16], const uint8_t *syms, int nsyms) {
        std::memset(fast, 0, sizeof(fast));
        std::memcpy(symbols, syms, (size_t)nsyms);
        int code = 0, index = 0;
        for (int len = 1; len <= 16; len++) {
            first_code[len] = code; first_index[len] = index;
            for (int i = 0; i < counts[len - 1]; i++, code++, index++) {
                if (code >= (1 << len)) return false;        // over-subscribed: more codes of this length than there can be
                if (len <= 9)
                    for (int pad = 0; pad < (1 << (9 - len)); pad++) fast[(code << (9 - len)) | pad] = (uint16_t)((len << 8) | symbols[index]);
            }
            last_code[len] = code - 1;                       // < first_code when the length has no codes
            code <<= 1;
        }
        present = true;
        return true;
    }
};

// Entropy-coded segment reader: removes 0xFF00 stuffing, stops at the first marker and feeds zero bits from there on.
struct Bits {
    const uint8_t *p = nullptr, *end = nullptr;
    uint64_t acc = 0;
    int n = 0;
    bool at_marker = false;

    void fill() {
        while (n <= 56) {
            uint64_t byte = 0;
            if (!at_marker && p < end) {
                if (*p != 0xFF) byte = *p++;
                else {
                    const uint8_t *q = p + 1;
                    while (q < end && *q == 0xFF) q++;       // fill bytes before a marker
                    if (q < end && *q == 0x00) { byte = 0xFF; p = q + 1; }
                    else { at_marker = true; p = q < end ? q - 1 : end; }
                }
            } else at_marker = true;
            acc |= byte << (56 - n);
            n += 8;
        }
    }
    uint32_t peek16() { if (n < 16) fill(); return (uint32_t)(acc >> 48); }
    void drop(int k) { acc <<= k; n -= k; }
    int take(int k) { if (n < k) fill(); int v = (int)(acc >> (64 - k)); drop(k); return v; }
    void restart(const uint8_t *from) { p = from; acc = 0; n = 0; at_marker = false; }
};

int decode_symbol(Bits &b, const Huffman &h) {
    const uint32_t bits = b.peek16();
    const uint16_t f = h.fast[bits >> 7];
    if (f) { b.drop(f >> 8); return f & 255; }
    for (int len = 10; len <= 16; len++) {
        const int code = (int)(bits >> (16 - len));
        if (code <= h.last_code[len] && code >= h.first_code[len]) { b.drop(len); return h.symbols[h.first_index[len] + code - h.first_code[len]]; }
    }
    return -1;
}

// a value of `size` bits in the sign convention of the standard (F.2.2.1): the upper half is positive
int receive_extend(Bits &b, int size) {
    const int v = b.take(size);
    return v < (1 << (size - 1)) ? v - (1 << size) + 1 : v;
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int x = 0, y = 0;                    // samples that belong to the image
    int w2 = 0, h2 = 0;                  // plane size, whole MCUs
    int dc_pred = 0;
    std::vector<uint8_t> plane;
};

constexpr int fix12(float x) { return (int)(x * 4096 + 0.5); }

// One 8-point pass of the inverse DCT (the "islow" integer factorisation with 12-bit constants): the even part comes back
// in e[0..3], the odd part in o[0..3]; output k is e[k] + o[3-k] and output 7-k is e[k] - o[3-k]. 64-bit so that corrupt
// coefficients cannot overflow; for coefficients an encoder produces every intermediate fits in 32 bits.
typedef long long wide;
inline void idct_1d(const wide s[8], wide e[4], wide o[4]) {
    wide p2 = s[2], p3 = s[6];
    wide p1 = (p2 + p3) * fix12(0.5411961f);
    wide t2 = p1 + p3 * fix12(-1.847759065f);
    wide t3 = p1 + p2 * fix12(0.765366865f);
    wide t0 = (s[0] + s[4]) * 4096, t1 = (s[0] - s[4]) * 4096;
    e[0] = t0 + t3; e[3] = t0 - t3; e[1] = t1 + t2; e[2] = t1 - t2;
    t0 = s[7]; t1 = s[5]; t2 = s[3]; t3 = s[1];
    p3 = t0 + t2;
    wide p4 = t1 + t3;
    p1 = t0 + t3; p2 = t1 + t2;
    const wide p5 = (p3 + p4) * fix12(1.175875602f);
    t0 *= fix12(0.298631336f); t1 *= fix12(2.053119869f); t2 *= fix12(3.072711026f); t3 *= fix12(1.501321110f);
    p1 = p5 + p1 * fix12(-0.899976223f); p2 = p5 + p2 * fix12(-2.562915447f);
    p3 *= fix12(-1.961570560f); p4 *= fix12(-0.390180644f);
    o[3] = t3 + p1 + p4; o[2] = t2 + p2 + p3; o[1] = t1 + p2 + p4; o[0] = t0 + p1 + p3;
}

inline uint8_t clamp8(wide x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

void idct_block(uint8_t *out, int stride, const int16_t coef[64]) {
    wide mid[64];
    for (int c = 0; c < 8; c++) {        // columns; the result keeps 2 fractional bits
        wide s[8], e[4], o[4];
        bool dc_only = true;
        for (int r = 0; r < 8; r++) { s[r] = coef[r * 8 + c]; if (r && s[r]) dc_only = false; }
        if (dc_only) { for (int r = 0; r < 8; r++) mid[r * 8 + c] = s[0] * 4; continue; }
        idct_1d(s, e, o);
        for (int k = 0; k < 4; k++) {
            mid[k * 8 + c] = (e[k] + 512 + o[3 - k]) >> 10;
            mid[(7 - k) * 8 + c] = (e[k] + 512 - o[3 - k]) >> 10;
        }
    }
    for (int r = 0; r < 8; r++) {        // rows; 17 bits to remove, +128 level shift folded into the rounding term
        wide e[4], o[4];
        idct_1d(&mid[r * 8], e, o);
        uint8_t *row = out + (size_t)r * stride;
        for (int k = 0; k < 4; k++) {
            const wide bias = 65536 + (128 << 17);
            row[k] = clamp8((e[k] + bias + o[3 - k]) >> 17);
            row[7 - k] = clamp8((e[k] + bias - o[3 - k]) >> 17);
        }
    }
}

// Chroma (or any sub-sampled plane) row -> full resolution. `near` is the source row closer to the output row.
const uint8_t *upsample_row(uint8_t *out, const uint8_t *near, const uint8_t *far, int w, int hs, int vs) {
    if (hs == 1 && vs == 1) return near;
    if (hs == 1 && vs == 2) { for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2); return out; }
    if (hs == 2 && vs == 1) {
        if (w == 1) { out[0] = out[1] = near[0]; return out; }
        out[0] = near[0];
        out[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
        for (int i = 1; i < w - 1; i++) {
            const int n = 3 * near[i] + 2;
            out[2 * i] = (uint8_t)((n + near[i - 1]) >> 2);
            out[2 * i + 1] = (uint8_t)((n + near[i + 1]) >> 2);
        }
        out[2 * (w - 1)] = (uint8_t)((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
        out[2 * (w - 1) + 1] = near[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {            // vertical 3:1 first, then horizontal 3:1 on the unrounded sums
        int cur = 3 * near[0] + far[0];
        if (w == 1) { out[0] = out[1] = (uint8_t)((cur + 2) >> 2); return out; }
        out[0] = (uint8_t)((cur + 2) >> 2);
        for (int i = 1; i < w; i++) {
            const int prev = cur;
            cur = 3 * near[i] + far[i];
            out[2 * i - 1] = (uint8_t)((3 * prev + cur + 8) >> 4);
            out[2 * i] = (uint8_t)((3 * cur + prev + 8) >> 4);
        }
        out[2 * w - 1] = (uint8_t)((cur + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; i++)          // other ratios: replicate
        for (int j = 0; j < hs; j++) out[i * hs + j] = near[i];
    return out;
}

constexpr int fix20(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }

void ycc_to_rgb_row(uint8_t *out, const uint8_t *y, const uint8_t *cb, const uint8_t *cr, int count) {
    for (int i = 0; i < count; i++, out += 3) {
        const int yf = (y[i] << 20) + (1 << 19);
        const int r_ = cr[i] - 128, b_ = cb[i] - 128;
        const int r = (yf + r_ * fix20(1.40200f)) >> 20;
        const int g = (int)(yf + r_ * -fix20(0.71414f) + (int)((unsigned)(b_ * -fix20(0.34414f)) & 0xffff0000u)) >> 20;
        const int b = (yf + b_ * fix20(1.77200f)) >> 20;
        out[0] = clamp8(r); out[1] = clamp8(g); out[2] = clamp8(b);
    }
}

struct Decoder {
    const uint8_t *file, *end;
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    int restart_interval = 0;
    bool jfif = false;
    int adobe_transform = -1;
    uint16_t quant[4][64];
    bool quant_present[4] = {false, false, false, false};
    Huffman dc[4], ac[4];
    Component comp[3];
    bool have_frame = false, have_scan = false;

    Decoder(const uint8_t *d, size_t n) : file(d), end(d + n) {}

    static int be16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

    bool parse_dqt(const uint8_t *p, int len) {
        while (len > 0) {
            const int pq = p[0] >> 4, t = p[0] & 15;
            if (pq > 1 || t > 3) return false;
            const int need = 1 + (pq ? 128 : 64);
            if (len < need) return false;
            for (int i = 0; i < 64; i++) quant[t][kNatural[i]] = (uint16_t)(pq ? be16(p + 1 + 2 * i) : p[1 + i]);
            quant_present[t] = true;
            p += need; len -= need;
        }
        return len == 0;
    }
    bool parse_dht(const uint8_t *p, int len) {
        while (len > 0) {
            if (len < 17) return false;
            const int cls = p[0] >> 4, t = p[0] & 15;
            if (cls > 1 || t > 3) return false;
            int total = 0;
            for (int i = 0; i < 16; i++) total += p[1 + i];
            if (total > 256 || len < 17 + total) return false;
            if (!(cls ? ac[t] : dc[t]).build(p + 1, p + 17, total)) return false;
            p += 17 + total; len -= 17 + total;
        }
        return len == 0;
    }
    bool parse_sof(const uint8_t *p, int len) {
        if (have_frame || len < 6 || p[0] != 8) return false;
        height = be16(p + 1); width = be16(p + 3); ncomp = p[5];
        if (width <= 0 || height <= 0 || (ncomp != 1 && ncomp != 3) || len != 6 + 3 * ncomp) return false;
        for (int i = 0; i < ncomp; i++) {
            Component &c = comp[i];
            c.id = p[6 + 3 * i]; c.h = p[7 + 3 * i] >> 4; c.v = p[7 + 3 * i] & 15; c.tq = p[8 + 3 * i];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return false;
            if (c.h > hmax) hmax = c.h;
            if (c.v > vmax) vmax = c.v;
        }
        for (int i = 0; i < ncomp; i++)
            if (hmax % comp[i].h || vmax % comp[i].v) return false;
        mcux = (width + 8 * hmax - 1) / (8 * hmax); mcuy = (height + 8 * vmax - 1) / (8 * vmax);
        // every coded block takes at least two bits (a DC code and an end-of-block code): a header that announces more
        // blocks than the file can hold is rejected before any plane is allocated
        size_t blocks = 0;
        for (int i = 0; i < ncomp; i++) blocks += (size_t)mcux * mcuy * comp[i].h * comp[i].v;
        if (blocks > 4 * (size_t)(end - file)) return false;
        for (int i = 0; i < ncomp; i++) {
            Component &c = comp[i];
            c.x = (width * c.h + hmax - 1) / hmax; c.y = (height * c.v + vmax - 1) / vmax;
            c.w2 = mcux * c.h * 8; c.h2 = mcuy * c.v * 8;
            c.plane.assign((size_t)c.w2 * c.h2, 0);
        }
        have_frame = true;
        return true;
    }

    bool decode_block(Bits &b, Component &c, int bx, int by) {
        const Huffman &hd = dc[c.td], &ha = ac[c.ta];
        const uint16_t *q = quant[c.tq];
        int16_t coef[64];
        std::memset(coef, 0, sizeof(coef));
        const int t = decode_symbol(b, hd);
        if (t < 0 || t > 15) return false;
        c.dc_pred += t ? receive_extend(b, t) : 0;
        const long long dc0 = (long long)c.dc_pred * q[0];
        if (c.dc_pred < -65536 || c.dc_pred > 65536 || dc0 < -32768 || dc0 > 32767) return false;   // no encoder produces this
        coef[0] = (int16_t)dc0;
        for (int k = 1; k < 64;) {
            const int rs = decode_symbol(b, ha);
            if (rs < 0) return false;
            const int run = rs >> 4, size = rs & 15;
            if (size == 0) {
                if (rs != 0xF0) break;   // end of block
                k += 16;
                continue;
            }
            k += run;
            if (k > 63) return false;
            const int pos = kNatural[k++];
            coef[pos] = (int16_t)(receive_extend(b, size) * q[pos]);
        }
        idct_block(&c.plane[(size_t)by * 8 * c.w2 + (size_t)bx * 8], c.w2, coef);
        return true;
    }

    // returns the position after the entropy-coded data (at a marker or the end of the file), nullptr on corrupt data
    const uint8_t *decode_scan(const uint8_t *p, int len) {
        if (!have_frame || len < 1) return nullptr;
        const int ns = p[0];
        if (ns < 1 || ns > ncomp || len != 4 + 2 * ns) return nullptr;
        int order[3];
        for (int i = 0; i < ns; i++) {
            int which = -1;
            for (int k = 0; k < ncomp; k++) if (comp[k].id == p[1 + 2 * i]) which = k;
            if (which < 0) return nullptr;
            Component &c = comp[which];
            c.td = p[2 + 2 * i] >> 4; c.ta = p[2 + 2 * i] & 15;
            if (c.td > 3 || c.ta > 3 || !dc[c.td].present || !ac[c.ta].present || !quant_present[c.tq]) return nullptr;
            order[i] = which;
        }
        if (p[1 + 2 * ns] != 0 || p[2 + 2 * ns] != 63 || p[3 + 2 * ns] != 0) return nullptr;   // spectral range of a sequential scan
        for (int k = 0; k < ncomp; k++) comp[k].dc_pred = 0;

        Bits b;
        b.end = end;
        b.restart(p + len);
        int until_restart = restart_interval ? restart_interval : 0x7fffffff;
        // between MCUs: at the end of a restart interval the coder pads to a byte and writes RSTn; without one the scan ends there
        auto interval_done = [&]() -> bool {
            if (--until_restart > 0) return true;
            b.fill();
            const uint8_t *m = b.p;
            if (!b.at_marker || m + 1 >= end || m[0] != 0xFF || m[1] < 0xD0 || m[1] > 0xD7) return false;
            b.restart(m + 2);
            for (int k = 0; k < ncomp; k++) comp[k].dc_pred = 0;
            until_restart = restart_interval;
            return true;
        };
        bool more = true;
        if (ns == 1) {                   // one component: its own block grid, not padded to whole MCUs
            Component &c = comp[order[0]];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int by = 0; by < bh && more; by++)
                for (int bx = 0; bx < bw && more; bx++) {
                    if (!decode_block(b, c, bx, by)) return nullptr;
                    more = interval_done();
                }
        } else {
            for (int my = 0; my < mcuy && more; my++)
                for (int mx = 0; mx < mcux && more; mx++) {
                    for (int i = 0; i < ns; i++) {
                        Component &c = comp[order[i]];
                        for (int v = 0; v < c.v; v++)
                            for (int h = 0; h < c.h; h++)
                                if (!decode_block(b, c, mx * c.h + h, my * c.v + v)) return nullptr;
                    }
                    more = interval_done();
                }
        }
        have_scan = true;
        // resume the marker search where the reader stopped; bytes it buffered past that point were zeros or not consumed
        b.fill();
        return b.at_marker ? b.p : end;
    }

    bool run(Image &img) {
        if (end - file < 4 || file[0] != 0xFF || file[1] != 0xD8) return false;
        const uint8_t *p = file + 2;
        while (p + 4 <= end) {
            if (p[0] != 0xFF) { p++; continue; }             // garbage between segments is skipped
            const int m = p[1];
            if (m == 0xFF) { p++; continue; }
            if (m == 0xD9) break;
            if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { p += 2; continue; }
            const int len = be16(p + 2) - 2;
            const uint8_t *body = p + 4;
            if (len < 0 || body + len > end) return false;
            bool ok = true;
            switch (m) {
                case 0xDB: ok = parse_dqt(body, len); break;
                case 0xC4: ok = parse_dht(body, len); break;
                case 0xC0: case 0xC1: ok = parse_sof(body, len); break;
                case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                    return false;                            // progressive, lossless, arithmetic: not handled
                case 0xDD: ok = len == 2; if (ok) restart_interval = be16(body); break;
                case 0xE0: if (len >= 5 && !std::memcmp(body, "JFIF", 5)) jfif = true; break;
                case 0xEE: if (len >= 12 && !std::memcmp(body, "Adobe", 6)) adobe_transform = body[11]; break;
                case 0xDA: {
                    const uint8_t *next = decode_scan(body, len);
                    if (!next) return false;
                    p = next;
                    continue;
                }
                default: break;
            }
            if (!ok) return false;
            p = body + len;
        }
        if (!have_scan) return false;
        assemble(img);
        return true;
    }

    void assemble(Image &img) {
        img.width = width; img.height = height; img.channels = ncomp == 3 ? 3 : 1;
        img.data.assign((size_t)width * height * img.channels, 0);
        const bool is_rgb = ncomp == 3 && ((comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') || (adobe_transform == 0 && !jfif));
        struct Walk { int hs, vs, ystep, ypos, w; const uint8_t *line0, *line1; std::vector<uint8_t> buf; } walk[3];
        for (int k = 0; k < ncomp; k++) {
            Walk &w = walk[k];
            w.hs = hmax / comp[k].h; w.vs = vmax / comp[k].v;
            w.ystep = w.vs >> 1; w.ypos = 0; w.w = (width + w.hs - 1) / w.hs;
            w.line0 = w.line1 = comp[k].plane.data();
            w.buf.assign((size_t)width + 8, 0);
        }
        for (int j = 0; j < height; j++) {
            const uint8_t *row[3] = {nullptr, nullptr, nullptr};
            for (int k = 0; k < ncomp; k++) {
                Walk &w = walk[k];
                // an output row in the lower half of its source row interpolates towards the next source row
                const bool lower = w.ystep >= (w.vs >> 1);
                row[k] = upsample_row(w.buf.data(), lower ? w.line1 : w.line0, lower ? w.line0 : w.line1, w.w, w.hs, w.vs);
                if (++w.ystep >= w.vs) {
                    w.ystep = 0;
                    w.line0 = w.line1;
                    if (++w.ypos < comp[k].y) w.line1 += comp[k].w2;
                }
            }
            uint8_t *out = &img.data[(size_t)j * width * img.channels];
            if (ncomp == 1) std::memcpy(out, row[0], (size_t)width);
            else if (is_rgb) for (int i = 0; i < width; i++) { out[3 * i] = row[0][i]; out[3 * i + 1] = row[1][i]; out[3 * i + 2] = row[2][i]; }
            else ycc_to_rgb_row(out, row[0], row[1], row[2], width);
        }
    }
};

}  // namespace

bool decode_jpeg(const uint8_t *bytes, size_t size, Image &img) {
    Decoder d(bytes, size);
    if (!d.run(img)) { img.data.clear(); img.width = img.height = img.channels = 0; return false; }
    return true;
}

}  // namespace klhost
