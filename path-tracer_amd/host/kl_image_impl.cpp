// Minimal PNG codec on top of zlib (inflate/deflate + the five scanline filters); JPEG data goes to kl_jpeg_impl.cpp.
#include "kl_image.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace klhost {
namespace {

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put32(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

bool read_file(const std::string &path, std::vector<uint8_t> &out) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    if (n < 0) { std::fclose(f); return false; }
    out.resize((size_t)n);
    bool ok = std::fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    std::fclose(f);
    return ok;
}
int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
void chunk(std::vector<uint8_t> &png, const char *type, const std::vector<uint8_t> &body) {
    put32(png, (uint32_t)body.size());
    size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), body.begin(), body.end());
    put32(png, (uint32_t)crc32(0L, png.data() + start, (uInt)(png.size() - start)));
}

}  // namespace

bool Image::f_load(const std::string &path) {
    std::vector<uint8_t> file;
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (!read_file(path, file)) return false;
    if (file.size() >= 3 && file[0] == 0xFF && file[1] == 0xD8 && file[2] == 0xFF) return decode_jpeg(file.data(), file.size(), *this);
    if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) return false;
    std::vector<uint8_t> idat, palette, trns;
    int bit_depth = 0, color = -1, interlace = 0;
    size_t pos = 8;
    while (pos + 12 <= file.size()) {
        uint32_t len = be32(&file[pos]);
        const char *type = (const char *)&file[pos + 4];
        if (len > file.size() || pos + 12 + len > file.size()) return false;
        const uint8_t *body = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            width = (int)be32(body); height = (int)be32(body + 4);
            bit_depth = body[8]; color = body[9]; interlace = body[12];
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + len;
    }
    // samples per pixel in the file, by colour type (0 grey, 2 RGB, 3 palette index, 4 grey + alpha, 6 RGBA)
    const int file_ch = color == 0 ? 1 : (color == 2 ? 3 : (color == 3 ? 1 : (color == 4 ? 2 : (color == 6 ? 4 : 0))));
    const bool depth_ok = color == 3 ? (bit_depth == 1 || bit_depth == 2 || bit_depth == 4 || bit_depth == 8)
                        : color == 0 ? (bit_depth == 1 || bit_depth == 2 || bit_depth == 4 || bit_depth == 8 || bit_depth == 16)
                                     : (bit_depth == 8 || bit_depth == 16);
    if (!file_ch || !depth_ok || interlace != 0 || width <= 0 || height <= 0 || width > (1 << 24) || height > (1 << 24)) return false;
    if (color == 3 && (palette.empty() || palette.size() % 3 != 0 || palette.size() > 768)) return false;
    const size_t row_bytes = ((size_t)width * file_ch * bit_depth + 7) / 8;
    const size_t bpp = (size_t)(file_ch * bit_depth + 7) / 8;          // the filters' "previous pixel" distance, at least one byte
    // deflate expands by at most 1032:1: a header that announces more pixels than the data can hold is refused before allocating
    if ((row_bytes + 1) * (size_t)height > idat.size() * 1040 + 1024) return false;
    std::vector<uint8_t> raw((row_bytes + 1) * (size_t)height);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return false;
    std::vector<uint8_t> lines(row_bytes * (size_t)height);
    for (int y = 0; y < height; y++) {
        const uint8_t *in = &raw[(row_bytes + 1) * (size_t)y];
        uint8_t *out = &lines[row_bytes * (size_t)y];
        const uint8_t *up = y ? out - row_bytes : nullptr;
        const int filter = in[0];
        for (size_t i = 0; i < row_bytes; i++) {
            int a = i >= bpp ? out[i - bpp] : 0;
            int b = up ? up[i] : 0;
            int c = (up && i >= bpp) ? up[i - bpp] : 0;
            int x = in[1 + i];
            switch (filter) {
                case 0: break;
                case 1: x += a; break;
                case 2: x += b; break;
                case 3: x += (a + b) / 2; break;
                case 4: x += paeth(a, b, c); break;
                default: return false;
            }
            out[i] = (uint8_t)x;
        }
    }
    // To 8 bits per sample, as stb_image hands them over: 16-bit samples keep their high byte, 1/2/4-bit grey is scaled to the
    // full range (x255, x85, x17), palette indices become RGB (RGBA when the file has a tRNS chunk), a tRNS colour key on
    // 8-bit grey / RGB adds an alpha channel that is 0 for the key and 255 elsewhere.
    const size_t npx = (size_t)width * height;
    std::vector<uint8_t> samples(npx * file_ch);
    const int scale = (color == 0 && bit_depth < 8) ? 255 / ((1 << bit_depth) - 1) : 1;
    for (int y = 0; y < height; y++) {
        const uint8_t *row = &lines[row_bytes * (size_t)y];
        uint8_t *dst = &samples[(size_t)y * width * file_ch];
        const size_t n = (size_t)width * file_ch;
        if (bit_depth == 8) std::memcpy(dst, row, n);
        else if (bit_depth == 16) for (size_t i = 0; i < n; i++) dst[i] = row[2 * i];
        else for (size_t i = 0; i < n; i++) {
            const int per = 8 / bit_depth, shift = (per - 1 - (int)(i % per)) * bit_depth;
            dst[i] = (uint8_t)(((row[i / per] >> shift) & ((1 << bit_depth) - 1)) * scale);
        }
    }
    if (color == 3) {
        const bool alpha = !trns.empty();
        channels = alpha ? 4 : 3;
        data.assign(npx * channels, 0);
        for (size_t i = 0; i < npx; i++) {
            const size_t k = samples[i];
            if (3 * k + 2 >= palette.size()) return false;                   // index past the palette
            uint8_t *o = &data[i * channels];
            o[0] = palette[3 * k]; o[1] = palette[3 * k + 1]; o[2] = palette[3 * k + 2];
            if (alpha) o[3] = k < trns.size() ? trns[k] : 255;
        }
        return true;
    }
    const bool keyed = !trns.empty() && (color == 0 || color == 2);
    if (keyed && (bit_depth == 16 || trns.size() != (size_t)file_ch * 2)) return false;   // 16-bit colour keys are not handled
    if (!keyed) { channels = file_ch; data.swap(samples); return true; }
    uint8_t key[3] = {0, 0, 0};
    for (int k = 0; k < file_ch; k++) key[k] = (uint8_t)(trns[2 * k + 1] * scale);
    channels = file_ch + 1;
    data.assign(npx * channels, 0);
    for (size_t i = 0; i < npx; i++) {
        bool is_key = true;
        for (int k = 0; k < file_ch; k++) { data[i * channels + k] = samples[i * file_ch + k]; is_key = is_key && samples[i * file_ch + k] == key[k]; }
        data[i * channels + file_ch] = is_key ? 0 : 255;
    }
    return true;
}

bool Image::f_load_raw(const std::string &path, int w, int h, int c) {
    std::vector<uint8_t> file;
    if (!read_file(path, file) || file.size() != (size_t)w * h * c) return false;
    data.swap(file); width = w; height = h; channels = c;
    return true;
}

bool Image::f_save_png(const std::string &path) const {
    if (data.empty() || channels < 1 || channels > 4) return false;
    const size_t stride = (size_t)width * channels;
    std::vector<uint8_t> raw((stride + 1) * (size_t)height);
    for (int y = 0; y < height; y++) {
        raw[(stride + 1) * (size_t)y] = 0;
        std::memcpy(&raw[(stride + 1) * (size_t)y + 1], &data[stride * (size_t)y], stride);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    comp.resize(clen);
    std::vector<uint8_t> png = {137, 80, 78, 71, 13, 10, 26, 10};
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)width); put32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(channels == 1 ? 0 : (channels == 2 ? 4 : (channels == 3 ? 2 : 6)));
    ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(png, "IHDR", ihdr); chunk(png, "IDAT", comp); chunk(png, "IEND", {});
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    std::fclose(f);
    return ok;
}

}  // namespace klhost
