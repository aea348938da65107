// 8-bit image I/O for the headless host: the role stb_image plays behind Image::f_load in the
// reference (source/I-O/File.cpp:13-29): decode to 8-bit, 1/3/4 channels, rows top to bottom, no flip.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace klhost {

struct Image {
    std::vector<uint8_t> data;
    int width = 0, height = 0, channels = 0;

    // Chosen by the file's first bytes, not its name, as stb_image does ("resources/Texture.png" holds JPEG data).
    // PNG: every colour type and bit depth, non-interlaced, reduced to 8 bits per sample the way stb_image does (1 to 4
    // channels). JPEG: baseline Huffman, grey or three components. false on failure.
    bool f_load(const std::string &path);
    // raw dump written by the tests: width*height*channels bytes
    bool f_load_raw(const std::string &path, int w, int h, int c);
    bool f_save_png(const std::string &path) const;
};

// Baseline JPEG bytes -> 8-bit grey or RGB with stb_image's arithmetic (kl_jpeg_impl.cpp). false on failure.
bool decode_jpeg(const uint8_t *bytes, size_t size, Image &img);

}  // namespace klhost
