// Sanitizer driver for the host's image loaders (CPU only): decodes every file named on the command line with
// Image::f_load (PNG and JPEG, well-formed or not: a refused file is fine, a sanitizer report is not) and writes the good
// ones back out as PNG. Built by `make asan` with -fsanitize=address,undefined; tools/asan/run.py feeds it damaged files.
#include <cstdio>
#include <string>

#include "kl_image.hpp"

int main(int argc, char **argv) {
    size_t loaded = 0, refused = 0, bytes = 0;
    for (int i = 1; i < argc; i++) {
        klhost::Image img;
        if (!img.f_load(argv[i])) { refused++; continue; }
        if (img.data.size() != (size_t)img.width * img.height * img.channels) { std::fprintf(stderr, "%s: size mismatch\n", argv[i]); return 3; }
        loaded++; bytes += img.data.size();
        klhost::Image back;
        const std::string copy = std::string(argv[i]) + ".copy.png";
        if (!img.f_save_png(copy) || !back.f_load(copy) || back.data != img.data) { std::fprintf(stderr, "%s: PNG round trip differs\n", argv[i]); return 4; }
    }
    std::printf("{\"images_loaded\": %zu, \"images_refused\": %zu, \"bytes\": %zu}\n", loaded, refused, bytes);
    return 0;
}
