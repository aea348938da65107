/* TEST INFRASTRUCTURE. Decodes an image with the stb_image.h that the reference vendors
 * (external/stb_image.h, compiled from where it lies; it is what Image::f_load calls,
 * source/I-O/File.cpp:13-29) and writes the raw 8-bit pixels, so that golden vectors see the
 * exact bytes the reference application would upload. usage: stb_decode in.png out.raw */
#define STB_IMAGE_IMPLEMENTATION
#include <stb_image.h>
#include <stdio.h>

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    int w, h, c;
    unsigned char *d = stbi_load(argv[1], &w, &h, &c, 0);
    if (!d) { fprintf(stderr, "stbi_load failed: %s\n", stbi_failure_reason()); return 1; }
    FILE *f = fopen(argv[2], "wb");
    fwrite(d, 1, (size_t)w * h * c, f);
    fclose(f);
    printf("%d %d %d\n", w, h, c);
    return 0;
}
