/* Sanitizer driver for the CPU oracle (test infrastructure): a small render of the built-in scene in every mode and of a
 * triangle scene through the oracle's BVH, under -fsanitize=address,undefined. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kl_oracle.h"

int main(void) {
    ko_params p; ko_default_params(&p);
    ko_camera cam; ko_default_camera(&cam);
    p.width = 48; p.height = 27; p.max_frames = 0;
    float *img = (float *)malloc((size_t)p.width * p.height * 16);
    ko_scene *s = ko_scene_builtin();
    for (uint32_t mode = 0; mode < 3; mode++) {
        p.mode = mode; p.spp = mode == 1 ? 2 : 1;
        if (ko_render_raw(s, &cam, &p, 3, 0, p.height, img, NULL, NULL, 2) != 0) return 2;
    }
    ko_scene_destroy(s);
    enum { N = 300 };
    ko_tri *t = (ko_tri *)calloc(N, sizeof *t);
    ko_material m[2]; memset(m, 0, sizeof m);
    m[0].color[0] = m[0].color[1] = m[0].color[2] = 1.0f; m[0].roughness = 1.0f; m[1] = m[0]; m[1].type = KO_GLASS; m[1].ior = 1.2f;
    unsigned seed = 12345;
    for (int i = 0; i < N; i++) {
        float c[3];
        for (int k = 0; k < 3; k++) { seed = seed * 1664525u + 1013904223u; c[k] = (float)(seed >> 8) / 16777216.0f * 3.0f - 1.5f; }
        c[1] += 1.5f; c[2] -= 2.0f;
        for (int k = 0; k < 3; k++) { t[i].v0[k] = c[k]; t[i].v1[k] = c[k] + (k == 0 ? 0.2f : 0.0f); t[i].v2[k] = c[k] + (k == 1 ? 0.2f : 0.0f); }
        t[i].material = (uint32_t)(i & 1);
    }
    s = ko_scene_create(NULL, 0, NULL, 0, t, N, m, 2);
    p.mode = KO_MODE_PT; p.use_bvh = 1; p.max_bounces = 6; p.spp = 1;
    if (!s || ko_render_raw(s, &cam, &p, 0, 0, p.height, img, NULL, NULL, 2) != 0) return 3;
    ko_scene_destroy(s);
    free(t); free(img);
    printf("{\"oracle\": \"ok\"}\n");
    return 0;
}
