/*
 * ref_harness.c -- TEST INFRASTRUCTURE, not part of the product.
 *
 * Runs the reference's own GLSL shaders (read at run time from the reference
 * checkout, never copied into this repository) on Mesa llvmpipe without X,
 * EGL or GLFW, by loading the software-rasteriser DRI driver directly.
 * It re-issues the uniform uploads and draws of the reference host loop
 * (source/Rendering/Opengl/GLSL_Renderer.cpp:281-328) and reads the float
 * result back, so that golden vectors and CPU-baseline timings come from the
 * reference implementation itself.
 *
 * Declared deviations from the shipped application (all optional except 1):
 *   1. colour attachments are GL_RGBA32F instead of the unsized GL_RGBA of
 *      source/Rendering/Opengl/FBT.cpp:6 so float values are observable;
 *   2. --zero-col turns `vec3 col;` (RawFrame.glsl:433,445) into
 *      `vec3 col = vec3(0);`;
 *   3. --bounces / --samples / --spp rewrite the RAY_BOUNCES / SAMPLES / SPP defines
 *      (RawFrame.glsl:36-38); --define NAME VALUE rewrites any other one-line
 *      #define of RawFrame.glsl (HDRI_STRENGTH, AO_LENGTH, MAX_DIST, EPSILON: :32-40);
 *   4. the accumulation pass ping-pongs between two textures instead of
 *      sampling the texture it renders to (GLSL_Renderer.cpp:312-324).
 *
 * Build: see oracle/Makefile (target _ref/glsl_ref); glad.c is compiled from
 * where it lies in the reference checkout.
 * Env:   MESA_GL_VERSION_OVERRIDE=4.6 MESA_GLSL_VERSION_OVERRIDE=460
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <glad/glad.h>
#include <GL/internal/dri_interface.h>

static int g_w = 64, g_h = 36;

static void ld_getDrawableInfo(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) {
    (void)d; (void)p; *x = 0; *y = 0; *w = g_w; *h = g_h;
}
static void ld_putImage(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}
static void ld_getImage(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) {
    (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4);
}
static void ld_putImage2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p;
}
static void ld_getImage2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p) {
    (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)stride * h);
}

static const __DRIswrastLoaderExtension g_loader = {
    .base = { __DRI_SWRAST_LOADER, 3 },
    .getDrawableInfo = ld_getDrawableInfo,
    .putImage = ld_putImage,
    .getImage = ld_getImage,
    .putImage2 = ld_putImage2,
    .getImage2 = ld_getImage2,
};
static const __DRIextension *g_loader_exts[] = { &g_loader.base, NULL };

static void *(*g_gpa)(const char *);
static void *gl_loader(const char *name) { return g_gpa(name); }

static char *read_file(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    char *buf = malloc((size_t)n + 1);
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path); exit(2); }
    buf[n] = 0; fclose(f);
    if (len) *len = (size_t)n;
    return buf;
}

/* replace the first occurrence of `from` by `to`; returns a new string */
static char *replace_once(char *src, const char *from, const char *to, int required) {
    char *p = strstr(src, from);
    if (!p) {
        if (required) { fprintf(stderr, "patch anchor not found: %s\n", from); exit(2); }
        return src;
    }
    size_t n = strlen(src), nf = strlen(from), nt = strlen(to);
    char *out = malloc(n - nf + nt + 1);
    memcpy(out, src, (size_t)(p - src));
    memcpy(out + (p - src), to, nt);
    strcpy(out + (p - src) + nt, p + nf);
    free(src);
    return out;
}
static char *replace_all(char *src, const char *from, const char *to) {
    int guard = 0;
    while (strstr(src, from) && guard++ < 16) src = replace_once(src, from, to, 1);
    return src;
}

/* the line `#define NAME <anything>` becomes `#define NAME VALUE` */
static char *redefine(char *src, const char *name, const char *value) {
    char key[128];
    snprintf(key, sizeof key, "#define %s ", name);
    char *p = strstr(src, key);
    if (!p || (p != src && p[-1] != '\n')) { fprintf(stderr, "no #define %s in the shader\n", name); exit(2); }
    char *eol = strchr(p, '\n');
    size_t n = strlen(src), head = (size_t)(p - src), tail = eol ? (size_t)(eol - src) : n;
    char *out = malloc(n + strlen(key) + strlen(value) + 2);
    memcpy(out, src, head);
    int w = sprintf(out + head, "%s%s", key, value);
    strcpy(out + head + w, src + tail);
    free(src);
    return out;
}

static GLuint compile(GLenum type, const char *src, const char *name) {
    GLuint s = glCreateShader(type);
    glShaderSource(s, 1, &src, NULL);
    glCompileShader(s);
    GLint ok = 0; glGetShaderiv(s, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[8192]; glGetShaderInfoLog(s, sizeof log, NULL, log);
        fprintf(stderr, "compile %s failed:\n%s\n", name, log); exit(3);
    }
    return s;
}
static GLuint link_program(GLuint vs, GLuint fs, const char *name) {
    GLuint p = glCreateProgram();
    glAttachShader(p, vs); glAttachShader(p, fs);
    glLinkProgram(p);
    GLint ok = 0; glGetProgramiv(p, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[8192]; glGetProgramInfoLog(p, sizeof log, NULL, log);
        fprintf(stderr, "link %s failed:\n%s\n", name, log); exit(3);
    }
    return p;
}

static int g_shipped_targets = 0;   /* --shipped-targets: unsized GL_RGBA attachments exactly as FBT.cpp:6 creates them (RGBA8 on llvmpipe) */

static GLuint make_f32_target(int w, int h, GLuint *fbo) {
    GLuint tex;
    glGenTextures(1, &tex);
    glBindTexture(GL_TEXTURE_2D, tex);
    glTexImage2D(GL_TEXTURE_2D, 0, g_shipped_targets ? GL_RGBA : GL_RGBA32F, w, h, 0, GL_RGBA, GL_FLOAT, NULL);
    /* NEAREST instead of the reference's LINEAR (FBT.cpp:8-9): the accumulation
       pass samples at exact texel centres, so the filter does not change values. */
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    glGenFramebuffers(1, fbo);
    glBindFramebuffer(GL_FRAMEBUFFER, *fbo);
    glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex, 0);
    if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) {
        fprintf(stderr, "fbo incomplete\n"); exit(3);
    }
    glBindFramebuffer(GL_FRAMEBUFFER, 0);
    return tex;
}

/* texture file: raw 8-bit pixels, rows top-to-bottom as decoded (no flip),
   same upload as source/Rendering/Opengl/Texture.cpp:3-16 */
static GLuint make_tex8(const char *path, int w, int h, int ch) {
    GLuint tex;
    glGenTextures(1, &tex);
    glBindTexture(GL_TEXTURE_2D, tex);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    GLenum fmt = ch == 4 ? GL_RGBA : ch == 3 ? GL_RGB : GL_RED;
    if (path) {
        size_t n; char *data = read_file(path, &n);
        if (n != (size_t)w * h * ch) { fprintf(stderr, "texture %s: size mismatch\n", path); exit(2); }
        glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, w, h, 0, fmt, GL_UNSIGNED_BYTE, data);
        free(data);
    } else {
        unsigned char white[4] = {255, 255, 255, 255};
        glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, 1, 1, 0, GL_RGBA, GL_UNSIGNED_BYTE, white);
    }
    return tex;
}

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv) {
    const char *shader_dir = "/root/reference/resources/Shaders";
    const char *out_raw = NULL, *out_acc = NULL;
    const char *hdri_path = NULL, *albedo_path = NULL;
    int hdri_w = 1, hdri_h = 1, hdri_c = 4, alb_w = 1, alb_h = 1, alb_c = 4;
    int mode = 2, first_frame = 0, nframes = 1, zero_col = 0, bounces = -1, samples = -1, spp = -1, timing = 0;
    float cam_pos[3] = {0, 1, 5}, cam_front[3] = {0, 0, -1}, cam_up[3] = {0, 1, 0};
    float focal = 0.05f, sensor = 0.036f;
    const char *def_name[8], *def_value[8]; int ndefs = 0;

    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
#define NEXT (i + 1 < argc ? argv[++i] : (fprintf(stderr, "missing value for %s\n", a), exit(2), ""))
        if (!strcmp(a, "--shaders")) shader_dir = NEXT;
        else if (!strcmp(a, "--w")) g_w = atoi(NEXT);
        else if (!strcmp(a, "--h")) g_h = atoi(NEXT);
        else if (!strcmp(a, "--mode")) mode = atoi(NEXT);
        else if (!strcmp(a, "--frame")) first_frame = atoi(NEXT);
        else if (!strcmp(a, "--frames")) nframes = atoi(NEXT);
        else if (!strcmp(a, "--zero-col")) zero_col = 1;
        else if (!strcmp(a, "--bounces")) bounces = atoi(NEXT);
        else if (!strcmp(a, "--samples")) samples = atoi(NEXT);
        else if (!strcmp(a, "--spp")) spp = atoi(NEXT);
        else if (!strcmp(a, "--timing")) timing = 1;
        else if (!strcmp(a, "--shipped-targets")) g_shipped_targets = 1;
        else if (!strcmp(a, "--out-raw")) out_raw = NEXT;
        else if (!strcmp(a, "--out-acc")) out_acc = NEXT;
        else if (!strcmp(a, "--hdri")) { hdri_path = NEXT; hdri_w = atoi(NEXT); hdri_h = atoi(NEXT); hdri_c = atoi(NEXT); }
        else if (!strcmp(a, "--albedo")) { albedo_path = NEXT; alb_w = atoi(NEXT); alb_h = atoi(NEXT); alb_c = atoi(NEXT); }
        else if (!strcmp(a, "--cam")) {
            for (int k = 0; k < 3; k++) cam_pos[k] = (float)atof(NEXT);
            for (int k = 0; k < 3; k++) cam_front[k] = (float)atof(NEXT);
            for (int k = 0; k < 3; k++) cam_up[k] = (float)atof(NEXT);
        }
        else if (!strcmp(a, "--focal")) focal = (float)atof(NEXT);
        else if (!strcmp(a, "--sensor")) sensor = (float)atof(NEXT);
        else if (!strcmp(a, "--define")) { if (ndefs == 8) { fprintf(stderr, "too many --define\n"); return 2; } def_name[ndefs] = NEXT; def_value[ndefs++] = NEXT; }
        else { fprintf(stderr, "unknown option %s\n", a); return 2; }
    }
    setenv("MESA_GL_VERSION_OVERRIDE", "4.6", 0);
    setenv("MESA_GLSL_VERSION_OVERRIDE", "460", 0);

    /* --- context creation through the swrast DRI driver --- */
    void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) { fprintf(stderr, "libglapi: %s\n", dlerror()); return 4; }
    g_gpa = (void *(*)(const char *))dlsym(glapi, "_glapi_get_proc_address");
    void *drv = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) drv = dlopen("swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) { fprintf(stderr, "swrast_dri: %s\n", dlerror()); return 4; }
    const __DRIextension **(*get_exts)(void) =
        (const __DRIextension **(*)(void))dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!g_gpa || !get_exts) { fprintf(stderr, "DRI entry points missing\n"); return 4; }
    const __DRIextension **exts = get_exts();
    const __DRIcoreExtension *core = NULL;
    const __DRIswrastExtension *swrast = NULL;
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension *)exts[i];
    }
    if (!core || !swrast || swrast->base.version < 4) { fprintf(stderr, "DRI core/swrast ext missing\n"); return 4; }
    const __DRIconfig **configs = NULL;
    __DRIscreen *scr = swrast->createNewScreen2(0, g_loader_exts, exts, &configs, NULL);
    if (!scr || !configs) { fprintf(stderr, "createNewScreen2 failed\n"); return 4; }
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 6 };
    unsigned err = 0;
    __DRIcontext *ctx = swrast->createContextAttribs(scr, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!ctx) { fprintf(stderr, "createContextAttribs failed err=%u\n", err); return 4; }
    __DRIdrawable *dr = swrast->createNewDrawable(scr, configs[0], NULL);
    if (!dr || !core->bindContext(ctx, dr, dr)) { fprintf(stderr, "bindContext failed\n"); return 4; }
    if (!gladLoadGLLoader(gl_loader)) { fprintf(stderr, "glad load failed\n"); return 4; }
    fprintf(stderr, "GL_RENDERER: %s\nGL_VERSION: %s\n", glGetString(GL_RENDERER), glGetString(GL_VERSION));

    /* --- shaders, read from the reference checkout, with declared patches --- */
    char path[1024];
    snprintf(path, sizeof path, "%s/Vert.glsl", shader_dir);
    char *vs_src = read_file(path, NULL);
    snprintf(path, sizeof path, "%s/RawFrame.glsl", shader_dir);
    char *raw_src = read_file(path, NULL);
    snprintf(path, sizeof path, "%s/Accumulation.glsl", shader_dir);
    char *acc_src = read_file(path, NULL);
    if (zero_col) raw_src = replace_all(raw_src, "vec3 col;", "vec3 col = vec3(0);");
    if (bounces >= 0) {
        char to[64]; snprintf(to, sizeof to, "#define RAY_BOUNCES   %d", bounces);
        raw_src = replace_once(raw_src, "#define RAY_BOUNCES   32", to, 1);
    }
    if (spp >= 0) {
        char to[64]; snprintf(to, sizeof to, "#define SPP           %d", spp);
        raw_src = replace_once(raw_src, "#define SPP           1", to, 1);
    }
    if (samples >= 0) {
        char to[64]; snprintf(to, sizeof to, "#define SAMPLES       %d", samples);
        raw_src = replace_once(raw_src, "#define SAMPLES       30", to, 1);
    }
    for (int k = 0; k < ndefs; k++) raw_src = redefine(raw_src, def_name[k], def_value[k]);
    GLuint vs = compile(GL_VERTEX_SHADER, vs_src, "Vert");
    GLuint raw_prog = link_program(vs, compile(GL_FRAGMENT_SHADER, raw_src, "RawFrame"), "RawFrame");
    GLuint acc_prog = link_program(vs, compile(GL_FRAGMENT_SHADER, acc_src, "Accumulation"), "Accumulation");

    /* --- full-screen quad, GLSL_Renderer.cpp:4-13,207-219 --- */
    const float verts[] = { -1, -1, 0, 0,  1, -1, 1, 0,  1, 1, 1, 1,  -1, 1, 0, 1 };
    const unsigned faces[] = { 0, 1, 2, 2, 3, 0 };
    GLuint vao, vbo, ebo;
    glGenVertexArrays(1, &vao); glBindVertexArray(vao);
    glGenBuffers(1, &vbo); glBindBuffer(GL_ARRAY_BUFFER, vbo);
    glBufferData(GL_ARRAY_BUFFER, sizeof verts, verts, GL_STATIC_DRAW);
    glGenBuffers(1, &ebo); glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, ebo);
    glBufferData(GL_ELEMENT_ARRAY_BUFFER, sizeof faces, faces, GL_STATIC_DRAW);
    glVertexAttribPointer(0, 2, GL_FLOAT, GL_FALSE, 4 * sizeof(float), (void *)0);
    glEnableVertexAttribArray(0);
    glVertexAttribPointer(1, 2, GL_FLOAT, GL_FALSE, 4 * sizeof(float), (void *)(2 * sizeof(float)));
    glEnableVertexAttribArray(1);

    GLuint raw_fbo, acc_fbo[2];
    GLuint raw_tex = make_f32_target(g_w, g_h, &raw_fbo);
    GLuint acc_tex[2];
    acc_tex[0] = make_f32_target(g_w, g_h, &acc_fbo[0]);
    acc_tex[1] = make_f32_target(g_w, g_h, &acc_fbo[1]);
    GLuint hdri_tex = make_tex8(hdri_path, hdri_w, hdri_h, hdri_c);
    GLuint alb_tex = make_tex8(albedo_path, alb_w, alb_h, alb_c);

    glViewport(0, 0, g_w, g_h);
    glClearColor(0, 0, 0, 1);
    for (int k = 0; k < 2; k++) { glBindFramebuffer(GL_FRAMEBUFFER, acc_fbo[k]); glClear(GL_COLOR_BUFFER_BIT); }

    size_t npx = (size_t)g_w * g_h;
    float *buf = malloc(npx * 4 * sizeof(float));
    FILE *fraw = out_raw ? fopen(out_raw, "wb") : NULL;
    int cur = 0; /* acc_tex[cur] holds the last accumulated frame */
    double t_total = 0.0; int timed = 0;

    for (int f = 0; f < nframes + (timing ? 1 : 0); f++) {
        /* in timing mode the first draw is an untimed warm-up (JIT) of frame first_frame */
        int frame = first_frame + (timing ? (f == 0 ? 0 : f - 1) : f);
        int camera_change = (frame == 0);
        double t0 = now_s();
        /* PASS 1: raw frame, GLSL_Renderer.cpp:283-306 */
        glBindFramebuffer(GL_FRAMEBUFFER, raw_fbo);
        glClear(GL_COLOR_BUFFER_BIT);
        glUseProgram(raw_prog);
        glUniform1f(glGetUniformLocation(raw_prog, "iTime"), 0.0f);
        glUniform1ui(glGetUniformLocation(raw_prog, "iFrame"), (GLuint)frame);
        glUniform2f(glGetUniformLocation(raw_prog, "iResolution"), (float)g_w, (float)g_h);
        glUniform1ui(glGetUniformLocation(raw_prog, "iRenderMode"), (GLuint)mode);
        glUniform1i(glGetUniformLocation(raw_prog, "iBidirectional"), 1);
        glUniform1f(glGetUniformLocation(raw_prog, "iCameraFocalLength"), focal);
        glUniform1f(glGetUniformLocation(raw_prog, "iCameraSensorWidth"), sensor);
        glUniform3fv(glGetUniformLocation(raw_prog, "iCameraPos"), 1, cam_pos);
        glUniform3fv(glGetUniformLocation(raw_prog, "iCameraFront"), 1, cam_front);
        glUniform3fv(glGetUniformLocation(raw_prog, "iCameraUp"), 1, cam_up);
        glUniform1i(glGetUniformLocation(raw_prog, "iCameraChange"), camera_change);
        glActiveTexture(GL_TEXTURE1); glBindTexture(GL_TEXTURE_2D, acc_tex[cur]);
        glUniform1i(glGetUniformLocation(raw_prog, "iLastFrame"), 1);
        glActiveTexture(GL_TEXTURE3); glBindTexture(GL_TEXTURE_2D, alb_tex);
        glUniform1i(glGetUniformLocation(raw_prog, "iAlbedo"), 3);
        glActiveTexture(GL_TEXTURE2); glBindTexture(GL_TEXTURE_2D, hdri_tex);
        glUniform1i(glGetUniformLocation(raw_prog, "iHdri"), 2);
        glDrawElements(GL_TRIANGLES, 6, GL_UNSIGNED_INT, 0);
        if (timing) {
            glFinish();
            double t1 = now_s();
            if (f > 0) { t_total += t1 - t0; timed++; }
            continue;
        }
        if (fraw) {
            glReadPixels(0, 0, g_w, g_h, GL_RGBA, GL_FLOAT, buf);
            fwrite(buf, sizeof(float), npx * 4, fraw);
        }
        /* PASS 2: accumulation, GLSL_Renderer.cpp:312-324 (ping-pong) */
        int nxt = cur ^ 1;
        glBindFramebuffer(GL_FRAMEBUFFER, acc_fbo[nxt]);
        glUseProgram(acc_prog);
        glUniform1f(glGetUniformLocation(acc_prog, "iTime"), 0.0f);
        glUniform1ui(glGetUniformLocation(acc_prog, "iFrame"), (GLuint)frame);
        glUniform1ui(glGetUniformLocation(acc_prog, "iRenderMode"), (GLuint)mode);
        glUniform1i(glGetUniformLocation(acc_prog, "iCameraChange"), camera_change);
        glActiveTexture(GL_TEXTURE0); glBindTexture(GL_TEXTURE_2D, raw_tex);
        glUniform1i(glGetUniformLocation(acc_prog, "iRawFrame"), 0);
        glActiveTexture(GL_TEXTURE1); glBindTexture(GL_TEXTURE_2D, acc_tex[cur]);
        glUniform1i(glGetUniformLocation(acc_prog, "iLastFrame"), 1);
        glDrawElements(GL_TRIANGLES, 6, GL_UNSIGNED_INT, 0);
        cur = nxt;
    }
    if (fraw) fclose(fraw);
    if (timing) {
        printf("{\"frames\": %d, \"seconds\": %.6f, \"ms_per_frame\": %.4f, \"samples_per_s\": %.1f, \"w\": %d, \"h\": %d, \"mode\": %d}\n",
               timed, t_total, 1e3 * t_total / (timed ? timed : 1), npx * (double)timed / t_total, g_w, g_h, mode);
    }
    if (out_acc && !timing) {
        glBindFramebuffer(GL_FRAMEBUFFER, acc_fbo[cur]);
        glReadPixels(0, 0, g_w, g_h, GL_RGBA, GL_FLOAT, buf);
        FILE *f = fopen(out_acc, "wb");
        fwrite(buf, sizeof(float), npx * 4, f);
        fclose(f);
    }
    free(buf);
    return 0;
}
