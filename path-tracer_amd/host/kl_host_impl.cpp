#include "kl_host.hpp"

#include <chrono>
#include <cmath>

namespace klhost {
namespace {
const double DEG_RAD = 0.01745329252;   // include/Include.h value of the reference
dvec3 add(dvec3 a, dvec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
dvec3 mul(dvec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
dvec3 cross(dvec3 a, dvec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
dvec3 normalize(dvec3 a) { double l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); return {a.x / l, a.y / l, a.z / l}; }
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

Camera::Camera() {
    focal_length = 0.05; sensor_width = 0.036; width = 1920; height = 1080;
    position = {0, 1, 5}; rotation = {-90, 0, 0};
    x_vector = {1, 0, 0}; y_vector = {0, 1, 0}; z_vector = {0, 0, -1};
}
void Camera::f_move(double i_x, double i_y, double i_z, double i_speed) {
    position = add(position, mul(x_vector, i_x * i_speed));
    position = add(position, mul(y_vector, i_y * i_speed));
    position = add(position, mul(z_vector, i_z * i_speed));
}
void Camera::f_rotate(double i_yaw, double i_pitch) {
    rotation.x += i_yaw; rotation.y += i_pitch;
    if (rotation.y > 89.0) rotation.y = 89.0;
    if (rotation.y < -89.0) rotation.y = -89.0;
    f_compile();
}
void Camera::f_compile() {
    const double yaw = rotation.x * DEG_RAD, pitch = rotation.y * DEG_RAD;
    z_vector = normalize({std::cos(yaw) * std::cos(pitch), std::sin(pitch), std::sin(yaw) * std::cos(pitch)});
    x_vector = normalize(cross(z_vector, {0, 1, 0}));
    y_vector = normalize(cross(x_vector, z_vector));
}
kl_camera Camera::f_upload() const {
    kl_camera c;
    c.pos[0] = (float)position.x; c.pos[1] = (float)position.y; c.pos[2] = (float)position.z;
    c.front[0] = (float)z_vector.x; c.front[1] = (float)z_vector.y; c.front[2] = (float)z_vector.z;
    c.up[0] = (float)y_vector.x; c.up[1] = (float)y_vector.y; c.up[2] = (float)y_vector.z;
    c.focal = (float)focal_length; c.sensor = (float)sensor_width;
    return c;
}

Render_Mode::Enum switchRenderMode(Render_Mode::Enum i_current) {
    int v = (int)i_current + 1;
    return v > 2 ? (Render_Mode::Enum)0 : (Render_Mode::Enum)v;
}

GLSL_Renderer::GLSL_Renderer() {
    iTime = 0.0; iFrame = 0; iResolution[0] = 1920; iResolution[1] = 1080;
    iBidirectional = true; iCameraChange = false;
    render_mode = Render_Mode::PATHTRACED;
    camera_move_sensitivity = 0.15; camera_view_sensitivity = 0.075;
    keys = std::vector<bool>(348, false);
    last_mouse[0] = iResolution[0] / 2.0; last_mouse[1] = iResolution[1] / 2.0;
    frame_time = 0.0; pause = false; should_close = false;
    ctx = nullptr; dirty = true;
}
GLSL_Renderer::~GLSL_Renderer() { if (ctx) kl_destroy(ctx); }

bool GLSL_Renderer::f_init(int device, const std::string &hdri_png, const std::string &albedo_png) {
    kl_status s = kl_create(&ctx, device, iResolution[0], iResolution[1]);
    if (s != KL_OK) { error = std::string("kl_create: ") + kl_status_string(s); return false; }
    if ((s = kl_load_builtin_scene(ctx)) != KL_OK) { error = kl_last_error(ctx); return false; }
    // a missing texture leaves the 1x1 white default, like the silent failure of Texture::f_init (Texture.cpp:5)
    Image img;
    if (!hdri_png.empty() && img.f_load(hdri_png)) kl_upload_texture(ctx, KL_TEX_HDRI, img.data.data(), img.width, img.height, img.channels);
    if (!albedo_png.empty() && img.f_load(albedo_png)) kl_upload_texture(ctx, KL_TEX_ALBEDO, img.data.data(), img.width, img.height, img.channels);
    dirty = true;
    return true;
}

void GLSL_Renderer::f_recompile() {
    if (ctx) kl_recompile(ctx);
    iCameraChange = true; iFrame = 0;
    dirty = true;
}

// every reset funnels here: the context restarts its accumulation when camera or mode are (re)sent
bool GLSL_Renderer::push_state() {
    if (!ctx) { error = "f_init was not called"; return false; }
    if (!dirty) return true;
    kl_camera c = camera.f_upload();
    if (kl_set_mode(ctx, (uint32_t)render_mode) != KL_OK || kl_set_camera(ctx, &c) != KL_OK) { error = kl_last_error(ctx); return false; }
    dirty = false;
    return true;
}

bool GLSL_Renderer::f_frame() {
    if (pause) return true;
    struct { int key; double x, y, z; } moves[] = {
        {KEY_D, 1, 0, 0}, {KEY_A, -1, 0, 0}, {KEY_E, 0, 1, 0}, {KEY_SPACE, 0, 1, 0},
        {KEY_Q, 0, -1, 0}, {KEY_LEFT_CONTROL, 0, -1, 0}, {KEY_W, 0, 0, 1}, {KEY_S, 0, 0, -1}};
    bool moved_up = false, moved_down = false;
    for (auto &m : moves) {
        if (!keys[m.key]) continue;
        // E/SPACE and Q/CTRL are one `||` each in the reference (:255,:260): a pair moves once
        if (m.y > 0) { if (moved_up) continue; moved_up = true; }
        if (m.y < 0) { if (moved_down) continue; moved_down = true; }
        camera.f_move(m.x, m.y, m.z, camera_move_sensitivity);
        iCameraChange = true; iFrame = 0; dirty = true;
    }
    const double t0 = now_s();
    if (iCameraChange) dirty = true;
    if (!push_state()) return false;
    if (kl_render_frames(ctx, 1, (uint32_t)iFrame) != KL_OK || kl_synchronize(ctx) != KL_OK) { error = kl_last_error(ctx); return false; }
    frame_time = now_s() - t0;
    iTime += frame_time;
    iFrame++;
    iCameraChange = false;
    return true;
}

bool GLSL_Renderer::f_run(size_t frames) {
    for (size_t i = 0; i < frames && !should_close; i++) if (!f_frame()) return false;
    return true;
}

void GLSL_Renderer::framebuffer_size_callback(int width, int height) {
    iResolution[0] = (uint32_t)width; iResolution[1] = (uint32_t)height;
    iFrame = 0;
    if (ctx) kl_resize(ctx, iResolution[0], iResolution[1]);
}
void GLSL_Renderer::cursor_position_callback(double xpos, double ypos) {
    if (!keys[MOUSE_BUTTON_RIGHT]) return;
    double xoffset = xpos - last_mouse[0], yoffset = last_mouse[1] - ypos;
    last_mouse[0] = xpos; last_mouse[1] = ypos;
    camera.f_rotate(xoffset * camera_view_sensitivity, yoffset * camera_view_sensitivity);
    iCameraChange = true; iFrame = 0; dirty = true;
}
void GLSL_Renderer::mouse_button_callback(int button, int action, double xpos, double ypos) {
    if (action == PRESS) keys[button] = true;
    else if (action == RELEASE) keys[button] = false;
    if (button == MOUSE_BUTTON_RIGHT) { last_mouse[0] = xpos; last_mouse[1] = ypos; }
}
void GLSL_Renderer::scroll_callback(double yoffset) {
    if (yoffset < 0) { iCameraChange = true; iFrame = 0; dirty = true; camera_move_sensitivity /= 1.1; }
    if (yoffset > 0) { iCameraChange = true; iFrame = 0; dirty = true; camera_move_sensitivity *= 1.1; }
}
void GLSL_Renderer::key_callback(int key, int action) {
    if (key == KEY_R && action == PRESS) f_recompile();
    if (key == KEY_V && action == PRESS) { render_mode = switchRenderMode(render_mode); iCameraChange = true; iFrame = 0; dirty = true; }
    if (key == KEY_C && action == PRESS) { camera = Camera(); iCameraChange = true; iFrame = 0; dirty = true; }
    if (key == KEY_ESCAPE && action == PRESS) should_close = true;
    if (key == KEY_P && action == PRESS) pause = !pause;
    if (key >= 0 && key < (int)keys.size()) {
        if (action == PRESS) keys[key] = true;
        else if (action == RELEASE) keys[key] = false;
    }
}

bool GLSL_Renderer::f_read_accumulation(std::vector<float> &rgba) {
    if (!ctx) return false;
    rgba.resize((size_t)iResolution[0] * iResolution[1] * 4);
    if (kl_read_accum(ctx, rgba.data(), rgba.size() * sizeof(float)) != KL_OK) { error = kl_last_error(ctx); return false; }
    return true;
}
bool GLSL_Renderer::f_save_png(const std::string &path) {
    if (!ctx) return false;
    Image img; img.width = (int)iResolution[0]; img.height = (int)iResolution[1]; img.channels = 4;
    std::vector<uint8_t> px((size_t)img.width * img.height * 4);
    if (kl_resolve(ctx, KL_TONEMAP_CLAMP, px.data(), px.size()) != KL_OK) { error = kl_last_error(ctx); return false; }
    img.data.resize(px.size());
    const size_t stride = (size_t)img.width * 4;
    for (int y = 0; y < img.height; y++)                     // buffer row 0 is the bottom of the picture
        std::copy(px.begin() + stride * y, px.begin() + stride * (y + 1), img.data.begin() + stride * (img.height - 1 - y));
    return img.f_save_png(path);
}

}  // namespace klhost
