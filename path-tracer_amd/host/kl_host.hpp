// Headless host side above the C ABI: the state and the frame loop body the reference keeps in
// GLSL_Renderer (include/Rendering/Opengl/GLSL_Renderer.h:29-75, source/.../GLSL_Renderer.cpp)
// and Camera (include/Object/Camera.h, source/Object/Camera.cpp), with the OpenGL dispatch replaced
// by kl_* calls and the GLFW callbacks turned into plain methods that a caller (or a test) invokes.
// Field and method names follow the reference so that code written against it reads the same.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "Math/kl_render.h"
#include "kl_image.hpp"

namespace klhost {

struct dvec3 { double x, y, z; };

// source/Object/Camera.cpp:3-43
struct Camera {
    double focal_length, sensor_width;
    uint16_t width, height;
    dvec3 position, rotation, x_vector, y_vector, z_vector;

    Camera();
    void f_move(double i_x, double i_y, double i_z, double i_speed);
    void f_rotate(double i_yaw, double i_pitch);     // pitch clamped to +-89 degrees
    void f_compile();                                // yaw/pitch -> basis vectors
    kl_camera f_upload() const;                      // double -> float narrowing of GLSL_Renderer.cpp:293-297
};

struct Render_Mode { enum Enum { AMBIENT_OCCLUSION = 0, PATHTRACED = 1, ZBUFFER = 2 }; };
Render_Mode::Enum switchRenderMode(Render_Mode::Enum i_current);   // GLSL_Renderer.h:23-27

// key / button / action codes of the events the reference reacts to (GLFW's values)
enum { KEY_SPACE = 32, KEY_A = 65, KEY_C = 67, KEY_D = 68, KEY_E = 69, KEY_P = 80, KEY_Q = 81, KEY_R = 82, KEY_S = 83,
       KEY_V = 86, KEY_W = 87, KEY_ESCAPE = 256, KEY_LEFT_CONTROL = 341, MOUSE_BUTTON_RIGHT = 1, RELEASE = 0, PRESS = 1 };

struct GLSL_Renderer {
    double iTime;
    size_t iFrame;
    uint32_t iResolution[2];
    bool iBidirectional, iCameraChange;
    Camera camera;
    Render_Mode::Enum render_mode;
    double camera_move_sensitivity, camera_view_sensitivity;
    std::vector<bool> keys;
    double last_mouse[2];
    double frame_time;
    bool pause, should_close;

    kl_ctx *ctx;
    std::string error;

    GLSL_Renderer();
    ~GLSL_Renderer();
    GLSL_Renderer(const GLSL_Renderer &) = delete;
    GLSL_Renderer &operator=(const GLSL_Renderer &) = delete;

    // f_init of the reference minus window and loop: context, built-in scene, optional textures. false on error.
    bool f_init(int device = 0, const std::string &hdri_png = "", const std::string &albedo_png = "");
    void f_recompile();                                   // GLSL_Renderer.cpp:48-65
    bool f_frame();                                       // one pass of the loop body, :244-356
    bool f_run(size_t frames);                            // the while loop, bounded

    // the five callbacks, :67-162
    void framebuffer_size_callback(int width, int height);
    void cursor_position_callback(double xpos, double ypos);
    void mouse_button_callback(int button, int action, double xpos, double ypos);
    void scroll_callback(double yoffset);
    void key_callback(int key, int action);

    bool f_read_accumulation(std::vector<float> &rgba);   // what accumulation_tex holds
    bool f_save_png(const std::string &path);             // what the present pass puts on screen, top row first

  private:
    bool push_state();
    bool dirty;
};

}  // namespace klhost
