// Shader.h — the reference's ComputeProgram (src/Shader.h:20-37) reduced to what survives without GLSL: the
// uniform setters the frame loop calls.  They forward to the C ABI; Use/Unuse are no-ops kept for source
// compatibility with main().  ShaderProgram (the fullscreen-quad display pass) becomes an optional writer
// of the accumulated image (sRGB-encoded PPM), which is all "display" means in a headless build.
#pragma once
#include <string>

#include "Camera.h"

struct ComputeProgram {
    // The reference compiles the file it is given (src/Shader.cpp:41-70).  Here the file NAME selects which of the reference's
    // two copies of the kernel the same HIP kernels reproduce: ".../rt.frag" -> the legacy literals (pt_shading_preset 1,
    // SURVEY App. B), anything else -> rt.comp's (preset 0).  Call after Renderer::Init.
    explicit ComputeProgram(const char *filepath = "");
    bool legacy() const { return isLegacy; }
    void Use() {}
    void Unuse() {}
    void SetUniform1i(const char *uName, int i);     // "currAccumPass" (remembered), "debugNormal", "maxBounces"
    void SetUniformCamera(Camera &cam);              // cam.position + cam.inverseView -> pt_set_camera
    void Dispatch();                                 // glDispatchCompute(W/8, H/8, 1) + barrier -> pt_render_pass(currAccumPass)
    int currAccumPass = 0;
private:
    std::string path;
    bool isLegacy = false;
};

struct ShaderProgram {
    ShaderProgram(const char * = "", const char * = "") {}
    void Use() {}
    void Unuse() {}
    // GL_FRAMEBUFFER_SRGB + textured quad (src/Renderer.cpp:135, compute_accum.frag:9-11): write the image as 8-bit sRGB PPM
    bool Present(const char *ppmPath);
    // the same transfer (clamp, linear -> sRGB, 8 bits, bottom row last) for an RGBA32F image already on the host
    static bool WritePPM(const float *rgba, int width, int height, const char *ppmPath);
};
