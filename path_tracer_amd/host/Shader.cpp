#include "Shader.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "Renderer.h"

ComputeProgram::ComputeProgram(const char *filepath) : path(filepath ? filepath : "")
{
    const std::string legacyName = "rt.frag";
    isLegacy = path.size() >= legacyName.size() && path.compare(path.size() - legacyName.size(), legacyName.size(), legacyName) == 0;
    if (!Renderer::ctx) return;                     // nothing to configure yet (the defaults are rt.comp's)
    PtShadingParams prm;
    if (Renderer::Check(pt_shading_preset(&prm, isLegacy ? 1 : 0), "pt_shading_preset"))
        Renderer::Check(pt_set_shading_params(Renderer::ctx, &prm), "pt_set_shading_params");
}

void ComputeProgram::SetUniform1i(const char *uName, int i)
{
    if (!std::strcmp(uName, "currAccumPass")) currAccumPass = i;
    else if (!std::strcmp(uName, "debugNormal")) Renderer::Check(pt_set_debug_normal(Renderer::ctx, i), "pt_set_debug_normal");
    else if (!std::strcmp(uName, "maxBounces")) Renderer::Check(pt_set_max_bounces(Renderer::ctx, i), "pt_set_max_bounces");
    else std::fprintf(stderr, "ComputeProgram: unknown uniform '%s'\n", uName);
}

void ComputeProgram::SetUniformCamera(Camera &cam)
{
    const float pos[3] = {cam.position.x, cam.position.y, cam.position.z};
    Renderer::Check(pt_set_camera(Renderer::ctx, pos, cam.inverseView.data()), "pt_set_camera");
}

void ComputeProgram::Dispatch() { Renderer::Check(pt_render_pass(Renderer::ctx, currAccumPass), "pt_render_pass"); }

static unsigned char srgb8(float v)
{
    if (!(v > 0.0f)) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    const float s = v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f;
    return (unsigned char)std::lround(s * 255.0f);
}

// The display step of the reference: the accumulation texture is drawn 1:1 onto the window (compute_accum.frag:9-11) with
// GL_FRAMEBUFFER_SRGB enabled (src/Renderer.cpp:135), i.e. clamp to [0,1], linear -> sRGB, 8 bits per channel; GL's texel
// origin is the bottom-left corner, a PPM's first row is the top one.
bool ShaderProgram::WritePPM(const float *rgba, int W, int H, const char *ppmPath)
{
    FILE *f = std::fopen(ppmPath, "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%d %d\n255\n", W, H);
    std::vector<unsigned char> row((size_t)W * 3);
    for (int y = H - 1; y >= 0; y--) {      // image row 0 is the bottom row (GL texel origin)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < 3; c++) row[(size_t)x * 3 + c] = srgb8(rgba[((size_t)y * W + x) * 4 + c]);
        std::fwrite(row.data(), 1, row.size(), f);
    }
    return std::fclose(f) == 0;
}

bool ShaderProgram::Present(const char *ppmPath)
{
    const int W = Renderer::screenWidth, H = Renderer::screenHeight;
    std::vector<float> img((size_t)W * H * 4);
    if (!Renderer::Check(pt_read_accum(Renderer::ctx, img.data(), img.size() * sizeof(float)), "pt_read_accum")) return false;
    return WritePPM(img.data(), W, H, ppmPath);
}
