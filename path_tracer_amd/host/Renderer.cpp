#include "Renderer.h"

#include <cstdio>

namespace Renderer {

pt_ctx *ctx = nullptr;
int screenWidth = 0;
int screenHeight = 0;
Camera camera;
Scene scene;

bool Check(int rc, const char *what)
{
    if (rc == PT_OK) return true;
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, pt_last_error(ctx));
    return false;
}

int Init(int width, int height, int device)
{
    if (width <= 0 || height <= 0) { width = 1920; height = 1080; }      // the reference falls back to the monitor size
    if (pt_create(&ctx, device) != PT_OK) {
        std::fprintf(stderr, "pt_create failed: %s\n", pt_last_error(nullptr));
        return 1;
    }
    screenWidth = width;
    screenHeight = height;
    if (!Check(pt_resize(ctx, width, height), "pt_resize")) return 1;
    std::printf("Screen size: %dx%d\n", screenWidth, screenHeight);
    return 0;
}

void Shutdown()
{
    pt_destroy(ctx);
    ctx = nullptr;
}

void MouseCallback(double xpos, double ypos)
{
    // src/Renderer.cpp:144-145: the differences are taken in double (xpos, ypos are doubles) and narrowed afterwards
    const float xOffset = (float)(xpos - (double)camera.mouseLastX), yOffset = (float)((double)camera.mouseLastY - ypos);
    camera.mouseLastX = (float)xpos;
    camera.mouseLastY = (float)ypos;
    camera.MouseLook(xOffset, yOffset);
}

}  // namespace Renderer
