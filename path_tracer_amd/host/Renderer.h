// Renderer.h — the reference's Renderer namespace (src/Renderer.h:30-45) without a window: Init creates the
// HIP context and the RGBA32F accumulation image instead of a GLFW window, GL context, quad VAO and
// accumulation texture (src/Renderer.cpp:33-140).  camera / scene / screenWidth / screenHeight keep their names.
#pragma once
#include "../../include/pt_hip.h"
#include "Camera.h"
#include "Object.h"
#include "Shader.h"

namespace Renderer {
extern pt_ctx *ctx;             // stands where window / accumTexID / rtFboID stood
extern int screenWidth;
extern int screenHeight;
extern Camera camera;
extern Scene scene;

int Init(int width, int height, int device = 0);    // 0 on success
void Shutdown();
void MouseCallback(double xpos, double ypos);       // src/Renderer.cpp:142-166
bool Check(int rc, const char *what);               // prints pt_last_error on failure
}  // namespace Renderer
