// camera_input_ref — TEST INFRASTRUCTURE (oracle).  Drives the reference's OWN Camera (src/Camera.cpp, compiled where it
// lies together with this file, see oracle/Makefile; nothing is copied) through a scripted key sequence, the way the
// reference's frame loop does (src/Main.cpp:110-120): `if (moving) { currAccumPass = 0; moving = false; }`, then
// ProcessInput(window, dt), then UpdateView().  glfwGetKey is the one GLFW symbol Camera.cpp needs; it is defined here
// and reads the key set of the current step.
// usage: camera_input_ref x y z yaw pitch  KEYS:DT [KEYS:DT ...]      KEYS = letters out of WASDEQ, or '-' for none
// Output per step, hex float32 words: position(3) forward(3) right(3) up(3) inverseView(16 column-major), then `moving`
// as it stands after ProcessInput (what the NEXT frame's reset test sees).
// Used by oracle/gen_goldens.py to write tests/golden/camera_input.json.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "Camera.h"

static std::string g_keys;
extern "C" int glfwGetKey(GLFWwindow *, int key) { return g_keys.find((char)key) != std::string::npos ? GLFW_PRESS : GLFW_RELEASE; }

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static void put(const glm::vec3 &v) { std::printf("%08x %08x %08x ", bits(v.x), bits(v.y), bits(v.z)); }

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    Camera cam(glm::vec3(std::strtof(argv[1], nullptr), std::strtof(argv[2], nullptr), std::strtof(argv[3], nullptr)));
    cam.yaw = std::strtof(argv[4], nullptr); cam.pitch = std::strtof(argv[5], nullptr);
    cam.UpdateView();                       // the loop's previous frame left forward/right/up up to date
    for (int i = 6; i < argc; i++) {
        const char *colon = std::strchr(argv[i], ':');
        if (!colon) return 2;
        g_keys.assign(argv[i], colon - argv[i]);
        const float dt = std::strtof(colon + 1, nullptr);
        if (cam.moving) cam.moving = false; // src/Main.cpp:110-117
        cam.ProcessInput(nullptr, dt);      // src/Main.cpp:119
        cam.UpdateView();                   // src/Main.cpp:120
        put(cam.position); put(cam.forward); put(cam.right); put(cam.up);
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) std::printf("%08x ", bits(cam.inverseView[c][r]));
        std::printf("%d\n", cam.moving ? 1 : 0);
    }
    return 0;
}
