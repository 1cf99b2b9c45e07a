// glm_camera_ref — TEST INFRASTRUCTURE (oracle).  Computes cam.inverseView for a given
// (position, yaw, pitch) through the reference's own vendored GLM 0.9.8.5 headers, which are
// compiled where they lie (-I/root/reference/dependencies/glm, see oracle/Makefile); nothing from
// the reference is copied.  The call sequence is the one Camera::UpdateView makes
// (src/Camera.cpp:13-26): radians -> cos/sin -> normalize -> cross -> lookAt -> inverse.
// Output: one line per camera, 19 hex float32 words: position(3) then inverseView column-major(16).
// Used by oracle/gen_goldens.py to write tests/golden/camera.json.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <glm.hpp>
#include <gtc/matrix_transform.hpp>

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

int main(int argc, char **argv)
{
    for (int i = 1; i + 4 < argc; i += 5) {
        glm::vec3 position(std::strtof(argv[i], nullptr), std::strtof(argv[i + 1], nullptr), std::strtof(argv[i + 2], nullptr));
        float yaw = std::strtof(argv[i + 3], nullptr), pitch = std::strtof(argv[i + 4], nullptr);
        const float cy = cos(glm::radians(yaw)), sy = sin(glm::radians(yaw));
        const float cp = cos(glm::radians(pitch)), sp = sin(glm::radians(pitch));
        const glm::vec3 fwd = glm::normalize(glm::vec3(cy * cp, sp, sy * cp));
        const glm::vec3 rgt = glm::normalize(glm::cross(glm::vec3(0.0, 1.0, 0.0), fwd));
        const glm::vec3 upv = glm::cross(fwd, rgt);
        const glm::mat4 inverseView = glm::inverse(glm::lookAt(position, position + fwd, upv));
        std::printf("%08x %08x %08x", bits(position.x), bits(position.y), bits(position.z));
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) std::printf(" %08x", bits(inverseView[c][r]));
        std::printf("\n");
    }
    return 0;
}
