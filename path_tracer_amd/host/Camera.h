// Camera.h — same surface as the reference's Camera (src/Camera.h:8-29): public fields speed / moving /
// pitch / yaw / position / forward / up / right / inverseView, UpdateView(), ProcessInput().
// Headless: ProcessInput takes a key mask instead of a GLFWwindow* (the reference polls WASDQE there,
// src/Camera.cpp:28-59).  The state becomes kernel arguments through Renderer::SetUniformCamera.
#pragma once
#include "Math.h"

enum CameraKey : unsigned { KEY_W = 1, KEY_S = 2, KEY_A = 4, KEY_D = 8, KEY_E = 16, KEY_Q = 32 };

struct Camera {
    float speed = 2.0f;
    bool moving = false;

    float pitch = 0.0f;
    float yaw = 0.0f;

    float mouseLastX = 0.0f;
    float mouseLastY = 0.0f;

    ptm::vec3 position = ptm::vec3(0.0, 0.0, 0.0);
    ptm::vec3 forward = ptm::vec3(0.0, 0.0, 0.0);
    ptm::vec3 up = ptm::vec3(0.0, 1.0, 0.0);
    ptm::vec3 right = ptm::vec3(1.0, 0.0, 0.0);
    ptm::mat4 inverseView;

    Camera() = default;
    explicit Camera(ptm::vec3 pos) : position(pos) {}
    void UpdateView();                                   // src/Camera.cpp:13-26
    void ProcessInput(unsigned keys, float dt);          // src/Camera.cpp:28-59
    void MouseLook(float xOffset, float yOffset);        // Renderer::MouseCallback's yaw/pitch update, src/Renderer.cpp:152-161
};
