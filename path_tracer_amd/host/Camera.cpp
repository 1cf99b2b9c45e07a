#include "Camera.h"

using namespace ptm;

// yaw/pitch -> forward/right/up -> inverseView = inverse(lookAt(pos, pos + forward, up))
void Camera::UpdateView()
{
    const float cy = std::cos(radians(yaw)), sy = std::sin(radians(yaw));
    const float cp = std::cos(radians(pitch)), sp = std::sin(radians(pitch));
    forward = normalize(vec3(cy * cp, sp, sy * cp));
    right = normalize(cross(vec3(0.0, 1.0, 0.0), forward));
    up = cross(forward, right);
    inverseView = inverse(lookAt(position, position + forward, up));
}

// The shader shoots rays along view-space +Z, which lookAtRH maps to world -forward; hence W moves
// along -forward exactly as the reference does.
void Camera::ProcessInput(unsigned keys, float dt)
{
    // `position -= forward * speed * dt` (src/Camera.cpp:33): (forward * speed) * dt, two roundings per component
    if (keys & KEY_W) { moving = true; position -= forward * speed * dt; }
    else if (keys & KEY_S) { moving = true; position += forward * speed * dt; }
    if (keys & KEY_A) { moving = true; position -= right * speed * dt; }
    else if (keys & KEY_D) { moving = true; position += right * speed * dt; }
    if (keys & KEY_E) { moving = true; position += up * speed * dt; }
    else if (keys & KEY_Q) { moving = true; position -= up * speed * dt; }
}

void Camera::MouseLook(float xOffset, float yOffset)
{
    if (std::fabs(xOffset) > 0 || std::fabs(yOffset) > 0) moving = true;
    const float sensitivity = 0.1f;
    xOffset *= sensitivity;
    yOffset *= sensitivity;
    yaw += xOffset;
    pitch -= yOffset;
    if (pitch > 89.0f) pitch = 89.0f;
    if (pitch < -89.0f) pitch = -89.0f;
    forward = normalize(vec3(std::cos(radians(yaw)) * std::cos(radians(pitch)), std::sin(radians(pitch)),
                             std::sin(radians(yaw)) * std::cos(radians(pitch))));
}
