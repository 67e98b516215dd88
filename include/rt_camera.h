/* rt_camera.h -- host-side producers of the camera and portal uniforms of the ray-tracing pass (C ABI, no GPU).
 *
 * These functions restate, in binary32 and in the operation order of the glm templates the reference instantiates
 * (glm is header-only and not pinned by the reference's build; the order below is that of glm 0.9.9.8's scalar code:
 * lookAtRH, perspectiveRH_NO, mat4*mat4, mat4*vec4, compute_inverse<4,4>, rotate, normalize = v * (1/sqrt(dot)),
 * dot summed x+y+z), what the reference computes on the CPU before every draw:
 *
 *   Camera                 app/core/src/Camera.cpp:5-47, app/core/include/Camera.hpp:11-17,46-50
 *   Portal                 app/core/src/Portal.cpp:15-58
 *   CameraManipulator      app/core/src/CameraManipulator.cpp:11-25 (SetCamera), :27-124 (Update), :126-210 (events)
 *   the uniform upload     app/core/src/App.cpp:120-121,129-138
 *
 * sin/cos/tan/acos/atan2 are the host C library's float functions, as in the reference.  No reference build of glm
 * is available to compare with, so bit parity of these functions is NOT pinned by a fixture: tests hold them to the
 * double-precision formulas and to their algebraic properties.
 *
 * All matrices are float[16] in glm::value_ptr order (column-major: element [col][row] at 4*col + row).
 */
#ifndef RT_CAMERA_H
#define RT_CAMERA_H

#include "rt_hotpath.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Camera.hpp:31-56.  Plain data: copy it freely. */
typedef struct rt_camera {
    float eye[3], world_up[3], at[3];
    float view[16];                 /* m_viewMatrix */
    float z_near, z_far, angle, aspect;
    float proj[16];                 /* m_projMatrix */
} rt_camera;

/* Camera::Camera() (Camera.cpp:5-8): view from (1,1,1) to the origin, 90 degrees, aspect 1, planes 0.01 / 1000. */
void rt_camera_init(rt_camera *camera);
/* Camera::SetView (Camera.cpp:12-18). */
void rt_camera_set_view(rt_camera *camera, const float eye[3], const float at[3], const float world_up[3]);
/* Camera::SetProj (Camera.cpp:20-27); SetAngle / SetAspect / SetZNear / SetZFar are this call with one value changed. */
void rt_camera_set_proj(rt_camera *camera, float angle, float aspect, float z_near, float z_far);
/* Camera::GetViewProj (Camera.hpp:17): proj * view. */
void rt_camera_view_proj(const rt_camera *camera, float out[16]);
/* glm::inverse(GetViewProj()) as uploaded at App.cpp:120. */
void rt_camera_inv_view_proj(const rt_camera *camera, float out[16]);

/* glm::normalize(vec3) = v * (1 / sqrt(dot(v, v))), as App's constructor applies it to the portal directions
 * (App.cpp:31-32). */
void rt_vec3_normalize(const float v[3], float out[3]);

/* Portal::GetDifferenceMatrixTo (Portal.cpp:15-26): rotation taking direction `from` to direction `to`, identity when
 * they are parallel.  Like the reference it does not clamp the cosine: directions whose dot product exceeds 1 by
 * rounding give NaN entries. */
void rt_portal_difference_matrix(const float from[3], const float to[3], float out[16]);
/* Portal::RayPortal (Portal.cpp:28-58): distance along the ray to the portal rectangle, or -1. */
float rt_portal_ray(const float portal_position[3], const float portal_direction[3], const float ray_position[3],
                    const float ray_direction[3], float closest_distance, float portal_width, float portal_height);

/* CameraManipulator.hpp:30-64.  Plain data. */
typedef struct rt_manipulator {
    float u, v, distance;
    float center[3];
    float speed;                                    /* 16 */
    float go_forward, go_right, go_up;
    int forward_pressed, backward_pressed, left_pressed, right_pressed, up_pressed, down_pressed;
    float prev_eye[3], prev_center[3], prev_world_up[3];
} rt_manipulator;

enum { RT_KEY_SHIFT = 1, RT_KEY_W, RT_KEY_S, RT_KEY_A, RT_KEY_D, RT_KEY_E, RT_KEY_Q };

/* Member initialisers of CameraManipulator.hpp:35-64. */
void rt_manipulator_init(rt_manipulator *m);
/* CameraManipulator::SetCamera (CameraManipulator.cpp:11-25): spherical coordinates of the current view. */
void rt_manipulator_set_camera(rt_manipulator *m, const rt_camera *camera);
/* CameraManipulator::Update (CameraManipulator.cpp:27-124): moves the camera by the pressed keys over delta_time
 * seconds, carries it through a portal its path crosses, and calls SetView when anything changed.  Returns 1 when the
 * view changed (the caller restarts the accumulation, App.cpp:90-92), else 0. */
int rt_manipulator_update(rt_manipulator *m, rt_camera *camera, float delta_time,
                          const float portal_position_1[3], const float portal_direction_1[3],
                          const float portal_position_2[3], const float portal_direction_2[3],
                          float portal_width, float portal_height);
/* KeyboardDown / KeyboardUp (CameraManipulator.cpp:126-191); key is one of RT_KEY_*, repeat as in SDL_KeyboardEvent. */
void rt_manipulator_key(rt_manipulator *m, int key, int down, int repeat);
/* MouseMove (CameraManipulator.cpp:193-205): relative motion in pixels with the left / right button held. */
void rt_manipulator_mouse_move(rt_manipulator *m, int left_button, int right_button, int xrel, int yrel);
/* MouseWheel (CameraManipulator.cpp:207-210). */
void rt_manipulator_mouse_wheel(rt_manipulator *m, int y);

/* The camera and portal part of the uniform upload (App.cpp:120-121,129-138): fills inv_view_proj_mat,
 * camera_position, z_near, z_far and the portal fields of `u`; time, blur_amount and max_recursion_limit are left. */
void rt_uniforms_set_camera(rt_uniforms *u, const rt_camera *camera);
void rt_uniforms_set_portals(rt_uniforms *u, const float portal_position_1[3], const float portal_direction_1[3],
                             const float portal_position_2[3], const float portal_direction_2[3],
                             float portal_width, float portal_height);

#ifdef __cplusplus
}
#endif
#endif
