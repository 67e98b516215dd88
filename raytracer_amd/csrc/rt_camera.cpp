// Host-side camera, portal and camera-manipulator arithmetic of include/rt_camera.h: binary32, one rounding per
// written operation (this file is built with -ffp-contract=off), in the order of glm's scalar templates.
// Each function names the reference lines it follows (paths relative to /root/reference/app/core).
#include "../../include/rt_camera.h"

#include <cmath>
#include <cstring>

namespace {

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
struct M4 { V4 c[4]; };   // columns

inline V3 v3(const float *p) { return {p[0], p[1], p[2]}; }
inline void put(float *p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline bool differs(V3 a, V3 b) { return a.x != b.x || a.y != b.y || a.z != b.z; }
inline V4 operator+(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator-(V4 a, V4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline V4 operator*(V4 a, V4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline V4 operator*(V4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

// glm::dot(vec3): the three products, then x + y + z from the left.
inline float dot(V3 a, V3 b) { const float px = a.x * b.x, py = a.y * b.y, pz = a.z * b.z; return px + py + pz; }
// glm::cross.
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
// glm::normalize: v * inversesqrt(dot(v, v)), inversesqrt(x) = 1 / sqrt(x).
inline V3 normalize(V3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

M4 identity() { return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}}; }
M4 load(const float *p) { M4 m; std::memcpy(&m, p, sizeof m); return m; }
void store(float *p, const M4 &m) { std::memcpy(p, &m, sizeof m); }

// glm::lookAtRH (Camera.cpp:17).
M4 look_at(V3 eye, V3 center, V3 up) {
    const V3 f = normalize(center - eye);
    const V3 s = normalize(cross(f, up));
    const V3 u = cross(s, f);
    M4 r = identity();
    r.c[0].x = s.x; r.c[1].x = s.y; r.c[2].x = s.z;
    r.c[0].y = u.x; r.c[1].y = u.y; r.c[2].y = u.z;
    r.c[0].z = -f.x; r.c[1].z = -f.y; r.c[2].z = -f.z;
    r.c[3].x = -dot(s, eye);
    r.c[3].y = -dot(u, eye);
    r.c[3].z = dot(f, eye);
    return r;
}

// glm::perspectiveRH_NO (Camera.cpp:7,26).
M4 perspective(float fovy, float aspect, float z_near, float z_far) {
    const float t = std::tan(fovy / 2.0f);
    M4 r{};
    r.c[0].x = 1.0f / (aspect * t);
    r.c[1].y = 1.0f / t;
    r.c[2].z = -(z_far + z_near) / (z_far - z_near);
    r.c[2].w = -1.0f;
    r.c[3].z = -(2.0f * z_far * z_near) / (z_far - z_near);
    return r;
}

// glm mat4 * mat4: each result column is A0*b0 + A1*b1 + A2*b2 + A3*b3, summed from the left (Camera.hpp:17).
M4 mul(const M4 &a, const M4 &b) {
    M4 r;
    for (int j = 0; j < 4; ++j)
        r.c[j] = ((a.c[0] * b.c[j].x + a.c[1] * b.c[j].y) + a.c[2] * b.c[j].z) + a.c[3] * b.c[j].w;
    return r;
}

// glm mat4 * vec4: (m0*x + m1*y) + (m2*z + m3*w) (CameraManipulator.cpp:64-72).
V4 mul(const M4 &m, V4 v) { return (m.c[0] * v.x + m.c[1] * v.y) + (m.c[2] * v.z + m.c[3] * v.w); }

// glm::inverse(mat4) by cofactors (App.cpp:120).
M4 inverse(const M4 &m) {
    const V4 *c = m.c;
    const float c00 = c[2].z * c[3].w - c[3].z * c[2].w, c02 = c[1].z * c[3].w - c[3].z * c[1].w, c03 = c[1].z * c[2].w - c[2].z * c[1].w;
    const float c04 = c[2].y * c[3].w - c[3].y * c[2].w, c06 = c[1].y * c[3].w - c[3].y * c[1].w, c07 = c[1].y * c[2].w - c[2].y * c[1].w;
    const float c08 = c[2].y * c[3].z - c[3].y * c[2].z, c10 = c[1].y * c[3].z - c[3].y * c[1].z, c11 = c[1].y * c[2].z - c[2].y * c[1].z;
    const float c12 = c[2].x * c[3].w - c[3].x * c[2].w, c14 = c[1].x * c[3].w - c[3].x * c[1].w, c15 = c[1].x * c[2].w - c[2].x * c[1].w;
    const float c16 = c[2].x * c[3].z - c[3].x * c[2].z, c18 = c[1].x * c[3].z - c[3].x * c[1].z, c19 = c[1].x * c[2].z - c[2].x * c[1].z;
    const float c20 = c[2].x * c[3].y - c[3].x * c[2].y, c22 = c[1].x * c[3].y - c[3].x * c[1].y, c23 = c[1].x * c[2].y - c[2].x * c[1].y;
    const V4 f0{c00, c00, c02, c03}, f1{c04, c04, c06, c07}, f2{c08, c08, c10, c11};
    const V4 f3{c12, c12, c14, c15}, f4{c16, c16, c18, c19}, f5{c20, c20, c22, c23};
    const V4 v0{c[1].x, c[0].x, c[0].x, c[0].x}, v1{c[1].y, c[0].y, c[0].y, c[0].y};
    const V4 v2{c[1].z, c[0].z, c[0].z, c[0].z}, v3{c[1].w, c[0].w, c[0].w, c[0].w};
    const V4 i0 = (v1 * f0 - v2 * f1) + v3 * f2;
    const V4 i1 = (v0 * f0 - v2 * f3) + v3 * f4;
    const V4 i2 = (v0 * f1 - v1 * f3) + v3 * f5;
    const V4 i3 = (v0 * f2 - v1 * f4) + v2 * f5;
    const V4 sa{+1, -1, +1, -1}, sb{-1, +1, -1, +1};
    const M4 adj{{i0 * sa, i1 * sb, i2 * sa, i3 * sb}};
    const V4 row0{adj.c[0].x, adj.c[1].x, adj.c[2].x, adj.c[3].x};
    const V4 d = c[0] * row0;
    const float det = (d.x + d.y) + (d.z + d.w);
    const float inv_det = 1.0f / det;
    return {{adj.c[0] * inv_det, adj.c[1] * inv_det, adj.c[2] * inv_det, adj.c[3] * inv_det}};
}

// glm::rotate(angle, axis) = rotate(mat4(1), angle, axis) (Portal.cpp:22).
M4 rotate(float angle, V3 v) {
    const float c = std::cos(angle), s = std::sin(angle);
    const V3 a = normalize(v);
    const V3 t = (1.0f - c) * a;
    const float r00 = c + t.x * a.x, r01 = t.x * a.y + s * a.z, r02 = t.x * a.z - s * a.y;
    const float r10 = t.y * a.x - s * a.z, r11 = c + t.y * a.y, r12 = t.y * a.z + s * a.x;
    const float r20 = t.z * a.x + s * a.y, r21 = t.z * a.y - s * a.x, r22 = c + t.z * a.z;
    const M4 m = identity();
    M4 r;
    r.c[0] = (m.c[0] * r00 + m.c[1] * r01) + m.c[2] * r02;
    r.c[1] = (m.c[0] * r10 + m.c[1] * r11) + m.c[2] * r12;
    r.c[2] = (m.c[0] * r20 + m.c[1] * r21) + m.c[2] * r22;
    r.c[3] = m.c[3];
    return r;
}

// Portal.cpp:15-26.
M4 difference_matrix(V3 from, V3 to) {
    const float angle = std::acos(dot(from, to));
    const V3 axis = cross(from, to);
    if (static_cast<double>(length(axis)) > 0.0) return rotate(angle, normalize(axis));
    return identity();
}

// Portal.cpp:28-58; the comparisons against 0.0001 and 0.5 * size are made in double, as written there.
float ray_portal(V3 position, V3 direction, V3 ray_position, V3 ray_direction, float closest, float width, float height) {
    const float d = dot(direction, ray_direction);
    if (static_cast<double>(std::fabs(d)) <= 0.0001) return -1.0f;
    const float t = dot(position - ray_position, direction) / d;
    if (t < 0.0f || t > closest) return -1.0f;
    const V3 point = ray_position + t * ray_direction;
    V3 right = cross(direction, V3{0.0f, 1.0f, 0.0f});
    if (static_cast<double>(length(right)) <= 0.0001) return -1.0f;
    right = normalize(right);
    const V3 up = normalize(cross(right, direction));
    const V3 c = point - position;
    if (static_cast<double>(std::fabs(dot(right, c))) < 0.5 * static_cast<double>(width) &&
        static_cast<double>(std::fabs(dot(up, c))) < 0.5 * static_cast<double>(height))
        return t;
    return -1.0f;
}

void set_view(rt_camera *cam, V3 eye, V3 at, V3 up) {
    put(cam->eye, eye); put(cam->at, at); put(cam->world_up, up);
    store(cam->view, look_at(eye, at, up));
}

// The branch of CameraManipulator.cpp:63-82 / :84-103 for a crossing of portal `a` towards portal `b`.
void teleport(rt_manipulator *m, V3 pos_a, V3 dir_a, V3 pos_b, V3 dir_b, float t, float distance, V3 look, V3 &ray_position, V3 &ray_direction, V3 &eye) {
    const M4 diff = difference_matrix(dir_a, dir_b);
    const V3 rel = (ray_position + t * ray_direction) - pos_a;
    const V4 p = mul(diff, V4{rel.x, rel.y, rel.z, 1.0f});
    ray_position = V3{p.x, p.y, p.z} + pos_b;
    const V4 d = mul(diff, V4{ray_direction.x, ray_direction.y, ray_direction.z, 0.0f});
    ray_direction = normalize(V3{d.x, d.y, d.z});
    ray_position = ray_position + ((distance - t) + 0.001f) * ray_direction;
    const V4 l = mul(diff, V4{look.x, look.y, look.z, 0.0f});
    const V3 new_look = normalize(V3{l.x, l.y, l.z});
    eye = ray_position;
    put(m->center, eye + m->distance * new_look);
    m->u = std::atan2(new_look.z, new_look.x);
    m->v = std::acos(new_look.y);
}

}  // namespace

extern "C" {

void rt_camera_init(rt_camera *cam) {
    if (!cam) return;
    std::memset(cam, 0, sizeof *cam);
    cam->z_near = 0.01f; cam->z_far = 1000.0f;
    cam->angle = 90.0f * 0.01745329251994329576923690768489f;   // glm::radians(90.0f)
    cam->aspect = 1.0f;
    set_view(cam, V3{1, 1, 1}, V3{0, 0, 0}, V3{0, 1, 0});
    store(cam->proj, perspective(cam->angle, cam->aspect, cam->z_near, cam->z_far));
}

void rt_camera_set_view(rt_camera *cam, const float eye[3], const float at[3], const float world_up[3]) {
    if (!cam || !eye || !at || !world_up) return;
    set_view(cam, v3(eye), v3(at), v3(world_up));
}

void rt_camera_set_proj(rt_camera *cam, float angle, float aspect, float z_near, float z_far) {
    if (!cam) return;
    cam->angle = angle; cam->aspect = aspect; cam->z_near = z_near; cam->z_far = z_far;
    store(cam->proj, perspective(angle, aspect, z_near, z_far));
}

void rt_camera_view_proj(const rt_camera *cam, float out[16]) {
    if (!cam || !out) return;
    store(out, mul(load(cam->proj), load(cam->view)));
}

void rt_camera_inv_view_proj(const rt_camera *cam, float out[16]) {
    if (!cam || !out) return;
    store(out, inverse(mul(load(cam->proj), load(cam->view))));
}

void rt_vec3_normalize(const float v[3], float out[3]) {
    if (!v || !out) return;
    put(out, normalize(v3(v)));
}

void rt_portal_difference_matrix(const float from[3], const float to[3], float out[16]) {
    if (!from || !to || !out) return;
    store(out, difference_matrix(v3(from), v3(to)));
}

float rt_portal_ray(const float portal_position[3], const float portal_direction[3], const float ray_position[3],
                    const float ray_direction[3], float closest_distance, float portal_width, float portal_height) {
    if (!portal_position || !portal_direction || !ray_position || !ray_direction) return -1.0f;
    return ray_portal(v3(portal_position), v3(portal_direction), v3(ray_position), v3(ray_direction), closest_distance, portal_width, portal_height);
}

void rt_manipulator_init(rt_manipulator *m) {
    if (!m) return;
    std::memset(m, 0, sizeof *m);
    m->speed = 16.0f;
}

void rt_manipulator_set_camera(rt_manipulator *m, const rt_camera *cam) {
    if (!m || !cam) return;
    const V3 center = v3(cam->at);
    put(m->center, center);
    const V3 to_aim = center - v3(cam->eye);
    m->distance = length(to_aim);
    m->u = std::atan2(to_aim.z, to_aim.x);
    m->v = std::acos(to_aim.y / m->distance);
}

int rt_manipulator_update(rt_manipulator *m, rt_camera *cam, float delta_time,
                          const float portal_position_1[3], const float portal_direction_1[3],
                          const float portal_position_2[3], const float portal_direction_2[3],
                          float portal_width, float portal_height) {
    if (!m || !cam || !portal_position_1 || !portal_direction_1 || !portal_position_2 || !portal_direction_2) return 0;
    const V3 pos1 = v3(portal_position_1), dir1 = v3(portal_direction_1), pos2 = v3(portal_position_2), dir2 = v3(portal_direction_2);
    const float sin_v = std::sin(m->v);
    const V3 look{std::cos(m->u) * sin_v, std::cos(m->v), std::sin(m->u) * sin_v};
    V3 center = v3(m->center);
    V3 eye = center - m->distance * look;
    const V3 up = v3(cam->world_up);
    const V3 right = normalize(cross(look, up));
    const V3 forward = cross(up, right);
    const V3 delta = (((m->go_forward * forward + m->go_right * right) + m->go_up * up) * m->speed) * delta_time;

    const V3 prev_eye = v3(m->prev_eye);
    V3 ray_position = prev_eye;
    V3 ray_direction = (eye + delta) - prev_eye;
    const float distance = length(ray_direction);
    bool teleported = false;
    if (distance != 0.0f) {
        ray_direction = normalize(ray_direction);
        const float t1 = ray_portal(pos1, dir1, ray_position, ray_direction, distance, portal_width, portal_height);
        const float t2 = ray_portal(pos2, dir2, ray_position, ray_direction, distance, portal_width, portal_height);
        if (t1 != -1.0f && (t1 < t2 || t2 == -1.0f)) {
            teleport(m, pos1, dir1, pos2, dir2, t1, distance, look, ray_position, ray_direction, eye);
            teleported = true;
        } else if (t2 != -1.0f && (t2 < t1 || t1 == -1.0f)) {
            teleport(m, pos2, dir2, pos1, dir1, t2, distance, look, ray_position, ray_direction, eye);
            teleported = true;
        }
        center = v3(m->center);
    }
    if (!teleported) {
        eye = eye + delta;
        center = center + delta;
        put(m->center, center);
    }
    if (differs(eye, prev_eye) || differs(center, v3(m->prev_center)) || differs(up, v3(m->prev_world_up))) {
        put(m->prev_eye, eye); put(m->prev_center, center); put(m->prev_world_up, up);
        set_view(cam, eye, center, up);
        return 1;
    }
    return 0;
}

void rt_manipulator_key(rt_manipulator *m, int key, int down, int repeat) {
    if (!m) return;
    if (down) {
        switch (key) {
            case RT_KEY_SHIFT: if (repeat == 0) m->speed /= 4.0f; break;
            case RT_KEY_W: m->forward_pressed = 1; m->go_forward = 1; break;
            case RT_KEY_S: m->backward_pressed = 1; m->go_forward = -1; break;
            case RT_KEY_A: m->left_pressed = 1; m->go_right = -1; break;
            case RT_KEY_D: m->right_pressed = 1; m->go_right = 1; break;
            case RT_KEY_E: m->up_pressed = 1; m->go_up = 1; break;
            case RT_KEY_Q: m->down_pressed = 1; m->go_up = -1; break;
            default: break;
        }
    } else {
        switch (key) {
            case RT_KEY_SHIFT: m->speed *= 4.0f; break;
            case RT_KEY_W: m->forward_pressed = 0; m->go_forward = m->backward_pressed ? -1.0f : 0.0f; break;
            case RT_KEY_S: m->backward_pressed = 0; m->go_forward = m->forward_pressed ? 1.0f : 0.0f; break;
            case RT_KEY_A: m->left_pressed = 0; m->go_right = m->right_pressed ? 1.0f : 0.0f; break;
            case RT_KEY_D: m->right_pressed = 0; m->go_right = m->left_pressed ? -1.0f : 0.0f; break;
            case RT_KEY_Q: m->down_pressed = 0; m->go_up = m->up_pressed ? 1.0f : 0.0f; break;
            case RT_KEY_E: m->up_pressed = 0; m->go_up = m->down_pressed ? -1.0f : 0.0f; break;
            default: break;
        }
    }
}

void rt_manipulator_mouse_move(rt_manipulator *m, int left_button, int right_button, int xrel, int yrel) {
    if (!m) return;
    if (left_button) {
        const float du = static_cast<float>(xrel) / 100.0f, dv = static_cast<float>(yrel) / 100.0f;
        m->u += du;
        const float v = m->v + dv;
        const float lo = v < 0.1f ? 0.1f : v;       // glm::clamp = min(max(x, lo), hi) with glm's comparisons
        m->v = 3.1f < lo ? 3.1f : lo;
    }
    if (right_button) m->distance += static_cast<float>(yrel) / 100.0f;
}

void rt_manipulator_mouse_wheel(rt_manipulator *m, int y) {
    if (!m) return;
    m->distance += static_cast<float>(y) * m->speed / -100.0f;
}

void rt_uniforms_set_camera(rt_uniforms *u, const rt_camera *cam) {
    if (!u || !cam) return;
    rt_camera_inv_view_proj(cam, u->inv_view_proj_mat);
    std::memcpy(u->camera_position, cam->eye, sizeof u->camera_position);
    u->z_near = cam->z_near; u->z_far = cam->z_far;
}

void rt_uniforms_set_portals(rt_uniforms *u, const float portal_position_1[3], const float portal_direction_1[3],
                             const float portal_position_2[3], const float portal_direction_2[3],
                             float portal_width, float portal_height) {
    if (!u || !portal_position_1 || !portal_direction_1 || !portal_position_2 || !portal_direction_2) return;
    std::memcpy(u->portal_position_1, portal_position_1, 12); std::memcpy(u->portal_direction_1, portal_direction_1, 12);
    std::memcpy(u->portal_position_2, portal_position_2, 12); std::memcpy(u->portal_direction_2, portal_direction_2, 12);
    u->portal_width = portal_width; u->portal_height = portal_height;
    rt_portal_difference_matrix(portal_direction_1, portal_direction_2, u->portal_1_to_2);
    rt_portal_difference_matrix(portal_direction_2, portal_direction_1, u->portal_2_to_1);
}

}  // extern "C"
