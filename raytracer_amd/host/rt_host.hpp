// C++ host side of the MI355X ray-tracing pass: the scene classes and the render pass of the reference application
// with their member and method names, implemented on the C ABI of include/rt_hotpath.h (header-only, no GL, no glm).
//
//   Mesh        mirrors Mesh        (app/core/include/Mesh.hpp:9-20,   app/core/src/Mesh.cpp:11-101)
//   Octree      mirrors Octree      (app/core/include/Octree.hpp:25-36, app/core/src/Octree.cpp:203-256)
//   RayTracePass replaces what App keeps for the ray-tracing draw: the four storage buffers (App.cpp:23-26), the
//               skybox texture (Skybox.cpp:46-57), the accumulation counter (App.cpp:29,90-92,104-105,209) and the
//               draw itself (App.cpp:107-140)
//   RasterPass  replaces what App keeps for the pass after it: the mesh object and texture (App.cpp:65-76), the light
//               (App.hpp:79-81) and the nine draws (App.cpp:142-166)
//   Camera, Portal, CameraManipulator mirror the classes of the same names (Camera.hpp:5-56, Portal.hpp,
//               CameraManipulator.hpp:12-65) on include/rt_camera.h: binary32 arithmetic in glm's operation order
//
// Errors are reported as rt::Error exceptions carrying the ABI's status code and message; the reference logs and
// exits (Mesh.cpp:20-26) or logs and continues (GL).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_camera.h"
#include "../../include/rt_overlay.h"
#include "../../include/rt_hotpath.h"

namespace rt {

struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct uvec4 { uint32_t x, y, z, w; };
struct mat4 {   // column-major, as glm::value_ptr yields it
    float m[16];
    static mat4 identity() { mat4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
    static mat4 scale(float s) { mat4 r = identity(); r.m[0] = r.m[5] = r.m[10] = s; return r; }
};

class Error : public std::runtime_error {
public:
    Error(int code, const char *msg) : std::runtime_error(msg ? msg : "rt error"), m_code(code) {}
    int code() const { return m_code; }
private:
    int m_code;
};
inline void check(int rc) { if (rc != RT_OK) throw Error(rc, rt_last_error()); }

class Mesh {
public:
    Mesh(const std::string &filename, size_t material_id, mat4 transform = mat4::identity()) {
        rt_mesh m{};
        check(rt_mesh_load_obj(filename.c_str(), (uint32_t)material_id, transform.m, &m));
        adopt(m);
    }
    // From arrays already in memory: positions [n], optional normals [n] (empty: accumulated face normals), faces.
    Mesh(const std::vector<vec3> &positions, const std::vector<vec3> &file_normals, const std::vector<uint32_t> &faces, size_t material_id, mat4 transform = mat4::identity()) {
        rt_mesh m{};
        check(rt_mesh_from_arrays(&positions[0].x, file_normals.empty() ? nullptr : &file_normals[0].x, positions.size(), faces.data(), faces.size() / 3,
                                  (uint32_t)material_id, transform.m, &m));
        adopt(m);
    }
    std::vector<vec4> m_vertecies;
    std::vector<vec4> m_normals;
    std::vector<uvec4> m_triangles;
private:
    void adopt(rt_mesh &m) {
        m_vertecies.resize(m.n_vert); m_normals.resize(m.n_vert); m_triangles.resize(m.n_tri);
        if (m.n_vert) { std::memcpy(m_vertecies.data(), m.vertices, m.n_vert * sizeof(vec4)); std::memcpy(m_normals.data(), m.normals, m.n_vert * sizeof(vec4)); }
        if (m.n_tri) std::memcpy(m_triangles.data(), m.triangles, m.n_tri * sizeof(uvec4));
        rt_mesh_free(&m);
    }
};

class Octree {
public:
    // build_device < 0: the host builder; otherwise the construction runs on that GPU (identical arrays)
    Octree(const std::vector<Mesh> &meshes, size_t depth_limit, size_t max_triangles_per_node, size_t max_triangles_per_leaf, size_t keep_triangles_after_this_many_overlaps,
           int build_device = -1) {
        std::vector<rt_mesh> in(meshes.size());
        for (size_t i = 0; i < meshes.size(); i++) {
            const Mesh &m = meshes[i];
            in[i].vertices = const_cast<float *>(m.m_vertecies.empty() ? nullptr : &m.m_vertecies[0].x);
            in[i].normals = const_cast<float *>(m.m_normals.empty() ? nullptr : &m.m_normals[0].x);
            in[i].triangles = const_cast<uint32_t *>(m.m_triangles.empty() ? nullptr : &m.m_triangles[0].x);
            in[i].n_vert = m.m_vertecies.size(); in[i].n_tri = m.m_triangles.size();
        }
        rt_octree o{};
        if (build_device < 0) check(rt_octree_build(in.data(), in.size(), depth_limit, max_triangles_per_node, max_triangles_per_leaf, keep_triangles_after_this_many_overlaps, &o));
        else check(rt_octree_build_device(in.data(), in.size(), depth_limit, max_triangles_per_node, max_triangles_per_leaf, keep_triangles_after_this_many_overlaps, build_device, &o));
        m_vertecies.resize(o.n_vert); m_normals.resize(o.n_vert); m_compressed_triangles.resize(o.n_ref); m_compressed_node_buffer.resize(o.n_words);
        if (o.n_vert) { std::memcpy(m_vertecies.data(), o.vertices, o.n_vert * sizeof(vec4)); std::memcpy(m_normals.data(), o.normals, o.n_vert * sizeof(vec4)); }
        if (o.n_ref) std::memcpy(m_compressed_triangles.data(), o.tri_refs, o.n_ref * sizeof(uvec4));
        if (o.n_words) std::memcpy(m_compressed_node_buffer.data(), o.nodes, o.n_words * sizeof(uint32_t));
        m_min = {o.bounds_min[0], o.bounds_min[1], o.bounds_min[2]}; m_max = {o.bounds_max[0], o.bounds_max[1], o.bounds_max[2]};
        m_max_depth = o.max_depth;
        rt_octree_free(&o);
    }
    vec3 GetMinBounds() const { return m_min; }
    vec3 GetMaxBounds() const { return m_max; }
    size_t GetMaxDepth() const { return m_max_depth; }

    std::vector<vec4> m_vertecies;
    std::vector<vec4> m_normals;
    std::vector<uint32_t> m_compressed_node_buffer;
    std::vector<uvec4> m_compressed_triangles;
private:
    vec3 m_min{}, m_max{};
    size_t m_max_depth = 0;
};

// The ray-tracing draw with its state.  The colour target is a caller-owned linear RGBA32F image in device memory
// ([height][width][4] floats, row 0 at the bottom) that persists across frames, like the reference's framebuffer
// attachment (Framebuffer.cpp:6); depth is optional.
class RayTracePass {
public:
    explicit RayTracePass(const Octree &octree) : m_min(octree.GetMinBounds()), m_max(octree.GetMaxBounds()) {
        const vec3 lo = octree.GetMinBounds(), hi = octree.GetMaxBounds();
        const float bmin[3] = {lo.x, lo.y, lo.z}, bmax[3] = {hi.x, hi.y, hi.z};
        check(rt_scene_create(octree.m_vertecies.empty() ? nullptr : &octree.m_vertecies[0].x, octree.m_normals.empty() ? nullptr : &octree.m_normals[0].x, octree.m_vertecies.size(),
                              octree.m_compressed_triangles.empty() ? nullptr : &octree.m_compressed_triangles[0].x, octree.m_compressed_triangles.size(),
                              octree.m_compressed_node_buffer.data(), octree.m_compressed_node_buffer.size(), bmin, bmax, &m_scene));
    }
    // Octree::Octree and the buffer uploads (App.cpp:22-26) in one step that stays on GPU `device`: no host copy of the octree exists.
    RayTracePass(const std::vector<Mesh> &meshes, size_t depth_limit, size_t max_triangles_per_node, size_t max_triangles_per_leaf, size_t keep_triangles_after_this_many_overlaps,
                 int device) {
        std::vector<rt_mesh> in(meshes.size());
        for (size_t i = 0; i < meshes.size(); i++) {
            const Mesh &m = meshes[i];
            in[i].vertices = const_cast<float *>(m.m_vertecies.empty() ? nullptr : &m.m_vertecies[0].x);
            in[i].normals = const_cast<float *>(m.m_normals.empty() ? nullptr : &m.m_normals[0].x);
            in[i].triangles = const_cast<uint32_t *>(m.m_triangles.empty() ? nullptr : &m.m_triangles[0].x);
            in[i].n_vert = m.m_vertecies.size(); in[i].n_tri = m.m_triangles.size();
        }
        float lo[3], hi[3];
        check(rt_scene_create_from_meshes(in.data(), in.size(), depth_limit, max_triangles_per_node, max_triangles_per_leaf, keep_triangles_after_this_many_overlaps, device, lo, hi, &m_scene));
        m_min = {lo[0], lo[1], lo[2]}; m_max = {hi[0], hi[1], hi[2]};
    }
    vec3 GetMinBounds() const { return m_min; }    // (of the octree, when the pass was made from meshes)
    vec3 GetMaxBounds() const { return m_max; }
    ~RayTracePass() { (void)rt_scene_destroy(m_scene); }
    RayTracePass(const RayTracePass &) = delete;
    RayTracePass &operator=(const RayTracePass &) = delete;

    // Six RGBA8 faces in the order +x,-x,+y,-y,+z,-z (Skybox.cpp:46-57).
    void SetSkybox(const uint8_t *const faces[6], int w, int h) { check(rt_skybox_set(m_scene, faces, w, h)); }
    // App.cpp:90-92,209: any camera change or resize restarts the accumulation.
    void ResetAccumulation() { m_still_frame_counter = 1; }
    uint32_t StillFrameCounter() const { return m_still_frame_counter; }
    // No counterpart in the reference: visit octree children nearest first (faster on deep scenes; outside bit
    // parity with the shader, whose fixed octant order is the default).
    void SetNearFirstTraversal(bool on) { check(rt_scene_set_traversal_order(m_scene, on ? RT_ORDER_NEAR_FIRST : RT_ORDER_REFERENCE)); }

    // App::Render's ray-tracing draw (App.cpp:97-140) into device memory; sync = false returns after the enqueue.
    void Render(const rt_uniforms &u, uint32_t width, uint32_t height, float *color_dev, float *depth_dev = nullptr, void *stream = nullptr, bool sync = false) {
        const rt_frame f = Frame(u, width, height, stream);
        const rt_target t = Target(color_dev, depth_dev, sync);
        check(rt_render(m_scene, &f, &t));
    }
    // n consecutive draws with an unchanged camera (uniforms differ in `time` only), issued as one sequence.
    void RenderFrames(const std::vector<rt_uniforms> &u, uint32_t width, uint32_t height, float *color_dev, float *depth_dev = nullptr, void *stream = nullptr, bool sync = false) {
        std::vector<rt_frame> f;
        f.reserve(u.size());
        for (const rt_uniforms &ui : u) f.push_back(Frame(ui, width, height, stream));
        const rt_target t = Target(color_dev, depth_dev, sync);
        check(rt_render_sequence(m_scene, f.data(), f.size(), &t));
    }
    rt_scene *Handle() { return m_scene; }
private:
    rt_frame Frame(const rt_uniforms &u, uint32_t width, uint32_t height, void *stream) {
        rt_frame f{};
        f.u = u; f.width = width; f.height = height;
        f.accum_index = m_still_frame_counter++;   // App.cpp:104-105
        f.blend = 1;                               // App.cpp:102
        f.band_rows = 16; f.band_parts = 1; f.band_part = 0;
        f.stream = stream;
        return f;
    }
    static rt_target Target(float *color_dev, float *depth_dev, bool sync) {
        rt_target t{};
        t.memory = RT_MEM_DEVICE; t.sync = sync ? 1 : 0; t.color = color_dev; t.depth = depth_dev;
        return t;
    }
    rt_scene *m_scene = nullptr;
    vec3 m_min{}, m_max{};
    uint32_t m_still_frame_counter = 1;            // App.cpp:29
};

// The raster pass after the ray-tracing draw (App.cpp:142-166) on the same device planes: the mesh (interleaved Vertex =
// position, normal, texcoord: GLUtils.hpp:29-34) and its RGBA8 texture (row 0 = bottom row, as TextureFromFile uploads it),
// the light, and one draw per instance matrix -- by default the reference's nine translations (App.cpp:157-159).
class RasterPass {
public:
    RasterPass(const float *vertices, size_t n_vert, const uint32_t *indices, size_t n_idx, const uint8_t *texture_rgba, uint32_t tex_w, uint32_t tex_h, int device = 0) {
        check(rt_overlay_mesh_create(vertices, n_vert, indices, n_idx, texture_rgba, tex_w, tex_h, device, &m_mesh));
        for (int i = -1; i <= 1; ++i)
            for (int j = -1; j <= 1; ++j) AddInstance({4.0f * (float)i, 4.0f + 4.0f * (float)j, 0.0f});
    }
    ~RasterPass() { (void)rt_overlay_mesh_destroy(m_mesh); }
    RasterPass(const RasterPass &) = delete;
    RasterPass &operator=(const RasterPass &) = delete;

    vec3 m_lightDir{-1.0f, -2.0f, -0.5f}, m_La{0.0f, 0.0f, 0.0f}, m_Ld{1.0f, 1.0f, 1.0f};   // App.hpp:79-81
    void ClearInstances() { m_world.clear(); m_world_it.clear(); }
    // glm::translate(t) and transpose(inverse(.)) of it: the inverse of a translation is the opposite translation, exactly
    void AddInstance(vec3 t) {
        mat4 w = mat4::identity(), it = mat4::identity();
        w.m[12] = t.x; w.m[13] = t.y; w.m[14] = t.z;
        it.m[3] = -t.x; it.m[7] = -t.y; it.m[11] = -t.z;
        m_world.push_back(w); m_world_it.push_back(it);
    }
    void AddInstance(const mat4 &world, const mat4 &world_inverse_transposed) { m_world.push_back(world); m_world_it.push_back(world_inverse_transposed); }

    // depth_dev: the plane RayTracePass::Render wrote (gl_FragDepth); depth24_dev: width*height uint32_t of device memory that
    // receives the attachment's 24-bit values and is then updated by the draws, like color_dev.
    void Render(const mat4 &view_proj, uint32_t width, uint32_t height, float *color_dev, const float *depth_dev, uint32_t *depth24_dev, void *stream = nullptr) {
        check(rt_depth_to_attachment(depth_dev, depth24_dev, (size_t)width * height, RT_MEM_DEVICE, stream));
        rt_overlay_pass p{};
        std::memcpy(p.view_proj, view_proj.m, sizeof p.view_proj);
        std::memcpy(p.light_dir, &m_lightDir.x, 12); std::memcpy(p.la, &m_La.x, 12); std::memcpy(p.ld, &m_Ld.x, 12);
        p.n_draws = (uint32_t)m_world.size();
        p.world = m_world.empty() ? nullptr : m_world[0].m; p.world_it = m_world_it.empty() ? nullptr : m_world_it[0].m;
        check(rt_overlay_draw(m_mesh, &p, color_dev, depth24_dev, width, height, RT_MEM_DEVICE, stream));
    }
private:
    rt_overlay_mesh *m_mesh = nullptr;
    std::vector<mat4> m_world, m_world_it;
};

// Camera.hpp:5-56.
class Camera {
public:
    Camera() { rt_camera_init(&m_c); }
    vec3 GetEye() const { return {m_c.eye[0], m_c.eye[1], m_c.eye[2]}; }
    vec3 GetAt() const { return {m_c.at[0], m_c.at[1], m_c.at[2]}; }
    vec3 GetWorldUp() const { return {m_c.world_up[0], m_c.world_up[1], m_c.world_up[2]}; }
    mat4 GetViewMatrix() const { mat4 r; std::memcpy(r.m, m_c.view, sizeof r.m); return r; }
    mat4 GetProj() const { mat4 r; std::memcpy(r.m, m_c.proj, sizeof r.m); return r; }
    mat4 GetViewProj() const { mat4 r; rt_camera_view_proj(&m_c, r.m); return r; }
    void SetView(vec3 eye, vec3 at, vec3 up) { rt_camera_set_view(&m_c, &eye.x, &at.x, &up.x); }
    void SetProj(float angle, float aspect, float zn, float zf) { rt_camera_set_proj(&m_c, angle, aspect, zn, zf); }
    float GetAngle() const { return m_c.angle; }
    void SetAngle(float a) { SetProj(a, m_c.aspect, m_c.z_near, m_c.z_far); }
    float GetAspect() const { return m_c.aspect; }
    void SetAspect(float a) { SetProj(m_c.angle, a, m_c.z_near, m_c.z_far); }
    float GetZNear() const { return m_c.z_near; }
    void SetZNear(float zn) { SetProj(m_c.angle, m_c.aspect, zn, m_c.z_far); }
    float GetZFar() const { return m_c.z_far; }
    void SetZFar(float zf) { SetProj(m_c.angle, m_c.aspect, m_c.z_near, zf); }
    rt_camera *Handle() { return &m_c; }
    const rt_camera *Handle() const { return &m_c; }
private:
    rt_camera m_c;
};

inline vec3 normalize(vec3 v) { vec3 r; rt_vec3_normalize(&v.x, &r.x); return r; }   // glm::normalize

// Portal.hpp / Portal.cpp:3-58.
class Portal {
public:
    Portal(vec3 position, vec3 direction) : m_position(position), m_direction(direction) {}
    vec3 GetPosition() const { return m_position; }
    vec3 GetDirection() const { return m_direction; }
    mat4 GetDifferenceMatrixTo(const Portal &destination_portal) const {
        mat4 r;
        rt_portal_difference_matrix(&m_direction.x, &destination_portal.m_direction.x, r.m);
        return r;
    }
    float RayPortal(vec3 ray_position, vec3 ray_direction, float closest_distance, float portal_width, float portal_height) const {
        return rt_portal_ray(&m_position.x, &m_direction.x, &ray_position.x, &ray_direction.x, closest_distance, portal_width, portal_height);
    }
private:
    vec3 m_position, m_direction;
};

// CameraManipulator.hpp:12-65; events arrive as plain numbers (key = RT_KEY_*).
class CameraManipulator {
public:
    CameraManipulator() { rt_manipulator_init(&m_m); }
    void SetCamera(Camera *camera) { m_pCamera = camera; if (camera) rt_manipulator_set_camera(&m_m, camera->Handle()); }
    bool Update(float delta_time, const Portal &portal_1, const Portal &portal_2, float portal_width, float portal_height) {
        if (!m_pCamera) return false;
        const vec3 p1 = portal_1.GetPosition(), d1 = portal_1.GetDirection(), p2 = portal_2.GetPosition(), d2 = portal_2.GetDirection();
        return rt_manipulator_update(&m_m, m_pCamera->Handle(), delta_time, &p1.x, &d1.x, &p2.x, &d2.x, portal_width, portal_height) != 0;
    }
    void SetSpeed(float speed) { m_m.speed = speed; }
    float GetSpeed() const { return m_m.speed; }
    void KeyboardDown(int key, int repeat = 0) { rt_manipulator_key(&m_m, key, 1, repeat); }
    void KeyboardUp(int key) { rt_manipulator_key(&m_m, key, 0, 0); }
    void MouseMove(int xrel, int yrel, bool left, bool right) { rt_manipulator_mouse_move(&m_m, left, right, xrel, yrel); }
    void MouseWheel(int y) { rt_manipulator_mouse_wheel(&m_m, y); }
private:
    Camera *m_pCamera = nullptr;
    rt_manipulator m_m;
};

// The uniforms of one draw as App::Render uploads them (App.cpp:120-138).
inline rt_uniforms FrameUniforms(const Camera &camera, const Portal &portal_1, const Portal &portal_2, float portal_width, float portal_height,
                                 float time, uint32_t max_recursion_limit = 5, float blur_amount = 0.00001f) {
    rt_uniforms u{};
    rt_uniforms_set_camera(&u, camera.Handle());
    const vec3 p1 = portal_1.GetPosition(), d1 = portal_1.GetDirection(), p2 = portal_2.GetPosition(), d2 = portal_2.GetDirection();
    rt_uniforms_set_portals(&u, &p1.x, &d1.x, &p2.x, &d2.x, portal_width, portal_height);
    u.time = time; u.max_recursion_limit = max_recursion_limit; u.blur_amount = blur_amount;
    return u;
}

// The depth-buffer side of the overlay pass on device planes (Framebuffer.cpp:10-12, App.cpp:54,142-145): the 24-bit
// values the reference's depth attachment holds for the plane RayTracePass::Render wrote, and the GL_LEQUAL merge of one
// layer of fragments from a rasteriser of the caller's choice.  Both are ordered on `stream`.
inline void DepthToAttachment(const float *depth_dev, uint32_t *depth24_dev, size_t n_pixels, void *stream = nullptr) {
    check(rt_depth_to_attachment(depth_dev, depth24_dev, n_pixels, RT_MEM_DEVICE, stream));
}
inline void OverlayComposite(float *color_dev, uint32_t *depth24_dev, const float *frag_color_dev, const uint32_t *frag_depth24_dev, const uint8_t *frag_mask_dev,
                             size_t n_pixels, void *stream = nullptr) {
    check(rt_overlay_composite(color_dev, depth24_dev, frag_color_dev, frag_depth24_dev, frag_mask_dev, n_pixels, RT_MEM_DEVICE, stream));
}

}  // namespace rt
