// Small C++ caller of rt_host.hpp, used by the tests: builds the octree of a triangle soup read from a file with the
// reference-named classes and optionally renders frames into device memory like the reference's App would.
//   host_demo <mesh.bin> <out.bin> [<uniforms.bin> <width> <height> <n_frames> [<overlay.bin>]]
// mesh.bin: u32 n_vert, u32 n_face, f32 pos[n_vert][3], u32 face[n_face][3]
// out.bin:  u64 n_vert, n_ref, n_words, f32 min[3], max[3], then vertices, normals, tri refs, node words;
//           with rendering: f32 rgba[height][width][4] and f32 depth[height][width] after n_frames accumulated draws
// uniforms.bin: n_frames x 70 words (rt_uniforms); the literal word `walk` instead of a file name makes the demo play the
//           reference's interactive loop itself: default view and portals (App.cpp:31-34,58-61), the W key held for all
//           but the last two frames at 1/60 s per frame (CameraManipulator::Update), accumulation restarted whenever the
//           view changed (App.cpp:90-92); the uniforms it used are appended to out.bin (n_frames x 70 words)
// overlay.bin (walk mode only): u32 n_vert, n_idx, tex_w, tex_h, f32 vertex[n_vert][8], u32 index[n_idx], u8 rgba[tex_h][tex_w][4]:
//           every frame then ends with the reference's raster pass (rt::RasterPass: nine draws, App.cpp:142-166) and out.bin
//           also receives the u32 depth24[height][width] plane
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <hip/hip_runtime_api.h>
#include "rt_host.hpp"

static void put(FILE *f, const void *p, size_t n) { if (n && fwrite(p, 1, n, f) != n) { perror("write"); exit(2); } }

int main(int argc, char **argv) {
    if (argc != 3 && argc != 7 && argc != 8) { fprintf(stderr, "usage: host_demo mesh.bin out.bin [uniforms.bin w h n [overlay.bin]]\n"); return 2; }
    FILE *in = fopen(argv[1], "rb");
    if (!in) { perror(argv[1]); return 2; }
    uint32_t nv = 0, nf = 0;
    if (fread(&nv, 4, 1, in) != 1 || fread(&nf, 4, 1, in) != 1) return 2;
    std::vector<rt::vec3> pos(nv);
    std::vector<uint32_t> faces(3 * (size_t)nf);
    if (fread(pos.data(), sizeof(rt::vec3), nv, in) != nv || fread(faces.data(), 12, nf, in) != nf) return 2;
    fclose(in);
    try {
        rt::Mesh mesh(pos, {}, faces, 1, rt::mat4::scale(2.0f));
        // App.cpp:22; with rendering (a GPU is there) the octree is built on the device, which gives the same arrays
        rt::Octree octree(std::vector<rt::Mesh>{mesh}, 18, 10, 6, 6, argc >= 7 ? 0 : -1);
        FILE *out = fopen(argv[2], "wb");
        if (!out) { perror(argv[2]); return 2; }
        const uint64_t sizes[3] = {octree.m_vertecies.size(), octree.m_compressed_triangles.size(), octree.m_compressed_node_buffer.size()};
        const rt::vec3 lo = octree.GetMinBounds(), hi = octree.GetMaxBounds();
        put(out, sizes, sizeof sizes); put(out, &lo, 12); put(out, &hi, 12);
        put(out, octree.m_vertecies.data(), sizes[0] * 16); put(out, octree.m_normals.data(), sizes[0] * 16);
        put(out, octree.m_compressed_triangles.data(), sizes[1] * 16); put(out, octree.m_compressed_node_buffer.data(), sizes[2] * 4);
        if (argc >= 7) {
            const uint32_t w = (uint32_t)atoi(argv[4]), h = (uint32_t)atoi(argv[5]), n = (uint32_t)atoi(argv[6]);
            const bool walk = std::string(argv[3]) == "walk";
            std::vector<rt_uniforms> u(n);
            if (!walk) {
                FILE *uf = fopen(argv[3], "rb");
                if (!uf || fread(u.data(), sizeof(rt_uniforms), n, uf) != n) { fprintf(stderr, "cannot read uniforms\n"); return 2; }
                fclose(uf);
            }
            float *color = nullptr, *depth = nullptr;
            if (hipMalloc((void **)&color, (size_t)w * h * 16) != hipSuccess || hipMalloc((void **)&depth, (size_t)w * h * 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 3; }
            (void)hipMemset(color, 0, (size_t)w * h * 16);
            rt::RayTracePass pass(octree);
            std::unique_ptr<rt::RasterPass> raster;
            uint32_t *depth24 = nullptr;
            if (argc == 8) {
                if (!walk) { fprintf(stderr, "the overlay needs the walk mode (it takes viewProj from the camera)\n"); return 2; }
                FILE *of = fopen(argv[7], "rb");
                uint32_t hd[4];
                if (!of || fread(hd, 4, 4, of) != 4) { fprintf(stderr, "cannot read the overlay file\n"); return 2; }
                std::vector<float> ov((size_t)hd[0] * 8); std::vector<uint32_t> oi(hd[1]); std::vector<uint8_t> ot((size_t)hd[2] * hd[3] * 4);
                if (fread(ov.data(), 4, ov.size(), of) != ov.size() || fread(oi.data(), 4, oi.size(), of) != oi.size() || fread(ot.data(), 1, ot.size(), of) != ot.size()) return 2;
                fclose(of);
                raster.reset(new rt::RasterPass(ov.data(), hd[0], oi.data(), oi.size(), ot.data(), hd[2], hd[3]));
                if (hipMalloc((void **)&depth24, (size_t)w * h * 4) != hipSuccess) return 3;
            }
            if (walk) {
                rt::Camera camera;
                camera.SetView({2, 2, 2}, {0, 0, 0}, {0, 1, 0});
                camera.SetAspect((float)w / (float)h);
                rt::CameraManipulator manipulator;
                manipulator.SetCamera(&camera);
                const rt::Portal portal_1({7.0f, 2.5f, 1.0f}, rt::normalize({2.0f, 0.0f, 0.5f})), portal_2({7.0f, 1.5f, 5.0f}, rt::normalize({0.0f, 0.0f, 1.0f}));
                manipulator.KeyboardDown(RT_KEY_W);
                for (uint32_t k = 0; k < n; ++k) {
                    if (k + 2 == n) manipulator.KeyboardUp(RT_KEY_W);
                    if (manipulator.Update(1.0f / 60.0f, portal_1, portal_2, 2.0f, 3.0f)) pass.ResetAccumulation();
                    u[k] = rt::FrameUniforms(camera, portal_1, portal_2, 2.0f, 3.0f, (float)k / 60.0f);
                    pass.Render(u[k], w, h, color, depth, nullptr, true);
                    if (raster) raster->Render(camera.GetViewProj(), w, h, color, depth, depth24);      // App.cpp:142-166
                }
            } else {
                pass.Render(u[0], w, h, color, depth, nullptr, true);                       // one draw ...
                if (n > 1) pass.RenderFrames(std::vector<rt_uniforms>(u.begin() + 1, u.end()), w, h, color, depth, nullptr, true);   // ... then the rest as a sequence
            }
            std::vector<float> hc((size_t)w * h * 4), hd((size_t)w * h);
            if (hipMemcpy(hc.data(), color, hc.size() * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(hd.data(), depth, hd.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 3;
            put(out, hc.data(), hc.size() * 4); put(out, hd.data(), hd.size() * 4);
            if (walk) put(out, u.data(), u.size() * sizeof(rt_uniforms));
            if (raster) {
                std::vector<uint32_t> hz((size_t)w * h);
                if (hipMemcpy(hz.data(), depth24, hz.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 3;
                put(out, hz.data(), hz.size() * 4);
                (void)hipFree(depth24);
            }
            (void)hipFree(color); (void)hipFree(depth);
        }
        fclose(out);
    } catch (const rt::Error &e) {
        fprintf(stderr, "rt error %d: %s\n", e.code(), e.what());
        return 1;
    }
    return 0;
}
