// Development aid: drives the host-side builders of include/rt_hotpath.h (rt_mesh_from_arrays, rt_octree_build: the threaded octree
// construction) on random triangle soups, for runs under the host sanitizers -- see tools/sanitize_host.sh.  Not part of the library.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../include/rt_hotpath.h"

extern "C" int rt_internal_set_error(int code, const char *msg) { std::fprintf(stderr, "error %d: %s\n", code, msg ? msg : ""); return code; }

int main(int argc, char **argv) {
    const int cases = argc > 1 ? std::atoi(argv[1]) : 40;
    std::mt19937 rng(argc > 2 ? (unsigned)std::atoi(argv[2]) : 1u);
    auto uni = [&](float a, float b) { return std::uniform_real_distribution<float>(a, b)(rng); };
    const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    size_t total_words = 0;
    for (int c = 0; c < cases; c++) {
        const int n_meshes = 1 + (int)(rng() % 3u);
        std::vector<rt_mesh> meshes((size_t)n_meshes);
        bool ok = true;
        for (int m = 0; m < n_meshes && ok; m++) {
            const int kind = (int)(rng() % 4u);
            const size_t n_tri = kind == 3 ? 1 + rng() % 4u : 50 + rng() % 3000u;
            std::vector<float> pos;
            std::vector<uint32_t> faces;
            for (size_t t = 0; t < n_tri; t++) {
                float cx = uni(-3, 3), cy = uni(0, 6), cz = uni(-3, 3), s = kind == 1 ? 0.5f : std::pow(10.0f, uni(-3.0f, -0.5f));
                if (kind == 1) { cx = std::round(cx * 2) / 2; cy = std::round(cy * 2) / 2; cz = std::round(cz * 2) / 2; }   // sheets on lattice planes
                for (int v = 0; v < 3; v++) {
                    float dx = uni(-s, s), dy = uni(-s, s), dz = uni(-s, s);
                    if (kind == 1) { dx = std::round(dx * 4) / 4; dy = 0.0f; dz = std::round(dz * 4) / 4; }
                    if (kind == 2 && v == 2 && (rng() % 5u) == 0) { dx = dy = dz = 0.0f; }                                   // degenerate ones
                    pos.push_back(cx + dx); pos.push_back(cy + dy); pos.push_back(cz + dz);
                }
                const uint32_t b = (uint32_t)(3 * t);
                faces.push_back(b); faces.push_back(b + 1); faces.push_back((rng() % 50u) == 0 ? b : b + 2);                 // (and repeated indices)
            }
            ok = rt_mesh_from_arrays(pos.data(), nullptr, pos.size() / 3, faces.data(), faces.size() / 3, (uint32_t)(rng() % 7u), identity, &meshes[(size_t)m]) == RT_OK;
            if (!ok) meshes.resize((size_t)m);
        }
        if (ok) {
            rt_octree oc;
            // (bounded so that a soup of large triangles with a limit of 0 per node cannot fill the memory with references)
            const size_t depth = 1 + rng() % 9u, per_node = rng() % 12u, per_leaf = rng() % 8u, keep = rng() % 9u;
            if (rt_octree_build(meshes.data(), meshes.size(), depth, per_node, per_leaf, keep, &oc) == RT_OK) {
                // every reference indexes an existing vertex; the words are read once so that the sanitizers see the whole output
                uint64_t sum = 0;
                for (size_t i = 0; i < oc.n_ref; i++) for (int k = 0; k < 3; k++) if (oc.tri_refs[4 * i + k] >= oc.n_vert) { std::fprintf(stderr, "case %d: reference out of range\n", c); return 1; }
                for (size_t i = 0; i < oc.n_words; i++) sum += oc.nodes[i];
                total_words += oc.n_words + (sum & 1u);
                rt_octree_free(&oc);
            }
        }
        for (rt_mesh &m : meshes) rt_mesh_free(&m);
    }
    // the OBJ reader on well-formed and garbled text: it must load or refuse, never read out of bounds
    const char *words[] = {"v", "vn", "vt", "f", "o", "g", "s", "usemtl", "#", "1", "-1", "0", "2/3", "1//2", "4/5/6", "-2/-1/-3", "1e40", "nan", "0.5", "-3.25", "abc", "/", "//", "1/", "99999999999", "\n", "\n", "\n"};
    size_t loaded = 0;
    for (int c = 0; c < cases * 10; c++) {
        std::string text;
        const bool garble = (rng() % 2u) == 0;
        const int nv = 1 + (int)(rng() % 30u);
        for (int i = 0; i < nv; i++) text += "v " + std::to_string(uni(-1, 1)) + " " + std::to_string(uni(-1, 1)) + " " + std::to_string(uni(-1, 1)) + "\n";
        if (rng() % 2u) for (int i = 0; i < nv; i++) text += "vn 0 1 0\n";
        for (int i = 0, nf = (int)(rng() % 40u); i < nf; i++) {
            const int corners = (rng() % 40u) == 0 ? 5 : 3 + (int)(rng() % 2u);      // (now and then a pentagon, which the reader refuses)
            text += "f";
            for (int k = 0; k < corners; k++) { const int idx = 1 + (int)(rng() % (unsigned)(nv + (garble ? 3 : 0))); text += " " + std::to_string((rng() % 7u) == 0 ? -idx : idx) + ((rng() % 3u) == 0 ? "//" + std::to_string(idx) : ""); }
            text += "\n";
        }
        if (garble) for (int i = 0, n = (int)(rng() % 60u); i < n; i++) { text.insert(rng() % (text.size() + 1), std::string(words[rng() % (sizeof words / sizeof *words)]) + " "); }
        const char *path = "/tmp/host_sanitize_case.obj";
        FILE *f = std::fopen(path, "wb");
        if (!f) return 1;
        std::fwrite(text.data(), 1, (rng() % 9u) == 0 ? text.size() / 2 : text.size(), f);      // (sometimes cut off in the middle)
        std::fclose(f);
        rt_mesh m;
        if (rt_mesh_load_obj(path, (uint32_t)(rng() % 7u), identity, &m) == RT_OK) {
            for (size_t i = 0; i < m.n_tri; i++) for (int k = 0; k < 3; k++) if (m.triangles[4 * i + k] >= m.n_vert) { std::fprintf(stderr, "obj case %d: index out of range\n", c); return 1; }
            loaded++;
            rt_mesh_free(&m);
        }
    }
    std::printf("%d cases, %zu node words; %d OBJ texts, %zu loaded\n", cases, total_words, cases * 10, loaded);
    return 0;
}
