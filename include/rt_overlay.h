/* rt_overlay.h -- the pass that follows the ray-tracing draw (C ABI, HIP kernels).
 *
 * After the fullscreen ray-tracing draw the reference rasterises nine textured meshes into the same framebuffer with
 * GL_DEPTH_TEST / GL_LEQUAL against the depth the ray tracer wrote through gl_FragDepth, blending disabled
 * (app/core/src/App.cpp:52-54,142-166).  The depth attachment is GL_DEPTH_COMPONENT24 (Framebuffer.cpp:10-12), so what
 * the rasterised fragments are tested against is a 24-bit value, not the float the shader wrote.  This header covers
 * the attachment value of a depth plane, the depth-tested merge of fragments produced by any rasteriser, the 8-bit image the
 * window receives, and the pass itself:
 * rt_overlay_draw rasterises an indexed mesh once per instance matrix with the reference's vertex and fragment stages
 * (Vert_PosNormTex.vert, Frag_LightingSimple.frag) into a colour plane and that 24-bit depth plane.
 */
#ifndef RT_OVERLAY_H
#define RT_OVERLAY_H

#include "rt_hotpath.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RT_DEPTH24_MAX 16777215u

/* Value the reference's 24-bit depth attachment holds for each pixel after the ray-tracing draw, as llvmpipe stores
 * it: depth24 = round-to-nearest-even(float(clamp(depth, 0, 1) * 16777215.0f)) -- the product is rounded to binary32
 * before the conversion to integer, which matters below 0.5.  `depth` is the plane rt_target.depth receives.  Pinned by
 * read-backs of the attachment in the reference runs on llvmpipe (the GL_FLOAT read-back of the attachment equals
 * float(depth24 / (2^24 - 1)) in every pixel of every fixture, depths from 0 to 1).
 * memory: RT_MEM_DEVICE (all planes on one HIP device, found from the pointers; the call is ordered on `stream`, a hipStream_t or NULL, and
 * returns after the enqueue) or RT_MEM_HOST (staged through device memory; returns when done). */
int rt_depth_to_attachment(const float *depth, uint32_t *depth24, size_t n_pixels, int memory, void *stream);

/* Per-fragment operations of the overlay draws (App.cpp:142-145: depth test on, GL_LEQUAL as set at App.cpp:54, depth
 * writes on, blending off): for every pixel i with frag_mask[i] != 0 (NULL mask: every pixel) whose fragment passes
 * frag_depth24[i] <= depth24[i], colour[i] = frag_color[i] (4 floats) and depth24[i] = frag_depth24[i].  One call merges
 * one layer of fragments (at most one per pixel); the reference's nine draws are nine layers applied in draw order, or
 * one layer already resolved by the rasteriser -- the result is the same because LEQUAL keeps the nearest of all
 * fragments and, between equal depths, the last drawn.  memory / stream as above. */
int rt_overlay_composite(float *color, uint32_t *depth24, const float *frag_color, const uint32_t *frag_depth24,
                         const uint8_t *frag_mask, size_t n_pixels, int memory, void *stream);

/* The mesh and texture of the overlay draws, resident on one HIP device.  vertices: [n_vert][8] floats = position, normal,
 * texcoord, the reference's Vertex (GLUtils.hpp:29-34) with the attribute layout of App.cpp:65-69; indices: n_idx (a multiple
 * of three) 32-bit indices, GL_TRIANGLES (App.cpp:163); texture_rgba: [tex_height][tex_width][4] bytes, row 0 = the row GL
 * calls the bottom one (TextureFromFile flips the image before the upload, GLUtils.cpp:187-202).  All three are host arrays and
 * are copied.  RT_ESCENE for an index beyond the vertex array. */
typedef struct rt_overlay_mesh rt_overlay_mesh;
int rt_overlay_mesh_create(const float *vertices, size_t n_vert, const uint32_t *indices, size_t n_idx,
                           const uint8_t *texture_rgba, uint32_t tex_width, uint32_t tex_height, int device, rt_overlay_mesh **out);
int rt_overlay_mesh_destroy(rt_overlay_mesh *mesh);

/* Uniforms of the pass (App.cpp:152-162): viewProj (column-major, Camera::GetViewProj), the light (App.hpp:79-81), and per
 * draw the matrices `world` and `worldIT` = transpose(inverse(world)), [n_draws][16] floats each, column-major, host memory. */
typedef struct rt_overlay_pass {
    float view_proj[16];
    float light_dir[3], la[3], ld[3];
    uint32_t n_draws;
    const float *world, *world_it;
} rt_overlay_pass;

/* The draws of App.cpp:157-164 over `color` ([height][width][4] floats) and `depth24` ([height][width], the values of
 * rt_depth_to_attachment), both updated in place; row 0 is the bottom row of the GL window, as in rt_target.  State as the
 * reference sets it: back faces (clockwise in the window) culled, GL_LEQUAL depth test with depth writes, no blending, viewport =
 * the whole frame, depth range [0, 1].  Coverage, depth and colour are those of the reference's GL on llvmpipe bit for bit: vertex
 * stage, clipping against the six frustum planes, 1/256-pixel snapping with the left/bottom fill rule, plane equations and their
 * evaluation, the fragment stage and the texture lookup -- GL_LINEAR_MIPMAP_LINEAR / GL_LINEAR with GL_REPEAT on the mip chain
 * glGenerateMipmap leaves (GLUtils.cpp:211-224), in that GL's 8-bit fixed point with its level of detail per 2 x 2 quad (pinned by
 * read-backs of that GL; the arithmetic is stated in rt_raster.hip).  draws x triangles x 7 must stay below 2^32 (RT_ELIMIT).
 * memory / stream as above; the mesh keeps scratch buffers (8 bytes per pixel and 176 bytes per possible triangle) between
 * calls, so calls on one mesh must not overlap. */
int rt_overlay_draw(rt_overlay_mesh *mesh, const rt_overlay_pass *pass, float *color, uint32_t *depth24,
                    uint32_t width, uint32_t height, int memory, void *stream);

/* What the window shows: Framebuffer::Blit (Framebuffer.cpp:33-35) copies the RGBA32F attachment onto the window's 8-bit
 * framebuffer.  byte = round-to-nearest-even(float(clamp(c, 0, 1) * 255.0f)) per component, NaN -> 0 (pinned by read-backs of the
 * window framebuffer on llvmpipe).  color: [n_pixels][4] floats; rgba8: [n_pixels][4] bytes.  memory / stream as above. */
int rt_present_rgba8(const float *color, uint8_t *rgba8, size_t n_pixels, int memory, void *stream);

/* rt_overlay_draw for one owner's rows of the frame, with rt_frame's rule (rt_hotpath.h): strips of band_rows rows are dealt out to
 * band_parts owners and this call draws the strips of band_part; with compact_rows the two planes hold only those rows
 * (rt_band_row_count() of them), else the whole frame, of which the other rows are left alone.  The fragments are those of the
 * whole-frame call: the union of the owners' results equals rt_overlay_draw's.  No counterpart in the reference (one GPU draws its
 * whole frame); it keeps the frame exact when several GPUs each accumulate their rows, since the reference's next ray-tracing draw
 * blends over the finished frame, rasterised pixels included. */
int rt_overlay_draw_band(rt_overlay_mesh *mesh, const rt_overlay_pass *pass, float *color, uint32_t *depth24,
                         uint32_t width, uint32_t height, uint32_t band_rows, uint32_t band_parts, uint32_t band_part,
                         uint32_t compact_rows, int memory, void *stream);

#ifdef __cplusplus
}
#endif
#endif
