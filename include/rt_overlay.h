/* rt_overlay.h -- the depth-buffer side of the pass that follows the ray-tracing draw (C ABI, HIP kernels).
 *
 * After the fullscreen ray-tracing draw the reference rasterises nine textured meshes into the same framebuffer with
 * GL_DEPTH_TEST / GL_LEQUAL against the depth the ray tracer wrote through gl_FragDepth, blending disabled
 * (app/core/src/App.cpp:52-54,142-166).  The depth attachment is GL_DEPTH_COMPONENT24 (Framebuffer.cpp:10-12), so what
 * the rasterised fragments are tested against is a 24-bit value, not the float the shader wrote.  This header covers
 * that fixed-function half: the attachment value of a depth plane, and the depth-tested merge of fragments produced by
 * any rasteriser.  Rasterising the meshes themselves (vertex stage, coverage, interpolation, texturing:
 * Vert_PosNormTex.vert, Frag_LightingSimple.frag) is NOT part of this library.
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

#ifdef __cplusplus
}
#endif
#endif
