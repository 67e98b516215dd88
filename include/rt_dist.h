/*
 * rt_dist.h -- the multi-GPU side of the MI355X ray-tracing pass behind the C ABI: one process per GPU, RCCL over xGMI.
 *
 * No counterpart in the reference, which is one process on one GL context.  The frame is cut into strips of `band_rows` rows dealt
 * round-robin to the ranks (rt_frame.band_rows / band_parts / band_part, compact_rows = 1: include/rt_hotpath.h); pixels are
 * independent (app/assets/ray_tracer.frag has no inter-pixel communication; the blend of App.cpp:49,104 is per pixel), so the only
 * exchanges are
 *   - once per scene: the four buffers of App.cpp:23-26, the octree bounds and the cubemap, broadcast from a root rank
 *     (rt_dist_scene_create, rt_dist_skybox_set);
 *   - once per presented frame: the ranks' bands gathered into the root's frame (rt_dist_gather_frame) -- grouped point-to-point
 *     sends, each peer's strips received directly at their rows of the frame: xGMI is point to point, every peer has its own link
 *     to the root, and nothing is staged or re-interleaved afterwards.
 * The N-rank image equals the one-rank image bit for bit.  RCCL (librccl.so.1) is loaded when rt_dist_* is first used; the rest
 * of the library does not depend on it.  Same conventions as rt_hotpath.h: RT_OK or an RT_E* code, rt_last_error().
 */
#ifndef RT_DIST_H
#define RT_DIST_H

#include "rt_hotpath.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_dist rt_dist;
#define RT_DIST_ID_BYTES 128

/* The root rank makes an id and hands it to the other ranks by whatever means the application has (a file, an environment
 * variable, MPI, a socket); every rank then joins with it.  `device`: the HIP device of this rank (one rank per device). */
int rt_dist_make_id(uint8_t id[RT_DIST_ID_BYTES]);
int rt_dist_init(const uint8_t id[RT_DIST_ID_BYTES], int world_size, int rank, int device, rt_dist **out);
int rt_dist_rank(const rt_dist *d);
int rt_dist_world_size(const rt_dist *d);
int rt_dist_destroy(rt_dist *d);

/* rt_scene_create on every rank from the ROOT's arrays (the other ranks pass null pointers and zero sizes; they are ignored
 * there): sizes and bounds, then the four arrays are broadcast device to device, and each rank derives its own traversal
 * records.  The root validates the arrays as rt_scene_create does; a failure there is reported on every rank. */
int rt_dist_scene_create(rt_dist *d, int root, const float *vertices, const float *normals, size_t n_vert,
                         const uint32_t *tri_refs, size_t n_ref, const uint32_t *nodes, size_t n_words,
                         const float bounds_min[3], const float bounds_max[3], rt_scene **out);
/* rt_skybox_set on every rank from the root's six faces. */
int rt_dist_skybox_set(rt_dist *d, int root, rt_scene *scene, const uint8_t *const faces[6], int w, int h);

/* Fills the band selection of `frame` for this rank: strips of `band_rows` rows (16 is the library's default choice),
 * band_parts = world size, band_part = rank, compact_rows = 1. */
int rt_dist_frame_bands(const rt_dist *d, uint32_t band_rows, rt_frame *frame);

/* Gathers one plane of a frame.  band: this rank's compact band buffer in device memory ([rt_band_row_count(frame)][width]
 * elements of bytes_per_pixel bytes: 16 for the colour plane, 4 for depth).  frame_plane: on the root, the full plane in device
 * memory ([height][width]); ignored elsewhere.  Enqueued on `stream` (a hipStream_t; the caller orders it after the renders that
 * fill `band`); returns after the enqueue.  Every rank must call it with the same frame geometry. */
int rt_dist_gather_frame(rt_dist *d, int root, const rt_frame *frame, const void *band, void *frame_plane, size_t bytes_per_pixel, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RT_DIST_H */
