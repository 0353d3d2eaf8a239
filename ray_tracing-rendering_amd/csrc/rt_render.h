/*
 * rt_render.h -- what every translation unit of the library shares about one render call: the kernel
 * parameter block, the tile -> pixel map of renderer/renderer.h:61-62, small wave helpers.
 */
#pragma once

#include "rt_device.h"

struct RenderK {
    int W, H;
    int x0, y0, x1, y1;
    int spp, max_depth, rr_start;
    uint32_t seed;
    int tiles_x, tiles_y;
    const int* tile_ids; /* owned tiles, reference dispatch numbering (renderer.h:61-62) */
    int n_tiles;
    int chunks;
    int integrator; /* RTR_INTEGRATOR_* (the wavefront's extend stage needs it for rays that miss) */
    double* partial;             /* [n_tiles*chunks][3][RTR_BLOCK] un-normalised sums */
    unsigned long long* stats;   /* samples, closest segments, shadow segments; [7] = workgroups a cancel interrupted */
    const uint32_t* cancel;      /* rtr_cancel(): id of the newest render it covers; this render stops once *cancel >= render_id */
    uint32_t render_id;
    int* done;                   /* [n_tiles*chunks]: 1 = the workgroup finished every sample of its chunk */
};

RT_DEV bool render_cancelled(const RenderK& P) {
    return __hip_atomic_load(P.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= P.render_id;
}

RT_DEV void tile_pixel(const RenderK& P, int slot, int tid, int& i, int& j, bool& active) {
    const int tile = P.tile_ids[slot];
    const int tile_y = (P.tiles_y - 1) - tile / P.tiles_x; /* renderer.h:61-62 */
    const int tile_x = tile % P.tiles_x;
    i = tile_x * 16 + (tid & 15);
    j = tile_y * 16 + (tid >> 4);
    active = i >= P.x0 && i < P.x1 && j >= P.y0 && j < P.y1;
}

RT_DEV unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

struct ResolveK {
    RenderK r;
    double* out; /* linear mean radiance, 3 doubles per pixel */
    long long row_stride;
};

