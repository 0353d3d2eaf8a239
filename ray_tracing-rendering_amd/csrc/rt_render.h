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
    /* Guided chunks (megakernel, spp_chunks = 0): the first `n_big` chunks of a pixel hold `big_spp` samples each, the
     * others `small_spp` (the last one what is left), and the launch runs every tile's big chunks before any small
     * one, so the launch drains through short workgroups.  small_spp = 0: `chunks` equal parts. */
    int n_big, big_spp, small_spp;
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

/* samples [s0, s1) of chunk c of a pixel */
RT_DEV void chunk_range(const RenderK& P, int c, int& s0, int& s1) {
    if (P.small_spp == 0) {
        s0 = (int)((long long)c * P.spp / P.chunks);
        s1 = (int)((long long)(c + 1) * P.spp / P.chunks);
    } else if (c < P.n_big) {
        s0 = c * P.big_spp, s1 = s0 + P.big_spp;
    } else {
        s0 = P.n_big * P.big_spp + (c - P.n_big) * P.small_spp, s1 = s0 + P.small_spp;
    }
    s0 = s0 < P.spp ? s0 : P.spp;
    s1 = s1 < P.spp ? s1 : P.spp;
    if (c == P.chunks - 1) s1 = P.spp;
}
/* which (owned tile, chunk) workgroup `b` of a megakernel launch renders: big chunks of every tile first */
RT_DEV void mega_work(const RenderK& P, int b, int& tile_slot, int& c) {
    if (P.small_spp == 0) {
        tile_slot = b / P.chunks, c = b % P.chunks;
    } else if (b < P.n_tiles * P.n_big) {
        tile_slot = b / P.n_big, c = b % P.n_big;
    } else {
        const int n_small = P.chunks - P.n_big, q = b - P.n_tiles * P.n_big;
        tile_slot = q / n_small, c = P.n_big + q % n_small;
    }
}

RT_DEV unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

struct ResolveK {
    RenderK r;
    double* out; /* linear mean radiance, 3 doubles per pixel */
    long long row_stride; /* pixels per row of `out`; < 0: `out` is PACKED -- owned tile k of the call at out[k * 768 ...]
                             as 16 rows (lowest y first) of 16 pixels, and tile_done[k] = 1 once its sums are stored */
    unsigned char* tile_done;
};

