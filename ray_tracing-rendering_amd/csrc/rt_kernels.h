/*
 * rt_kernels.h -- __global__ kernels of the path tracer (gfx950).
 *
 *  k_mega      megakernel: one workgroup = one 16x16 image tile (renderer/renderer.h:40-67),
 *              one lane = one pixel.  A lane runs its samples back to back: when a path ends
 *              the lane starts its pixel's next camera sample in the same loop iteration
 *              (in-lane path regeneration), so the 64 lanes of a wave stay busy although path
 *              lengths differ (P(k=4)=0.45, long tail).  Samples of a pixel are
 *              summed in sample order like renderer.h:72-79.
 *  k_resolve   adds the per-chunk partial sums of a pixel in chunk order, scales by 1/spp
 *              (renderer.h:131) and stores linear mean radiance.
 *  k_li        Integrator::Li of single camera samples or caller-given rays, one lane each (rtr_li_samples / rtr_li_rays).
 */
#pragma once

#include "rt_render.h"

#ifndef RTR_MEGA_WAVES
#define RTR_MEGA_WAVES 4 /* min waves per SIMD the register allocator must leave room for */
#endif
#ifndef RTR_PROGRAM_WAVES
#define RTR_PROGRAM_WAVES 3 /* media scenes are traversal-latency bound: a third wave pays for the spills it causes */
#endif

/* Waves per SIMD the register allocator must leave room for, per kernel variant (measured: scene 23
 * 4.36 -> 5.12, scene 1 1.36 -> 1.63 Gsamples/s at 3 instead of 2; the variants with the full light
 * set or the reference-order walk lose 10-30 % there to spills) */
constexpr int mega_waves(int integ, int trav, int ms) {
    if (ms == RT_MS_LEAN) return RTR_MEGA_WAVES;
    if (trav == RT_TRAV_PROGRAM) return RTR_PROGRAM_WAVES;
    if (trav == RT_TRAV_MEDIA || trav == RT_TRAV_EXACT) return 2;
    if (ms == RT_MS_QUADLIT) return 3;
    if (integ == RTR_INTEGRATOR_RR || integ == RTR_INTEGRATOR_PATH) return 3;
    return 2;
}

/* Per-lane path state that the ray casts do not touch lives in LDS between shading steps
 * ("parked"), so it does not occupy VGPRs across the traversal loops: throughput, radiance of
 * the sample, pixel sum, previous BSDF pdf and the pending light contribution.  Word k of lane l
 * is park[k * RTR_BLOCK + l] (8-byte words: conflict-free ds_read_b64 / ds_write_b64). */
#define RT_PARK_WORDS 19
struct Park {
    double* base;
    RT_DEV V3 get3(int k) const { return mk(base[k * RTR_BLOCK], base[(k + 1) * RTR_BLOCK], base[(k + 2) * RTR_BLOCK]); }
    RT_DEV void set3(int k, V3 v) const {
        base[k * RTR_BLOCK] = v.x, base[(k + 1) * RTR_BLOCK] = v.y, base[(k + 2) * RTR_BLOCK] = v.z;
    }
    RT_DEV double get(int k) const { return base[k * RTR_BLOCK]; }
    RT_DEV void set(int k, double v) const { base[k * RTR_BLOCK] = v; }
};
enum { PK_THR = 0, PK_L = 3, PK_ACC = 6, PK_PDF = 9, PK_NCLOSEST = 10, PK_NSHADOW = 11, /* every variant */
       PK_CONTRIB = 12, PK_SWI = 15, PK_STMAX = 18 };                                       /* deferred shadow ray only */
/* words a variant parks: the deferred shadow request exists only where the shadow ray is cast after the
 * BSDF sample (MIS-type integrators on scenes without media); fewer words = more workgroups per CU where
 * LDS, not registers, is the limit (the lean RR kernel: 92 VGPRs) */
constexpr int park_words(int integ, int trav) {
    return (integ == RTR_INTEGRATOR_RR || integ == RTR_INTEGRATOR_PATH || trav == RT_TRAV_PROGRAM || trav == RT_TRAV_MEDIA)
               ? 12
               : RT_PARK_WORDS;
}

template <int INTEG, int TRAV, int MS>
__global__ void __launch_bounds__(RTR_BLOCK, mega_waves(INTEG, TRAV, MS))
    k_mega(const DScene* __restrict__ scp, const RenderK P, const int stack_words) {
    extern __shared__ int lds_stack[];
    const DScene& sc = *scp;
    const Stack st{lds_stack + threadIdx.x};
    const Park pk{reinterpret_cast<double*>(lds_stack + stack_words * RTR_BLOCK) + threadIdx.x};
#ifdef RTR_REGION_PROFILE
    if (threadIdx.x < 4 * 2 * RT_PROF_REGIONS + 8) rt_prof_lds[threadIdx.x] = 0;
    if (threadIdx.x + 256 < 4 * 2 * RT_PROF_REGIONS + 8) rt_prof_lds[threadIdx.x + 256] = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) rt_prof_lds[4 * 2 * RT_PROF_REGIONS + (threadIdx.x >> 6) * 2] = __builtin_readcyclecounter();
#endif
    int slot, chunk;
    mega_work(P, blockIdx.x, slot, chunk);
    const int cell = slot * P.chunks + chunk; /* partial sum / completion word of this (tile, chunk) */
    int i, j;
    bool active;
    tile_pixel(P, slot, threadIdx.x, i, j, active);
    /* samples [s, s_end) of this pixel belong to this chunk */
    int s, s_end_;
    chunk_range(P, chunk, s, s_end_);
    const int s_end = s_end_;
    pk.set3(PK_ACC, mk(0, 0, 0));
    PathCounters cnt;
    cnt.closest = 0, cnt.shadow = 0;
    uint32_t n_samples = 0;
    PathState ps;
    uint32_t rng = 1;
    bool fresh = true;
    bool done = !active || s >= s_end;
    if (TRAV == RT_TRAV_MEDIA) {
        /* media draw random numbers inside both ray casts: keep the reference's statement order */
        V3 acc = mk(0, 0, 0);
        while (!done) {
            if (fresh) { /* renderer.h:73-75 under the per-sample seed */
                if ((s & 7) == 0 && render_cancelled(P)) break;
                rng = rtr_sample_seed_inline(P.seed, P.W, i, j, s);
                const Real u = (i + rng_next(rng)) / (P.W - 1);
                const Real v = (j + rng_next(rng)) / (P.H - 1);
                V3 ro, rd;
                Real tm;
                camera_get_ray(sc.camera, u, v, rng, ro, rd, tm);
                path_begin(ps, ro, rd, tm);
                fresh = false;
            }
            if (!bounce<INTEG, TRAV>(sc, ps, rng, st, P.max_depth, P.rr_start, cnt)) {
                acc = add(acc, ps.L); /* renderer.h:77-78 */
                ++n_samples;
                ++s;
                fresh = true;
                done = s >= s_end;
            }
        }
        pk.set3(PK_ACC, acc);
    } else {
        /* Without media the shadow ray draws nothing, so it can be cast AFTER the BSDF sample of
         * the same bounce, when the hit record is dead.  Every live lane runs the same phases in
         * every iteration (closest hit, shade, shadow ray, end-of-sample + regeneration), so a wave
         * stays in lockstep although path lengths differ; the sums still see their terms in the
         * reference's order (emission, then the light sample of the same bounce). */
        auto begin_sample = [&]() { /* renderer.h:73-75 under the per-sample seed */
            int pi, pj;
            bool in_region;
            tile_pixel(P, slot, threadIdx.x, pi, pj, in_region); /* recomputed: not worth two live registers */
            rng = rtr_sample_seed_inline(P.seed, P.W, pi, pj, s);
            const Real u = (pi + rng_next(rng)) / (P.W - 1);
            const Real v = (pj + rng_next(rng)) / (P.H - 1);
            camera_get_ray(sc.camera, u, v, rng, ps.ro, ps.rd, ps.tm);
            ps.depth = 0, ps.specular_bounce = false;
            pk.set3(PK_THR, mk(1.0, 1.0, 1.0));
            pk.set3(PK_L, mk(0.0, 0.0, 0.0));
            pk.set(PK_PDF, 0.0);
        };
        /* cast counters live in LDS as well (exact in a double up to 2^53) */
        pk.set(PK_NCLOSEST, 0.0);
        pk.set(PK_NSHADOW, 0.0);
        if (!done) begin_sample();
#ifdef RTR_PHASE_CLOCKS
        long long clk_closest = 0, clk_shade = 0, clk_shadow = 0, clk_other = 0, clk_t = wall_clock64();
#define RTR_CLK(acc) do { const long long now_ = wall_clock64(); acc += now_ - clk_t; clk_t = now_; } while (0)
#else
#define RTR_CLK(acc) do { } while (0)
#endif
        while (!done) {
            bool pending = false, ended = false;
            RTR_CLK(clk_other);
            RT_REGION(RG_OTHER);
            {
                Hit rec;
                rec.u = 0, rec.v = 0;
                pk.set(PK_NCLOSEST, pk.get(PK_NCLOSEST) + 1.0);
                const bool hit_any = cast_closest<TRAV, MS == RT_MS_FULL>(sc, ps.ro, ps.rd, ps.tm, rec, rng, st);
                RTR_CLK(clk_closest);
                if (!hit_any) {
                    RT_REGION(RG_MISS);
                    pk.set3(PK_L, add(pk.get3(PK_L), miss_radiance<INTEG, MS>(sc, pk.get3(PK_THR), ps.ro, ps.rd, ps.depth,
                                                                          ps.specular_bounce, pk.get(PK_PDF))));
                    ended = true;
                } else {
                    ps.thr = pk.get3(PK_THR);
                    /* shading adds at most ONE term (the emission) to L before the light sample is
                     * resolved, so it can start from zero and be added to the parked sum afterwards:
                     * L + e is the same rounding as the reference's L += e */
                    ps.L = mk(0.0, 0.0, 0.0);
                    ps.prev_bsdf_pdf = pk.get(PK_PDF);
                    bool go;
                    if (INTEG == RTR_INTEGRATOR_RR) {
                        go = shade_rr<MS>(sc, ps, rec, rng, P.rr_start);
                    } else if (INTEG == RTR_INTEGRATOR_PATH) {
                        go = shade_path<MS>(sc, ps, rec, rng);
                    } else {
                        const V3 wo = neg(unit(ps.rd));
                        ShadowReq rq;
                        MatCtx mc = mat_prepare<MS>(sc, rec);
                        shade_a_mis<MS, INTEG>(sc, ps, rec, mc, wo, rng, rq);
                        if (TRAV == RT_TRAV_PROGRAM) {
                            /* media draw inside the shadow cast: it keeps its place between the light
                             * sample and the BSDF sample (mis_path_integrator.h:96-106) */
                            if (ps.L.x != 0.0 || ps.L.y != 0.0 || ps.L.z != 0.0) {
                                pk.set3(PK_L, add(pk.get3(PK_L), ps.L));
                                ps.L = mk(0.0, 0.0, 0.0);
                            }
                            if (rq.valid) {
                                pk.set(PK_NSHADOW, pk.get(PK_NSHADOW) + 1.0);
                                if (!cast_shadow<TRAV>(sc, rec.p, rq.wi, rq.tmax, rng, st))
                                    pk.set3(PK_L, add(pk.get3(PK_L), rq.contrib));
                                mc = mat_prepare<MS>(sc, rec); /* cheaper than keeping it in registers across the cast */
                            }
                        } else if (rq.valid) { /* parked until the shadow ray is cast */
                            pending = true;
                            pk.set3(PK_SWI, rq.wi);
                            pk.set(PK_STMAX, rq.tmax);
                            pk.set3(PK_CONTRIB, rq.contrib);
                        }
                        go = shade_b_mis<MS, INTEG>(sc, ps, rec, mc, wo, rng, P.rr_start);
                    }
                    RT_REGION(RG_PARK);
                    ps.ro = rec.p; /* next ray origin and shadow ray origin */
                    pk.set3(PK_THR, ps.thr);
                    if (ps.L.x != 0.0 || ps.L.y != 0.0 || ps.L.z != 0.0) pk.set3(PK_L, add(pk.get3(PK_L), ps.L));
                    pk.set(PK_PDF, ps.prev_bsdf_pdf);
                    ended = !go || ++ps.depth >= P.max_depth;
                }
            }
            RTR_CLK(clk_shade);
            RT_REGION(RG_OTHER);
            if (pending) { /* mis_path_integrator.h:210-213, origin = the hit point = ps.ro */
                pk.set(PK_NSHADOW, pk.get(PK_NSHADOW) + 1.0);
                if (!cast_shadow<TRAV>(sc, ps.ro, pk.get3(PK_SWI), pk.get(PK_STMAX), rng, st))
                    pk.set3(PK_L, add(pk.get3(PK_L), pk.get3(PK_CONTRIB)));
            }
            RTR_CLK(clk_shadow);
            RT_REGION(RG_REGEN);
            if (ended) {
                pk.set3(PK_ACC, add(pk.get3(PK_ACC), pk.get3(PK_L))); /* renderer.h:77-78 */
                ++n_samples;
                ++s;
                /* rtr_cancel(): polled every 8th sample of a pixel -- the load is a dependent memory round
                 * trip in the lane's critical path */
                done = s >= s_end || ((s & 7) == 0 && render_cancelled(P));
                if (!done) begin_sample();
            }
        }
        cnt.closest = (uint32_t)pk.get(PK_NCLOSEST);
        cnt.shadow = (uint32_t)pk.get(PK_NSHADOW);
#ifdef RTR_PHASE_CLOCKS
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&P.stats[3], (unsigned long long)clk_closest);
            atomicAdd(&P.stats[4], (unsigned long long)clk_shade);
            atomicAdd(&P.stats[5], (unsigned long long)clk_shadow);
            atomicAdd(&P.stats[6], (unsigned long long)clk_other);
        }
#endif
    }
#ifdef RTR_REGION_PROFILE
    RT_REGION(RG_OTHER);
    __syncthreads();
    if (threadIdx.x < 2 * RT_PROF_REGIONS) {
        unsigned long long v = 0;
        for (int w = 0; w < 4; ++w) v += rt_prof_lds[w * 2 * RT_PROF_REGIONS + threadIdx.x];
        if (v) atomicAdd(&P.stats[RT_PROF_BASE + threadIdx.x], v);
    }
#endif
    const V3 acc = pk.get3(PK_ACC);
    double* out = P.partial + (size_t)cell * 3 * RTR_BLOCK + threadIdx.x;
    out[0] = acc.x;
    out[RTR_BLOCK] = acc.y;
    out[2 * RTR_BLOCK] = acc.z;
    unsigned long long a = wave_sum(n_samples), b = wave_sum(cnt.closest), c = wave_sum(cnt.shadow);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&P.stats[0], a);
        atomicAdd(&P.stats[1], b);
        atomicAdd(&P.stats[2], c);
    }
    /* a cancelled render leaves unfinished tiles untouched, like the reference's workers
     * (renderer.h:52-59): k_resolve stores a tile only when all its chunks ran to the end */
    const int interrupted = __syncthreads_or(active && s < s_end);
    if (threadIdx.x == 0) {
        P.done[cell] = !interrupted;
        if (interrupted) atomicAdd(&P.stats[7], 1ull);
    }
}


/* rtr_li_samples / rtr_li_rays: Integrator::Li (renderer/integrator.h:12-19) of n camera samples of the image P
 * describes, or of n caller-given rays; one lane each.  `in` per item: camera sample = (i, j, s) as three int32 in
 * the first 12 bytes; ray = rtr_li_ray.  `out` per item: radiance (3 doubles), rng state at exit, closest / shadow
 * segment counts. */
struct LiOut {
    double L[3];
    uint32_t rng_exit;
    int32_t n_closest, n_shadow, pad;
};
template <int INTEG, int TRAV>
__global__ void __launch_bounds__(RTR_BLOCK) k_li(const DScene sc, const RenderK P, const int32_t* __restrict__ ijs,
                                                   const rtr_li_ray* __restrict__ rays, LiOut* __restrict__ out, long long n) {
    extern __shared__ int lds_stack[];
    const Stack st{lds_stack + threadIdx.x};
    const long long k = (long long)blockIdx.x * RTR_BLOCK + threadIdx.x;
    if (k >= n) return;
    uint32_t rng;
    V3 ro, rd;
    Real tm;
    if (rays) {
        const rtr_li_ray r = rays[k];
        ro = ld3(r.origin), rd = ld3(r.direction), tm = r.time, rng = r.rng_state;
    } else {
        const int i = ijs[3 * k], j = ijs[3 * k + 1], s = ijs[3 * k + 2];
        rng = rtr_sample_seed_inline(P.seed, P.W, i, j, s);
        const Real u = (i + rng_next(rng)) / (P.W - 1);
        const Real v = (j + rng_next(rng)) / (P.H - 1);
        camera_get_ray(sc.camera, u, v, rng, ro, rd, tm);
    }
    PathState ps;
    path_begin(ps, ro, rd, tm);
    PathCounters cnt;
    cnt.closest = 0, cnt.shadow = 0;
    while (bounce<INTEG, TRAV>(sc, ps, rng, st, P.max_depth, P.rr_start, cnt)) {
    }
    LiOut o;
    o.L[0] = ps.L.x, o.L[1] = ps.L.y, o.L[2] = ps.L.z;
    o.rng_exit = rng, o.n_closest = (int)cnt.closest, o.n_shadow = (int)cnt.shadow, o.pad = 0;
    out[k] = o;
}

#ifdef RTR_TU_CAPI /* non-template kernels live in one translation unit */
__global__ void __launch_bounds__(RTR_BLOCK) k_resolve(const ResolveK R) {
    const RenderK& P = R.r;
    int i, j;
    bool active;
    tile_pixel(P, blockIdx.x, threadIdx.x, i, j, active);
    for (int c = 0; c < P.chunks; ++c) /* wave-uniform: scalar loads */
        if (!P.done[blockIdx.x * P.chunks + c]) return;
    const bool packed = R.row_stride < 0;
    if (packed && threadIdx.x == 0) R.tile_done[blockIdx.x] = 1;
    if (!active) return;
    const double* in = P.partial + (size_t)blockIdx.x * P.chunks * 3 * RTR_BLOCK + threadIdx.x;
    double r = 0, g = 0, b = 0;
    for (int c = 0; c < P.chunks; ++c) {
        if (c == 0) {
            r = in[0], g = in[RTR_BLOCK], b = in[2 * RTR_BLOCK];
        } else {
            r += in[0], g += in[RTR_BLOCK], b += in[2 * RTR_BLOCK];
        }
        in += 3 * RTR_BLOCK;
    }
    const double scale = 1.0 / P.spp; /* renderer.h:131 */
    double* o = packed ? R.out + ((long long)blockIdx.x * RTR_BLOCK + threadIdx.x) * 3
                       : R.out + ((long long)(j - P.y0) * R.row_stride + (i - P.x0)) * 3;
    o[0] = scale * r;
    o[1] = scale * g;
    o[2] = scale * b;
}

#endif /* RTR_TU_CAPI */
