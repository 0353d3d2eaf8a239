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
 *  k_test_*    one lane per golden-vector record (include/rtr_testrec.h): device unit parity.
 */
#pragma once

#include "rt_render.h"
#include "rtr_testrec.h"

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

#ifdef RTR_TU_CAPI /* non-template kernels live in one translation unit */
__global__ void __launch_bounds__(RTR_BLOCK) k_resolve(const ResolveK R) {
    const RenderK& P = R.r;
    int i, j;
    bool active;
    tile_pixel(P, blockIdx.x, threadIdx.x, i, j, active);
    if (!active) return;
    for (int c = 0; c < P.chunks; ++c) /* wave-uniform: scalar loads */
        if (!P.done[blockIdx.x * P.chunks + c]) return;
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
    double* o = R.out + ((long long)(j - P.y0) * R.row_stride + (i - P.x0)) * 3;
    o[0] = scale * r;
    o[1] = scale * g;
    o[2] = scale * b;
}

/* ---- device unit kernels over golden-vector records ---------------------------------------- */
template <int TRAV>
__global__ void __launch_bounds__(RTR_BLOCK) k_test_hits(const DScene sc, rtr_hit_record* recs, long long n) {
    extern __shared__ int lds_stack[];
    const Stack st{lds_stack + threadIdx.x};
    const long long k = (long long)blockIdx.x * RTR_BLOCK + threadIdx.x;
    if (k >= n) return;
    rtr_hit_record r = recs[k];
    uint32_t rng = r.rng_in;
    Hit rec;
    rec.u = rec.v = __builtin_nan("");
    rec.mat = -1;
    rec.t = 0, rec.p = mk(0, 0, 0), rec.n = mk(0, 0, 0), rec.front = false;
    Real tmax = r.t_max;
    bool h;
    if (TRAV == RT_TRAV_FAST) {
        int ref, inst;
        h = trace_fast<false>(sc, 0, sc.n_finst, ld3(r.o), ld3(r.d), r.time, r.t_min, tmax, ref, inst, st, 0);
        if (h) fast_finish<true>(sc, ld3(r.o), ld3(r.d), r.time, tmax, ref, inst, rec);
    } else if (TRAV == RT_TRAV_PROGRAM) {
        h = cast_closest<TRAV>(sc, ld3(r.o), ld3(r.d), r.time, rec, rng, st, r.t_min, tmax);
    } else {
        h = traverse<true, TRAV == RT_TRAV_MEDIA>(sc, sc.root, ld3(r.o), ld3(r.d), r.time, r.t_min, tmax, rec, rng, st, 0);
    }
    r.rng_out = rng;
    r.hit = h;
    r.front_face = h ? (int)rec.front : 0;
    r.material = h ? rec.mat : -1;
    r.pad = 0;
    r.t = h ? rec.t : 0;
    r.p[0] = h ? rec.p.x : 0, r.p[1] = h ? rec.p.y : 0, r.p[2] = h ? rec.p.z : 0;
    r.n[0] = h ? rec.n.x : 0, r.n[1] = h ? rec.n.y : 0, r.n[2] = h ? rec.n.z : 0;
    r.u = h ? rec.u : 0, r.v = h ? rec.v : 0;
    recs[k] = r;
}

__global__ void __launch_bounds__(RTR_BLOCK) k_test_materials(const DScene sc, rtr_mat_record* recs, long long n) {
    const long long k = (long long)blockIdx.x * RTR_BLOCK + threadIdx.x;
    if (k >= n) return;
    rtr_mat_record r = recs[k];
    Hit rec;
    rec.p = ld3(r.p), rec.n = ld3(r.n);
    rec.u = r.u, rec.v = r.v, rec.t = 1.0;
    rec.front = r.front_face != 0;
    rec.mat = r.material;
    const V3 wo = ld3(r.wo), wi = ld3(r.wi_in);
    uint32_t rng = r.rng_in;
    BSDFSample bs;
    bs.wi = mk(0, 0, 0), bs.f = mk(0, 0, 0), bs.pdf = 0, bs.is_specular = false, bs.is_transmission = false;
    const MatCtx mc = mat_prepare(sc, rec);
    const bool ok = mat_sample(mc, rec, wo, bs, rng);
    r.rng_out = rng;
    r.sample_ok = ok, r.is_specular = bs.is_specular, r.pad = 0;
    r.is_transmission = bs.is_transmission;
    r.s_wi[0] = bs.wi.x, r.s_wi[1] = bs.wi.y, r.s_wi[2] = bs.wi.z;
    r.s_f[0] = bs.f.x, r.s_f[1] = bs.f.y, r.s_f[2] = bs.f.z;
    r.s_pdf = bs.pdf;
    const V3 e = mat_eval(mc, wo, wi);
    r.eval[0] = e.x, r.eval[1] = e.y, r.eval[2] = e.z;
    r.pdf = mat_pdf(mc, rec, wo, wi);
    const V3 em = mat_emitted(mc, rec);
    r.emitted[0] = em.x, r.emitted[1] = em.y, r.emitted[2] = em.z;
    recs[k] = r;
}

__global__ void __launch_bounds__(RTR_BLOCK) k_test_lights(const DScene sc, rtr_light_record* recs, long long n) {
    const long long k = (long long)blockIdx.x * RTR_BLOCK + threadIdx.x;
    if (k >= n) return;
    rtr_light_record r = recs[k];
    const rtr_light l = ld_const(sc.lights, r.light);
    uint32_t rng = 0x2545F491u; /* the uniform environment light draws its direction itself */
    LightSample s = light_sample(l, ld3(r.p), r.u[0], r.u[1], rng, sc.image_bytes);
    r.Li[0] = s.Li.x, r.Li[1] = s.Li.y, r.Li[2] = s.Li.z;
    r.wi[0] = s.wi.x, r.wi[1] = s.wi.y, r.wi[2] = s.wi.z;
    r.pdf = s.pdf, r.dist = s.dist, r.is_delta = s.is_delta, r.pad2 = 0;
    r.pdf_dir = light_pdf(l, ld3(r.p), ld3(r.dir), sc.image_bytes);
    recs[k] = r;
}

template <int INTEG, int TRAV>
__global__ void __launch_bounds__(RTR_BLOCK) k_test_li(const DScene sc, const RenderK P, rtr_li_record* recs,
                                                        long long n) {
    extern __shared__ int lds_stack[];
    const Stack st{lds_stack + threadIdx.x};
    const long long k = (long long)blockIdx.x * RTR_BLOCK + threadIdx.x;
    if (k >= n) return;
    rtr_li_record r = recs[k];
    uint32_t rng = rtr_sample_seed_inline(P.seed, P.W, r.i, r.j, r.s);
    const Real u = (r.i + rng_next(rng)) / (P.W - 1);
    const Real v = (r.j + rng_next(rng)) / (P.H - 1);
    V3 ro, rd;
    Real tm;
    camera_get_ray(sc.camera, u, v, rng, ro, rd, tm);
    PathState ps;
    path_begin(ps, ro, rd, tm);
    PathCounters cnt;
    cnt.closest = 0, cnt.shadow = 0;
    while (bounce<INTEG, TRAV>(sc, ps, rng, st, P.max_depth, P.rr_start, cnt)) {
    }
    r.rng_exit = rng;
    r.L[0] = ps.L.x, r.L[1] = ps.L.y, r.L[2] = ps.L.z;
    r.n_closest = (int)cnt.closest, r.n_shadow = (int)cnt.shadow;
    recs[k] = r;
}
/* rtr_test_stream8: 8 bytes per lane in, 8 bytes per lane out */
__global__ void __launch_bounds__(RTR_BLOCK) k_test_sincos(unsigned long long* mismatches) {
    unsigned long long bad = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * RTR_BLOCK;
    for (unsigned long long s = (unsigned long long)blockIdx.x * RTR_BLOCK + threadIdx.x; s < (1ull << 32); s += stride) {
        const Real phi = 2.0 * RT_PI * ((uint32_t)s * 2.3283064365386963e-10); /* as random_cosine_direction / pbr_sample */
        Real s2, c2;
        sincos(phi, &s2, &c2);
        const Real s1 = sin(phi), c1 = cos(phi);
        bad += (__double_as_longlong(s1) != __double_as_longlong(s2)) | (__double_as_longlong(c1) != __double_as_longlong(c2));
    }
    bad = wave_sum(bad);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(mismatches, bad);
}

/* rtr_test_shared_division: div_shared() against the compiler's n / d on 2^32 operand pairs.  Three quarters of them
 * take both operands from the whole range the short form is used in (|d| in [2^-100, 2^100), |n| in [2^-300, 2^200),
 * random mantissas and signs); the rest are shaped like the rectangle test's (k - o) / d: a difference of two
 * coordinates below 1000 over a direction component in (-1, 1), small values of both included.  Quotients of
 * numerators below 2^-300 are only required to stay below 2^-200 (see div_shared). */
__global__ void __launch_bounds__(RTR_BLOCK) k_test_shared_div(unsigned long long* mismatches, unsigned per_thread) {
    unsigned long long x = 0x9E3779B97F4A7C15ull * ((unsigned long long)blockIdx.x * RTR_BLOCK + threadIdx.x + 1);
    auto next = [&]() {
        x ^= x >> 12, x ^= x << 25, x ^= x >> 27;
        return x * 0x2545F4914F6CDD1Dull;
    };
    auto make = [&](int emin, int espan) { /* +-1.m * 2^e, e in [emin, emin + espan) */
        const unsigned long long u = next();
        const int e = emin + (int)((u >> 52) % (unsigned)espan);
        const double m = __longlong_as_double((u & 0x800FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
        return ldexp(m, e);
    };
    unsigned long long bad = 0;
    for (unsigned k = 0; k < per_thread; ++k) {
        double n, d;
        if (k & 3) {
            d = make(-100, 200), n = make(-300, 500);
        } else {
            const double o = (double)(long long)(next() >> 11) * 0x1p-53 * 2000.0 - 1000.0;
            const double p = (k & 4) ? o + make(-60, 60) : (double)(long long)(next() >> 11) * 0x1p-53 * 2000.0 - 1000.0;
            n = p - o;
            d = (k & 8) ? make(-40, 40) : (double)(long long)(next() >> 11) * 0x1p-52 - 1.0;
        }
        if (!rcp_safe(d)) continue;
        const double r = rcp_refined(d);
        const double t = div_shared<true>(n, d, r, false), ref = n / d;
        if (__builtin_fabs(n) >= 0x1p-300)
            bad += __double_as_longlong(t) != __double_as_longlong(ref);
        else
            bad += !(__builtin_fabs(t) < 0x1p-200) || !(__builtin_fabs(ref) < 0x1p-200);
        const double g = div_shared<true>(n, d, r, true); /* guarded: every numerator */
        bad += __double_as_longlong(g) != __double_as_longlong(ref);
    }
    bad = wave_sum(bad);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(mismatches, bad);
}

/* rtr_test_issue_rates: shader cycles per wave-instruction, one instruction class per launch.  Every wave runs
 * `iters` trips of 32 independent instructions of the class between two s_memtime reads; with four waves on each SIMD
 * (grid = 4 workgroups per CU) the quotient cycles x 1 / (32 iters) of a wave is four times the SIMD's cost per
 * instruction when the class is bound by its pipe, and the single-wave issue cost when it is not.  out[0] += cycles of
 * every wave, out[1] += waves. */
#define RT_REP32(S) S S S S S S S S S S S S S S S S S S S S S S S S S S S S S S S S
template <int KIND>
__global__ void __launch_bounds__(RTR_BLOCK) k_test_issue_rate(unsigned long long* out, int iters, double seed) {
    double a = seed + threadIdx.x, b = seed * 0.5, c = 1.0 / (seed + 3.0), d = seed;
    float fa = (float)a, fb = (float)b, fc = (float)c;
    int ia = threadIdx.x, ib = 3;
    unsigned long long m = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile(RT_REP32("v_fma_f64 %0, %1, %2, %1\n") : "+v"(d) : "v"(b), "v"(c));
        if (KIND == 1) asm volatile(RT_REP32("v_add_f64 %0, %1, %2\n") : "+v"(d) : "v"(b), "v"(c));
        if (KIND == 2) asm volatile(RT_REP32("v_mul_f64 %0, %1, %2\n") : "+v"(d) : "v"(b), "v"(c));
        if (KIND == 3) asm volatile(RT_REP32("v_rcp_f64 %0, %1\n") : "+v"(d) : "v"(b));
        if (KIND == 4) asm volatile(RT_REP32("v_rsq_f64 %0, %1\n") : "+v"(d) : "v"(b));
        if (KIND == 5) asm volatile(RT_REP32("v_cmp_lt_f64 %0, %1, %2\n") : "=s"(m) : "v"(b), "v"(c));
        if (KIND == 6) asm volatile(RT_REP32("v_cndmask_b32 %0, %1, %2, vcc\n") : "+v"(ia) : "v"(ib), "v"(ia) : "vcc");
        if (KIND == 7) asm volatile(RT_REP32("v_mov_b32 %0, %1\n") : "+v"(ia) : "v"(ib));
        if (KIND == 8) asm volatile(RT_REP32("v_fma_f32 %0, %1, %2, %1\n") : "+v"(fa) : "v"(fb), "v"(fc));
        if (KIND == 9) asm volatile(RT_REP32("s_and_b64 %0, %0, exec\n") : "+s"(m) : : "scc");
        if (KIND == 10) asm volatile(RT_REP32("v_div_scale_f64 %0, vcc, %1, %2, %1\n") : "+v"(d) : "v"(b), "v"(c) : "vcc");
        if (KIND == 11) asm volatile(RT_REP32("v_div_fixup_f64 %0, %1, %2, %1\n") : "+v"(d) : "v"(b), "v"(c));
        if (KIND == 12) /* the rectangle test's mix: one compare into a scalar pair and the scalar AND that uses it */
            asm volatile(RT_REP32("v_cmp_lt_f64 %0, %1, %2\ns_and_b64 %0, %0, exec\n") : "=s"(m) : "v"(b), "v"(c) : "scc");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (d == 12345.678 || fa == 1.5f || ia == -77 || m == 0x1234567ull) out[2] = 1; /* keep the results alive */
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], t1 - t0);
        atomicAdd(&out[1], 1ull);
    }
}

__global__ void __launch_bounds__(RTR_BLOCK) k_stream8(const double* __restrict__ in, double* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * RTR_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * RTR_BLOCK)
        out[i] = in[i] + 1.0;
}
#endif /* RTR_TU_CAPI */
