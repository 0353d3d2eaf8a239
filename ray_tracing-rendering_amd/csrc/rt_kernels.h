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
    if (rt_is_program(trav)) return RTR_PROGRAM_WAVES;
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
    return (integ == RTR_INTEGRATOR_RR || integ == RTR_INTEGRATOR_PATH || rt_is_program(trav) || trav == RT_TRAV_MEDIA)
               ? 12
               : RT_PARK_WORDS;
}

/* ---- material-sorted shading inside the workgroup ---------------------------------------------------------------
 * The variant for "every material, QuadLights, flat compiled scene" (scene 23's kernel) regroups the 256 paths of the
 * tile by the material class of their hit before it shades them.  Without it a wave runs the code of every class
 * that one of its lanes hit, one after the other, with the lanes whose ray left the scene idle all the while (half of
 * them on scene 23).  Per bounce: the lane that OWNS a path casts its ray, adds the emission term (it needs the ray
 * that hit the emitter) and takes a ticket in its class (one LDS atomic per wave and class); after a barrier every lane
 * knows the class totals, classes are laid out in wave-aligned ranges where 256 slots allow it, the owner writes
 * (hit point, normal, incoming direction, generator state, material) into its slot; after a second barrier lane i
 * SHADES slot i -- light sample, BSDF sample, roulette: waves past the last item skip shading altogether -- and writes
 * throughput / pdf / pending light contribution straight into the owner's parked words and the new direction into the
 * slot; after a third barrier the owner takes them back.  A path's generator state travels with it and every sum is
 * formed by the same operands in the same order, so the image is the unsorted kernel's bit for bit
 * (test_sorted_shading_equals_unsorted).  MEASURED ON SCENE 23: 4 063 against 6 617 Msamples/s unsorted -- the three
 * barriers cost 16 % of the wave cycles, tickets and exchange 7 %, the owner's emission pass and the second material
 * fetch another 6 %, and with four classes in four wave-aligned ranges every wave still shades (one class each):
 * profiles/r03_sorted_regions.txt.  It is therefore NOT what RTR_PIPELINE_AUTO runs; RTR_FLAG_SORTED_SHADING asks
 * for it. */
/* which (integrator, traversal, material set) has a sorted instantiation: RTR_FLAG_SORTED_SHADING selects it */
constexpr bool mega_sortable(int integ, int trav, int ms) {
    return integ == RTR_INTEGRATOR_MIS && trav == RT_TRAV_FLAT && ms == RT_MS_QUADLIT;
}
/* parked words of the sorted variant (the pixel sum lives in the partial-sum buffer itself) + the exchange slot */
enum { SK_THR = 0, SK_L = 3, SK_PDF = 6, SK_NCLOSEST = 7, SK_NSHADOW = 8, SK_CONTRIB = 9, SK_SWI = 12, SK_STMAX = 15, SK_X = 16,
       SK_WORDS = 26 };
enum { SX_P = 0, SX_N = 3, SX_RD = 6, SX_PACK = 9 }; /* slot words: hit point, normal, direction (in: incoming, out: next), packed */
RT_DEV int material_class(int type) { /* what shades alike */
    return type == RTR_MAT_LAMBERTIAN ? 0 : (type == RTR_MAT_PBR ? 1 : ((type == RTR_MAT_DIELECTRIC || type == RTR_MAT_METAL) ? 2 : 3));
}

template <int INTEG, int TRAV, int MS, bool SORT = false>
__global__ void __launch_bounds__(RTR_BLOCK, mega_waves(INTEG, TRAV, MS))
    k_mega(const DScene* __restrict__ scp, const RenderK P, const int stack_words) {
    static_assert(!SORT || mega_sortable(INTEG, TRAV, MS), "no sorted variant of this kernel");
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
    if (!SORT) pk.set3(PK_ACC, mk(0, 0, 0));
    PathCounters cnt;
    cnt.closest = 0, cnt.shadow = 0;
    uint32_t n_samples = 0;
    PathState ps;
    uint32_t rng = 1;
    bool fresh = true;
    bool done = !active || s >= s_end;
    if (SORT) {
        /* the class counters live in the traversal-stack words, which a flat scan never touches: one static word more
         * would round the workgroup's LDS up to the next 512 bytes and cost the third workgroup per CU */
        int* const s_cnt = lds_stack;
        const double* const xin = pk.base - threadIdx.x + SK_X * RTR_BLOCK; /* slot q, word k: xin[k * RTR_BLOCK + q] */
        double* const xw = const_cast<double*>(xin);
        double* const accp = P.partial + (size_t)cell * 3 * RTR_BLOCK + threadIdx.x; /* this lane's pixel sum */
        accp[0] = 0.0, accp[RTR_BLOCK] = 0.0, accp[2 * RTR_BLOCK] = 0.0;
        if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
        auto begin_sample = [&]() { /* renderer.h:73-75 under the per-sample seed */
            int pi, pj;
            bool in_region;
            tile_pixel(P, slot, threadIdx.x, pi, pj, in_region);
            rng = rtr_sample_seed_inline(P.seed, P.W, pi, pj, s);
            const Real u = (pi + rng_next(rng)) / (P.W - 1);
            const Real v = (pj + rng_next(rng)) / (P.H - 1);
            camera_get_ray(sc.camera, u, v, rng, ps.ro, ps.rd, ps.tm);
            ps.depth = 0, ps.specular_bounce = false;
            pk.set3(SK_THR, mk(1.0, 1.0, 1.0));
            pk.set3(SK_L, mk(0.0, 0.0, 0.0));
            pk.set(SK_PDF, 0.0);
        };
        pk.set(SK_NCLOSEST, 0.0);
        pk.set(SK_NSHADOW, 0.0);
        if (!done) begin_sample();
        __syncthreads();
        for (;;) {
            bool pending = false, ended = false, shade_me = false;
            int cls = 0, ticket = 0;
            Hit rec;
            rec.u = 0, rec.v = 0, rec.mat = 0, rec.front = false;
            RT_REGION(RG_OTHER);
            if (!done) {
                pk.set(SK_NCLOSEST, pk.get(SK_NCLOSEST) + 1.0);
                const bool hit_any = cast_closest<TRAV, false>(sc, ps.ro, ps.rd, ps.tm, rec, rng, st);
                if (!hit_any) {
                    RT_REGION(RG_MISS);
                    pk.set3(SK_L, add(pk.get3(SK_L), miss_radiance<INTEG, MS>(sc, pk.get3(SK_THR), ps.ro, ps.rd, ps.depth,
                                                                          ps.specular_bounce, pk.get(SK_PDF))));
                    ended = true;
                } else { /* the emission term of mis_path_integrator.h:72-94 needs the ray that hit: it stays with the owner */
                    ps.thr = pk.get3(SK_THR);
                    ps.L = mk(0.0, 0.0, 0.0);
                    ps.prev_bsdf_pdf = pk.get(SK_PDF);
                    const MatCtx mc = mat_prepare<MS>(sc, rec);
                    ShadowReq none;
                    shade_a_mis<MS, INTEG, 1>(sc, ps, rec, mc, mk(0, 0, 0), rng, none);
                    if (ps.L.x != 0.0 || ps.L.y != 0.0 || ps.L.z != 0.0) pk.set3(SK_L, add(pk.get3(SK_L), ps.L));
                    shade_me = true;
                    cls = material_class(mc.type);
                }
            }
            /* tickets: one LDS atomic per wave and class, ranks inside the wave by counting lanes below */
            RT_REGION(RG_EXCHANGE);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long long m = __ballot(shade_me && cls == k);
                if (m) { /* wave-uniform */
                    int base = 0;
                    if ((int)__lane_id() == __builtin_ctzll(m)) base = atomicAdd(&s_cnt[k], __builtin_popcountll(m));
                    base = __shfl(base, __builtin_ctzll(m), 64);
                    if (shade_me && cls == k) ticket = base + (int)__builtin_popcountll(m & ((1ull << __lane_id()) - 1ull));
                }
            }
            RT_REGION(RG_BARRIER);
            if (!__syncthreads_or(!done)) break; /* barrier 1: every ticket is taken; nothing left to do = leave together */
            RT_REGION(RG_EXCHANGE);
            const int c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2], c3 = s_cnt[3];
            int b0 = 0, b1 = (c0 + 63) & ~63, b2 = b1 + ((c1 + 63) & ~63), b3 = b2 + ((c2 + 63) & ~63);
            if (b3 + c3 > RTR_BLOCK) b1 = c0, b2 = c0 + c1, b3 = c0 + c1 + c2; /* no room for wave-aligned classes: packed */
            const int pos = (cls == 0 ? b0 : (cls == 1 ? b1 : (cls == 2 ? b2 : b3))) + ticket;
            if (shade_me) {
                xw[(SX_P + 0) * RTR_BLOCK + pos] = rec.p.x, xw[(SX_P + 1) * RTR_BLOCK + pos] = rec.p.y, xw[(SX_P + 2) * RTR_BLOCK + pos] = rec.p.z;
                xw[(SX_N + 0) * RTR_BLOCK + pos] = rec.n.x, xw[(SX_N + 1) * RTR_BLOCK + pos] = rec.n.y, xw[(SX_N + 2) * RTR_BLOCK + pos] = rec.n.z;
                xw[(SX_RD + 0) * RTR_BLOCK + pos] = ps.rd.x, xw[(SX_RD + 1) * RTR_BLOCK + pos] = ps.rd.y, xw[(SX_RD + 2) * RTR_BLOCK + pos] = ps.rd.z;
                const unsigned long long pack = (unsigned long long)rng | ((unsigned long long)(rec.mat & 0xffff) << 32) |
                                                ((unsigned long long)threadIdx.x << 48) | ((unsigned long long)(rec.front ? 1 : 0) << 56) |
                                                ((unsigned long long)(ps.depth >= P.rr_start ? 1 : 0) << 57);
                xw[SX_PACK * RTR_BLOCK + pos] = __longlong_as_double((long long)pack);
            }
            RT_REGION(RG_BARRIER);
            __syncthreads(); /* barrier 2: the slots are filled */
            RT_REGION(RG_EXCHANGE);
            if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0; /* everyone has read the totals; the next tickets come after barrier 3 */
            {
                const int q = threadIdx.x;
                const bool mine = (q >= b0 && q < b0 + c0) || (q >= b1 && q < b1 + c1) || (q >= b2 && q < b2 + c2) || (q >= b3 && q < b3 + c3);
                if (mine) { /* shade slot q: mis_path_integrator.h:96-146 */
                    const unsigned long long pack = (unsigned long long)__double_as_longlong(xin[SX_PACK * RTR_BLOCK + q]);
                    Hit h;
                    h.u = 0, h.v = 0, h.t = 0;
                    h.p = mk(xin[(SX_P + 0) * RTR_BLOCK + q], xin[(SX_P + 1) * RTR_BLOCK + q], xin[(SX_P + 2) * RTR_BLOCK + q]);
                    h.n = mk(xin[(SX_N + 0) * RTR_BLOCK + q], xin[(SX_N + 1) * RTR_BLOCK + q], xin[(SX_N + 2) * RTR_BLOCK + q]);
                    h.mat = (int)((pack >> 32) & 0xffff);
                    h.front = ((pack >> 56) & 1) != 0;
                    const int owner = (int)((pack >> 48) & 0xff);
                    const Park po{pk.base - threadIdx.x + owner}; /* the owner's parked words */
                    PathState sh;
                    sh.rd = mk(xin[(SX_RD + 0) * RTR_BLOCK + q], xin[(SX_RD + 1) * RTR_BLOCK + q], xin[(SX_RD + 2) * RTR_BLOCK + q]);
                    sh.ro = h.p, sh.tm = 0;
                    sh.thr = po.get3(SK_THR);
                    sh.L = mk(0.0, 0.0, 0.0);
                    sh.prev_bsdf_pdf = 0.0;
                    sh.depth = ((pack >> 57) & 1) ? P.rr_start : P.rr_start - 1; /* only `depth >= rr_start` is read */
                    sh.specular_bounce = false;
                    uint32_t r2 = (uint32_t)pack;
                    const V3 wo = neg(unit(sh.rd));
                    ShadowReq rq;
                    const MatCtx mc = mat_prepare<MS>(sc, h);
                    shade_a_mis<MS, INTEG, 2>(sc, sh, h, mc, wo, r2, rq);
                    if (rq.valid) {
                        po.set3(SK_SWI, rq.wi);
                        po.set(SK_STMAX, rq.tmax);
                        po.set3(SK_CONTRIB, rq.contrib);
                    }
                    const bool go = shade_b_mis<MS, INTEG>(sc, sh, h, mc, wo, r2, P.rr_start);
                    po.set3(SK_THR, sh.thr);
                    po.set(SK_PDF, sh.prev_bsdf_pdf);
                    xw[(SX_RD + 0) * RTR_BLOCK + q] = sh.rd.x, xw[(SX_RD + 1) * RTR_BLOCK + q] = sh.rd.y, xw[(SX_RD + 2) * RTR_BLOCK + q] = sh.rd.z;
                    const unsigned long long back = (unsigned long long)r2 | ((unsigned long long)(go ? 1 : 0) << 32) |
                                                    ((unsigned long long)(sh.specular_bounce ? 1 : 0) << 33) |
                                                    ((unsigned long long)(rq.valid ? 1 : 0) << 34);
                    xw[SX_PACK * RTR_BLOCK + q] = __longlong_as_double((long long)back);
                }
            }
            RT_REGION(RG_BARRIER);
            __syncthreads(); /* barrier 3: results are back */
            RT_REGION(RG_PARK);
            if (shade_me) {
                const unsigned long long back = (unsigned long long)__double_as_longlong(xin[SX_PACK * RTR_BLOCK + pos]);
                rng = (uint32_t)back;
                ps.ro = mk(xin[(SX_P + 0) * RTR_BLOCK + pos], xin[(SX_P + 1) * RTR_BLOCK + pos], xin[(SX_P + 2) * RTR_BLOCK + pos]);
                ps.rd = mk(xin[(SX_RD + 0) * RTR_BLOCK + pos], xin[(SX_RD + 1) * RTR_BLOCK + pos], xin[(SX_RD + 2) * RTR_BLOCK + pos]);
                ps.specular_bounce = ((back >> 33) & 1) != 0;
                pending = ((back >> 34) & 1) != 0;
                const bool go = ((back >> 32) & 1) != 0;
                ended = !go || ++ps.depth >= P.max_depth;
            }
            RT_REGION(RG_OTHER);
            if (pending) { /* mis_path_integrator.h:210-213, origin = the hit point = ps.ro */
                pk.set(SK_NSHADOW, pk.get(SK_NSHADOW) + 1.0);
                if (!cast_shadow<TRAV>(sc, ps.ro, pk.get3(SK_SWI), pk.get(SK_STMAX), rng, st))
                    pk.set3(SK_L, add(pk.get3(SK_L), pk.get3(SK_CONTRIB)));
            }
            RT_REGION(RG_REGEN);
            if (ended) {
                const V3 L = pk.get3(SK_L);
                accp[0] += L.x, accp[RTR_BLOCK] += L.y, accp[2 * RTR_BLOCK] += L.z; /* renderer.h:77-78 */
                ++n_samples;
                ++s;
                done = s >= s_end || ((s & 7) == 0 && render_cancelled(P));
                if (!done) begin_sample();
            }
        }
        cnt.closest = (uint32_t)pk.get(SK_NCLOSEST);
        cnt.shadow = (uint32_t)pk.get(SK_NSHADOW);
    } else if (TRAV == RT_TRAV_MEDIA) {
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
                        if (rt_is_program(TRAV)) {
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
    if (!SORT) { /* (the sorted variant summed into the buffer itself) */
        const V3 acc = pk.get3(PK_ACC);
        double* out = P.partial + (size_t)cell * 3 * RTR_BLOCK + threadIdx.x;
        out[0] = acc.x;
        out[RTR_BLOCK] = acc.y;
        out[2 * RTR_BLOCK] = acc.z;
    }
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
