/*
 * rt_wavefront.h -- the wavefront pipeline: the integrator loop of
 * renderer/mis_path_integrator.h:25-150 (and rr_path_integrator.h:21-59) cut at its two
 * ray casts into stage kernels that exchange path state through SoA arrays in HBM.
 *
 * Path pool.  Slot = (owned tile, spp chunk, pixel of the tile): the same numbering as the
 * megakernel's (blockIdx, threadIdx).  A slot runs the samples of its chunk back to back
 * (in-slot regeneration), so per-pixel sums keep the sample order of renderer.h:72-79 and
 * the pool stays full until slots run out of samples.  Every array below is indexed by
 * slot, so consecutive lanes touch consecutive 8-byte words (coalesced 512 B per wave).
 *
 * One iteration (one bounce of every live path):
 *   wf_extend   flush a finished sample into the pixel sum, start the next camera sample
 *               (renderer.h:73-75), cast the closest-hit ray (mis_path_integrator.h:37),
 *               handle the miss, else store the hit and append the slot to the queue of its
 *               material type  -> material-sorted shading waves.
 *   wf_shade    over the concatenated material queues: emission + MIS weight, light sample
 *               (:72-103,191-229), BSDF sample, throughput, Russian roulette (:105-146);
 *               writes the next ray and, if the light sample is usable, a shadow request.
 *   wf_connect  slots whose shade stage left a shadow request: occlusion ray (:210-213);
 *               unoccluded -> L += contrib.
 * Scenes with participating media draw random numbers INSIDE both ray casts
 * (constant_medium.h:85), so there the shade stage is split around the connect stage
 * (wf_shade_a, wf_connect, wf_shade_b) to keep the reference's draw order.
 *
 * Material queues are built only when a scene mixes material classes (otherwise every wave is
 * already uniform and the shade stage walks the slots directly).  Their counters are
 * double-buffered by iteration parity: wf_extend of iteration i clears the counters of parity
 * (i+1)&1, which no kernel of iteration i touches.  Cast counts are kept per slot and summed once
 * per render: nothing in the iteration touches a shared address except one block-aggregated
 * atomic per queue.
 */
#pragma once

#include "rt_kernels.h"
#include "rtr_hip_test.h"

#include <atomic>
#include <string>

#define WF_NTYPES RTR_MAT_TYPE_COUNT

enum { WF_DONE = 0, WF_NEED_SAMPLE = 1, WF_HIT = 2, WF_CONTINUE = 3 };
/* flags word: bits 0-1 status, bit 2 specular_bounce, bit 3 "no sample yet", bit 4 shadow request
 * pending, bits 8.. depth */
#define WF_STATUS(f) ((f) & 3)
#define WF_SPEC 4
#define WF_FIRST 8
#define WF_SHADOW 16

struct WfState {
    /* ray */
    double *ox, *oy, *oz, *dx, *dy, *dz, *tm;
    /* path */
    double *tx, *ty, *tz, *lx, *ly, *lz, *ax, *ay, *az, *pdf;
    uint32_t* rng;
    int32_t* samp;
    int32_t* flags;
    /* hit record */
    double *ht, *hpx, *hpy, *hpz, *hnx, *hny, *hnz, *hu, *hv;
    int32_t* hmat; /* material | front_face << 30 */
    /* shadow request */
    double *swx, *swy, *swz, *stmax, *scx, *scy, *scz;
    /* per-slot cast counters (summed once at the end: no per-iteration atomics) */
    uint32_t *n_closest, *n_shadow;
    /* material queues (only for scenes with several material classes) */
    int32_t* q_mat;     /* [WF_NTYPES][n_slots] */
    uint32_t* counters; /* [2][WF_NTYPES + 2]: per parity: material queue sizes */
    uint32_t* n_live;   /* slots not yet WF_DONE */
    int n_slots;
};
#define WF_CNT_STRIDE (WF_NTYPES + 2)

struct WavefrontPool {
    void* slab = nullptr;
    size_t slab_bytes = 0;
    uint32_t* h_live = nullptr; /* pinned */
    void release() {
        if (slab) (void)hipFree(slab);
        if (h_live) (void)hipHostFree(h_live);
        slab = nullptr, slab_bytes = 0, h_live = nullptr;
    }
};

RT_DEV V3 ldv(const double* x, const double* y, const double* z, int i) { return mk(x[i], y[i], z[i]); }
RT_DEV void stv(double* x, double* y, double* z, int i, V3 v) { x[i] = v.x, y[i] = v.y, z[i] = v.z; }

/* slot -> pixel (same layout as k_mega) */
RT_DEV void slot_pixel(const RenderK& P, int slot, int& i, int& j, int& chunk, bool& active) {
    const int blk = slot / RTR_BLOCK, tid = slot % RTR_BLOCK;
    chunk = blk % P.chunks;
    tile_pixel(P, blk / P.chunks, tid, i, j, active);
}

/* Append this block's items to global queues with ONE global atomic per queue per block:
 * lanes take a position inside the block with an LDS atomic, a leader reserves the block's
 * range, then every lane writes its slot.  (One global atomic per lane on a single counter
 * serialises at ~11 ns each on MI355X and dominated the first version of this pipeline.)
 * Must be called by all threads of the block; `q` < 0 means "nothing to append". */
template <int NQ>
RT_DEV void wf_block_append(uint32_t* lds_cnt /* [2*NQ] */, uint32_t* gcnt, int32_t* qbase, size_t qstride, int q,
                            int slot) {
    if (threadIdx.x < 2 * NQ) lds_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t pos = 0;
    if (q >= 0) pos = atomicAdd(&lds_cnt[q], 1u);
    __syncthreads();
    if (threadIdx.x < NQ) {
        const uint32_t c = lds_cnt[threadIdx.x];
        if (c) lds_cnt[NQ + threadIdx.x] = atomicAdd(&gcnt[threadIdx.x], c);
    }
    __syncthreads();
    if (q >= 0) qbase[(size_t)q * qstride + lds_cnt[NQ + q] + pos] = slot;
}

__global__ void __launch_bounds__(RTR_BLOCK) wf_init(const WfState S, const RenderK P) {
    const int slot = blockIdx.x * RTR_BLOCK + threadIdx.x;
    if (slot >= S.n_slots) return;
    int i, j, chunk;
    bool active;
    slot_pixel(P, slot, i, j, chunk, active);
    const int s0 = (int)((long long)chunk * P.spp / P.chunks);
    const int s1 = (int)((long long)(chunk + 1) * P.spp / P.chunks);
    active = active && s0 < s1;
    S.ax[slot] = 0, S.ay[slot] = 0, S.az[slot] = 0;
    S.lx[slot] = 0, S.ly[slot] = 0, S.lz[slot] = 0;
    S.samp[slot] = s0 - 1;
    S.n_closest[slot] = 0, S.n_shadow[slot] = 0;
    S.flags[slot] = active ? (WF_NEED_SAMPLE | WF_FIRST) : WF_DONE;
    if (!active) { /* pixels outside the region still own a partial-sum cell */
        double* out = P.partial + (size_t)(slot / RTR_BLOCK) * 3 * RTR_BLOCK + (slot % RTR_BLOCK);
        out[0] = 0, out[RTR_BLOCK] = 0, out[2 * RTR_BLOCK] = 0;
    }
    if (threadIdx.x == 0) P.done[slot / RTR_BLOCK] = 1; /* k_resolve runs only after every slot has finished */
    const unsigned long long live = wave_sum(active ? 1ull : 0ull);
    if ((threadIdx.x & 63) == 0 && live) atomicAdd(S.n_live, (uint32_t)live);
    if (slot < 2 * WF_CNT_STRIDE) S.counters[slot] = 0;
}

/* once per render: fold the per-slot counters into the render statistics */
__global__ void __launch_bounds__(RTR_BLOCK) wf_finish(const WfState S, const RenderK P) {
    const int slot = blockIdx.x * RTR_BLOCK + threadIdx.x;
    unsigned long long a = 0, b = 0, c = 0;
    if (slot < S.n_slots) {
        int i, j, chunk;
        bool active;
        slot_pixel(P, slot, i, j, chunk, active);
        const int s0 = (int)((long long)chunk * P.spp / P.chunks);
        if (active) a = (unsigned long long)(S.samp[slot] - s0); /* samp ends at s_end: finished samples */
        b = S.n_closest[slot], c = S.n_shadow[slot];
    }
    a = wave_sum(a), b = wave_sum(b), c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) {
        if (a) atomicAdd(&P.stats[0], a);
        if (b) atomicAdd(&P.stats[1], b);
        if (c) atomicAdd(&P.stats[2], c);
    }
}

/* RICH = false: scenes lit by QuadLights only whose textures read no (u,v) (rtr_upload_scene
 * decides): no environment-light code on the miss branch, no (u,v) reconstruction */
template <int TRAV, bool SORT, bool RICH>
__global__ void __launch_bounds__(RTR_BLOCK, 4) wf_extend(const DScene* __restrict__ scp, const WfState S,
                                                          const RenderK P, const int parity) {
    extern __shared__ int lds_stack[];
    const DScene& sc = *scp;
    const Stack st{lds_stack + threadIdx.x};
    if (SORT && blockIdx.x == 0 && threadIdx.x < WF_CNT_STRIDE)
        S.counters[(parity ^ 1) * WF_CNT_STRIDE + threadIdx.x] = 0;
    uint32_t* cnt = S.counters + parity * WF_CNT_STRIDE;
    unsigned n_done = 0;
    __shared__ uint32_t lds_cnt[2 * WF_NTYPES];
    for (int base = blockIdx.x * RTR_BLOCK; base < S.n_slots; base += gridDim.x * RTR_BLOCK) {
        const int slot = base + threadIdx.x;
        int qtype = -1;
        do {
            if (slot >= S.n_slots) break;
            int flags = S.flags[slot];
            const int status = WF_STATUS(flags);
            if (status == WF_DONE) break;
            V3 ro, rd;
            Real tm;
            uint32_t rng;
            if (status == WF_NEED_SAMPLE) {
                int i, j, chunk;
                bool active;
                slot_pixel(P, slot, i, j, chunk, active);
                V3 acc = ldv(S.ax, S.ay, S.az, slot);
                if (!(flags & WF_FIRST)) acc = add(acc, ldv(S.lx, S.ly, S.lz, slot)); /* renderer.h:77-78 */
                const int s = S.samp[slot] + 1;
                S.samp[slot] = s;
                const int s_end = (int)((long long)(chunk + 1) * P.spp / P.chunks);
                if (s >= s_end) {
                    S.flags[slot] = WF_DONE;
                    double* out = P.partial + (size_t)(slot / RTR_BLOCK) * 3 * RTR_BLOCK + (slot % RTR_BLOCK);
                    out[0] = acc.x, out[RTR_BLOCK] = acc.y, out[2 * RTR_BLOCK] = acc.z;
                    ++n_done;
                    break;
                }
                stv(S.ax, S.ay, S.az, slot, acc);
                rng = rtr_sample_seed_inline(P.seed, P.W, i, j, s);
                const Real u = (i + rng_next(rng)) / (P.W - 1);
                const Real v = (j + rng_next(rng)) / (P.H - 1);
                camera_get_ray(sc.camera, u, v, rng, ro, rd, tm);
                stv(S.ox, S.oy, S.oz, slot, ro);
                stv(S.dx, S.dy, S.dz, slot, rd);
                S.tm[slot] = tm;
                stv(S.tx, S.ty, S.tz, slot, mk(1.0, 1.0, 1.0));
                stv(S.lx, S.ly, S.lz, slot, mk(0.0, 0.0, 0.0));
                S.pdf[slot] = 0.0;
                flags = 0; /* depth 0, not specular */
            } else {
                ro = ldv(S.ox, S.oy, S.oz, slot);
                rd = ldv(S.dx, S.dy, S.dz, slot);
                tm = S.tm[slot];
                rng = S.rng[slot];
                flags &= ~3;
            }
            Hit rec;
            rec.u = 0, rec.v = 0;
            S.n_closest[slot] += 1;
            if (!cast_closest<TRAV, RICH>(sc, ro, rd, tm, rec, rng, st)) {
                /* mis_path_integrator.h:37-67, rr_path_integrator.h:31-33 */
                const V3 thr = ldv(S.tx, S.ty, S.tz, slot);
                V3 add_l;
                if (P.integrator == RTR_INTEGRATOR_MIS)
                    add_l = miss_radiance<RTR_INTEGRATOR_MIS, RICH ? RT_MS_FULL : RT_MS_QUADLIT>(sc, thr, ro, rd, flags >> 8, (flags & WF_SPEC) != 0,
                                                              S.pdf[slot]);
                else
                    add_l = mul(thr, ld3(sc.background));
                stv(S.lx, S.ly, S.lz, slot, add(ldv(S.lx, S.ly, S.lz, slot), add_l));
                S.flags[slot] = flags | WF_NEED_SAMPLE;
                S.rng[slot] = rng;
                break;
            }
            S.ht[slot] = rec.t;
            stv(S.hpx, S.hpy, S.hpz, slot, rec.p);
            stv(S.hnx, S.hny, S.hnz, slot, rec.n);
            if (sc.needs_uv) S.hu[slot] = rec.u, S.hv[slot] = rec.v;
            S.hmat[slot] = rec.mat | (rec.front ? (1 << 30) : 0);
            S.flags[slot] = flags | WF_HIT;
            S.rng[slot] = rng;
            if (SORT) qtype = as_const(sc.materials)[rec.mat].type;
        } while (false);
        if (SORT) wf_block_append<WF_NTYPES>(lds_cnt, cnt, S.q_mat, (size_t)S.n_slots, qtype, slot);
    }
    const unsigned long long d = wave_sum(n_done);
    if ((threadIdx.x & 63) == 0 && d) atomicSub(S.n_live, (uint32_t)d);
}

/* k-th entry of the concatenated material queues */
RT_DEV int wf_sorted_slot(const WfState& S, const uint32_t* cnt, uint32_t k) {
#pragma unroll
    for (int t = 0; t < WF_NTYPES; ++t) {
        const uint32_t c = cnt[t];
        if (k < c) return S.q_mat[(size_t)t * S.n_slots + k];
        k -= c;
    }
    return -1;
}

RT_DEV void wf_load_hit(const DScene& sc, const WfState& S, int slot, Hit& rec) {
    rec.t = S.ht[slot];
    rec.p = ldv(S.hpx, S.hpy, S.hpz, slot);
    rec.n = ldv(S.hnx, S.hny, S.hnz, slot);
    rec.u = 0, rec.v = 0;
    if (sc.needs_uv) rec.u = S.hu[slot], rec.v = S.hv[slot];
    const int m = S.hmat[slot];
    rec.mat = m & ~(1 << 30);
    rec.front = (m >> 30) & 1;
}

RT_DEV void wf_load_path(const WfState& S, int slot, int flags, PathState& ps) {
    ps.ro = ldv(S.ox, S.oy, S.oz, slot);
    ps.rd = ldv(S.dx, S.dy, S.dz, slot);
    ps.tm = S.tm[slot];
    ps.thr = ldv(S.tx, S.ty, S.tz, slot);
    ps.L = ldv(S.lx, S.ly, S.lz, slot);
    ps.prev_bsdf_pdf = S.pdf[slot];
    ps.depth = flags >> 8;
    ps.specular_bounce = (flags & WF_SPEC) != 0;
}

/* after shade_b / shade_rr: store the continued path or mark the sample finished */
RT_DEV void wf_store_path(const WfState& S, int slot, const PathState& ps, bool go, int max_depth, int extra_flags) {
    int depth = ps.depth;
    if (go) {
        stv(S.ox, S.oy, S.oz, slot, ps.ro);
        stv(S.dx, S.dy, S.dz, slot, ps.rd);
        stv(S.tx, S.ty, S.tz, slot, ps.thr);
        S.pdf[slot] = ps.prev_bsdf_pdf;
        go = ++depth < max_depth;
    }
    S.flags[slot] =
        (go ? WF_CONTINUE : WF_NEED_SAMPLE) | (ps.specular_bounce ? WF_SPEC : 0) | (depth << 8) | extra_flags;
}

RT_DEV void wf_store_shadow(const WfState& S, int slot, const ShadowReq& rq) {
    stv(S.swx, S.swy, S.swz, slot, rq.wi);
    S.stmax[slot] = rq.tmax;
    stv(S.scx, S.scy, S.scz, slot, rq.contrib);
}

/* PHASE 0: whole shading (no media).  PHASE 1: first half only (emission + light sample).
 * PHASE 2: second half only (BSDF sample + roulette), after the connect stage.
 * SORT: walk the material-sorted queues; otherwise walk the slots and take those that hit. */
template <int INTEG, int PHASE, int MS, bool SORT>
__global__ void __launch_bounds__(RTR_BLOCK, MS == RT_MS_LEAN ? 4 : 2) wf_shade(const DScene* __restrict__ scp,
                                                                                const WfState S, const RenderK P,
                                                                                const int parity) {
    const DScene& sc = *scp;
    const uint32_t* cnt = S.counters + parity * WF_CNT_STRIDE;
    uint32_t total = (uint32_t)S.n_slots;
    if (SORT) {
        total = 0;
#pragma unroll
        for (int t = 0; t < WF_NTYPES; ++t) total += cnt[t];
    }
    for (uint32_t k = blockIdx.x * RTR_BLOCK + threadIdx.x; k < total; k += gridDim.x * RTR_BLOCK) {
        const int slot = SORT ? wf_sorted_slot(S, cnt, k) : (int)k;
        const int flags = S.flags[slot];
        if (!SORT && WF_STATUS(flags) != WF_HIT) continue;
        Hit rec;
        wf_load_hit(sc, S, slot, rec);
        PathState ps;
        wf_load_path(S, slot, flags, ps);
        uint32_t rng = S.rng[slot];
        bool go;
        int extra = flags & WF_SHADOW; /* phase 2 keeps what phase 1 requested */
        if (INTEG == RTR_INTEGRATOR_MIS) {
            const V3 wo = neg(unit(ps.rd));
            const MatCtx mc = mat_prepare<MS>(sc, rec);
            if (PHASE != 2) {
                const V3 L0 = ps.L;
                ShadowReq rq;
                shade_a_mis<MS>(sc, ps, rec, mc, wo, rng, rq);
                if (ps.L.x != L0.x || ps.L.y != L0.y || ps.L.z != L0.z) stv(S.lx, S.ly, S.lz, slot, ps.L);
                extra = 0;
                if (rq.valid) {
                    wf_store_shadow(S, slot, rq);
                    extra = WF_SHADOW;
                }
            }
            if (PHASE == 1) {
                S.rng[slot] = rng;
                if (extra) S.flags[slot] = flags | WF_SHADOW;
                continue;
            }
            go = shade_b_mis<MS>(sc, ps, rec, mc, wo, rng, P.rr_start);
        } else {
            const V3 L0 = ps.L;
            go = shade_rr<MS>(sc, ps, rec, rng, P.rr_start);
            if (ps.L.x != L0.x || ps.L.y != L0.y || ps.L.z != L0.z) stv(S.lx, S.ly, S.lz, slot, ps.L);
            extra = 0;
        }
        wf_store_path(S, slot, ps, go, P.max_depth, PHASE == 2 ? 0 : extra);
        S.rng[slot] = rng;
    }
}

/* occlusion rays of the pending light connections (mis_path_integrator.h:210-213) */
template <int TRAV>
__global__ void __launch_bounds__(RTR_BLOCK, 4) wf_connect(const DScene* __restrict__ scp, const WfState S,
                                                           const RenderK P) {
    extern __shared__ int lds_stack[];
    const DScene& sc = *scp;
    const Stack st{lds_stack + threadIdx.x};
    for (int slot = blockIdx.x * RTR_BLOCK + threadIdx.x; slot < S.n_slots; slot += gridDim.x * RTR_BLOCK) {
        const int flags = S.flags[slot];
        if (!(flags & WF_SHADOW)) continue;
        S.flags[slot] = flags & ~WF_SHADOW;
        /* shadow_ray origin = rec.p (:210); the hit record outlives the shade stage */
        const V3 o = ldv(S.hpx, S.hpy, S.hpz, slot);
        const V3 wi = ldv(S.swx, S.swy, S.swz, slot);
        const bool MEDIA = TRAV == RT_TRAV_MEDIA || TRAV == RT_TRAV_PROGRAM;
        uint32_t rng = MEDIA ? S.rng[slot] : 1u;
        S.n_shadow[slot] += 1;
        const bool hit = cast_shadow<TRAV>(sc, o, wi, S.stmax[slot], rng, st);
        if (MEDIA) S.rng[slot] = rng;
        if (!hit) stv(S.lx, S.ly, S.lz, slot, add(ldv(S.lx, S.ly, S.lz, slot), ldv(S.scx, S.scy, S.scz, slot)));
    }
}

/* ---- host driver ------------------------------------------------------------------------------ */
inline int wf_fail(std::string& err, int code, const std::string& m) {
    err = m;
    return code;
}
#define WF_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return wf_fail(err, RTR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

inline int wf_alloc(WavefrontPool& pool, WfState& S, int n_slots, bool sort, std::string& err) {
    const size_t n = (size_t)n_slots;
    const size_t n_f64 = 33; /* ray 7, path 10, hit 9, shadow 7 */
    const size_t bytes = n_f64 * n * 8 + 6 * n * 4 /* rng samp flags hmat n_closest n_shadow */ +
                         (sort ? (size_t)WF_NTYPES * n * 4 : 0) + (2 * WF_CNT_STRIDE + 2) * 4 + 256;
    if (pool.slab_bytes < bytes) {
        if (pool.slab) WF_HIP(hipFree(pool.slab));
        pool.slab = nullptr, pool.slab_bytes = 0;
        hipError_t e = hipMalloc(&pool.slab, bytes);
        if (e != hipSuccess) return wf_fail(err, RTR_ERR_NOMEM, std::string("hipMalloc(path pool): ") + hipGetErrorString(e));
        pool.slab_bytes = bytes;
    }
    if (!pool.h_live) WF_HIP(hipHostMalloc(reinterpret_cast<void**>(&pool.h_live), 64, hipHostMallocDefault));
    char* p = static_cast<char*>(pool.slab);
    auto f64 = [&]() {
        double* r = reinterpret_cast<double*>(p);
        p += n * 8;
        return r;
    };
    double** arrs[] = {&S.ox, &S.oy, &S.oz, &S.dx, &S.dy, &S.dz, &S.tm, &S.tx, &S.ty, &S.tz, &S.lx,
                       &S.ly, &S.lz, &S.ax, &S.ay, &S.az, &S.pdf, &S.ht, &S.hpx, &S.hpy, &S.hpz, &S.hnx,
                       &S.hny, &S.hnz, &S.hu, &S.hv, &S.swx, &S.swy, &S.swz, &S.stmax, &S.scx, &S.scy, &S.scz};
    static_assert(sizeof(arrs) / sizeof(arrs[0]) == 33, "array count");
    for (double** a : arrs) *a = f64();
    auto i32 = [&](size_t count) {
        int32_t* r = reinterpret_cast<int32_t*>(p);
        p += count * 4;
        return r;
    };
    S.rng = reinterpret_cast<uint32_t*>(i32(n));
    S.samp = i32(n);
    S.flags = i32(n);
    S.hmat = i32(n);
    S.n_closest = reinterpret_cast<uint32_t*>(i32(n));
    S.n_shadow = reinterpret_cast<uint32_t*>(i32(n));
    S.q_mat = sort ? i32((size_t)WF_NTYPES * n) : nullptr;
    S.counters = reinterpret_cast<uint32_t*>(i32(2 * WF_CNT_STRIDE));
    S.n_live = reinterpret_cast<uint32_t*>(i32(2));
    S.n_slots = n_slots;
    return RTR_OK;
}

template <typename K>
inline int wf_lds_attr(K kernel, size_t bytes, std::string& err) {
    if (bytes > 160 * 1024)
        return wf_fail(err, RTR_ERR_UNSUPPORTED, "this traversal of the scene needs a deeper stack than 160 KiB of LDS holds");
    if (bytes > 64 * 1024)
        WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)bytes));
    return RTR_OK;
}

/* Runs the whole render on `stream` and returns when it has finished (the iteration loop is
 * driven from the host, which polls the live-slot counter every `check` iterations). */
inline int wavefront_render(WavefrontPool& pool, const DScene* sc, const bool has_lights, const bool lean,
                            const bool quadlit,
                            const bool sort, const int trav, const size_t lds, const RenderK& Pin, int integrator,
                            double* d_rgb, int64_t row_stride, hipStream_t stream, std::atomic<uint32_t>* cancelled_upto,
                            int* launches, std::string& err) {
    RenderK P = Pin;
    const long long n_slots_ll = (long long)P.n_tiles * P.chunks * RTR_BLOCK;
    if (n_slots_ll > (1ll << 30)) return wf_fail(err, RTR_ERR_UNSUPPORTED, "path pool larger than 2^30 slots");
    WfState S{};
    int rc = wf_alloc(pool, S, (int)n_slots_ll, sort, err);
    if (rc) return rc;
    const bool media = trav == RT_TRAV_MEDIA || trav == RT_TRAV_PROGRAM;
    const bool mis = integrator == RTR_INTEGRATOR_MIS;
    const dim3 block(RTR_BLOCK);
    const int n_blocks = (S.n_slots + RTR_BLOCK - 1) / RTR_BLOCK;
    const dim3 grid((unsigned)n_blocks); /* one slot per lane; kernels keep the grid-stride form */
    int n_launch = 0;
    WF_HIP(hipMemsetAsync(S.n_live, 0, 8, stream));
    hipLaunchKernelGGL(wf_init, grid, block, 0, stream, S, P);
    ++n_launch;

#define WF_EXTEND_R(T, R)                                                                                \
    do {                                                                                                 \
        if ((rc = wf_lds_attr(wf_extend<T, true, R>, lds, err)) || (rc = wf_lds_attr(wf_extend<T, false, R>, lds, err))) \
            return rc;                                                                                   \
        if (sort)                                                                                        \
            hipLaunchKernelGGL((wf_extend<T, true, R>), grid, block, lds, stream, sc, S, P, par);        \
        else                                                                                             \
            hipLaunchKernelGGL((wf_extend<T, false, R>), grid, block, lds, stream, sc, S, P, par);       \
        ++n_launch;                                                                                      \
    } while (0)
#define WF_EXTEND(T)                \
    do {                            \
        if (lean || quadlit)        \
            WF_EXTEND_R(T, false);  \
        else                        \
            WF_EXTEND_R(T, true);   \
    } while (0)
#define WF_SHADE(I, PH, M)                                                                               \
    do {                                                                                                 \
        if (sort)                                                                                        \
            hipLaunchKernelGGL((wf_shade<I, PH, M, true>), grid, block, 0, stream, sc, S, P, par);       \
        else                                                                                             \
            hipLaunchKernelGGL((wf_shade<I, PH, M, false>), grid, block, 0, stream, sc, S, P, par);      \
        ++n_launch;                                                                                      \
    } while (0)
#define WF_CONNECT(T)                                                                                    \
    do {                                                                                                 \
        if ((rc = wf_lds_attr(wf_connect<T>, lds, err))) return rc;                                      \
        hipLaunchKernelGGL(wf_connect<T>, grid, block, lds, stream, sc, S, P);                           \
        ++n_launch;                                                                                      \
    } while (0)

    int iter = 0;
    const int check = 32;
    bool cancelled = false;
    for (;;) {
        for (int b = 0; b < check; ++b, ++iter) {
            const int par = iter & 1;
            if (trav == RT_TRAV_FLAT)
                WF_EXTEND(RT_TRAV_FLAT);
            else if (trav == RT_TRAV_FAST)
                WF_EXTEND(RT_TRAV_FAST);
            else if (trav == RT_TRAV_PROGRAM)
                WF_EXTEND_R(RT_TRAV_PROGRAM, true);
            else if (media)
                WF_EXTEND_R(RT_TRAV_MEDIA, true);
            else
                WF_EXTEND(RT_TRAV_EXACT);
            if (!mis) {
                if (lean)
                    WF_SHADE(RTR_INTEGRATOR_RR, 0, RT_MS_LEAN);
                else
                    WF_SHADE(RTR_INTEGRATOR_RR, 0, RT_MS_FULL);
            } else if (!media) {
                if (lean)
                    WF_SHADE(RTR_INTEGRATOR_MIS, 0, RT_MS_LEAN);
                else if (quadlit)
                    WF_SHADE(RTR_INTEGRATOR_MIS, 0, RT_MS_QUADLIT);
                else
                    WF_SHADE(RTR_INTEGRATOR_MIS, 0, RT_MS_FULL);
                if (has_lights) {
                    if (trav == RT_TRAV_FLAT)
                        WF_CONNECT(RT_TRAV_FLAT);
                    else if (trav == RT_TRAV_FAST)
                        WF_CONNECT(RT_TRAV_FAST);
                    else
                        WF_CONNECT(RT_TRAV_EXACT);
                }
            } else {
                if (quadlit)
                    WF_SHADE(RTR_INTEGRATOR_MIS, 1, RT_MS_QUADLIT);
                else
                    WF_SHADE(RTR_INTEGRATOR_MIS, 1, RT_MS_FULL);
                if (has_lights) {
                    if (trav == RT_TRAV_PROGRAM)
                        WF_CONNECT(RT_TRAV_PROGRAM);
                    else
                        WF_CONNECT(RT_TRAV_MEDIA);
                }
                WF_SHADE(RTR_INTEGRATOR_MIS, 2, RT_MS_FULL);
            }
        }
        WF_HIP(hipGetLastError());
        WF_HIP(hipMemcpyAsync(pool.h_live, S.n_live, 4, hipMemcpyDeviceToHost, stream));
        WF_HIP(hipStreamSynchronize(stream));
        if (*pool.h_live == 0) break;
        if (cancelled_upto && cancelled_upto->load() >= P.render_id) {
            cancelled = true;
            break;
        }
    }
#undef WF_EXTEND
#undef WF_EXTEND_R
#undef WF_SHADE
#undef WF_CONNECT
    hipLaunchKernelGGL(wf_finish, grid, block, 0, stream, S, P);
    ++n_launch;
    if (cancelled) { /* unfinished pixels have no sum yet: leave the caller's buffer untouched */
        if (launches) *launches = n_launch;
        return RTR_ERR_CANCELLED;
    }
    ResolveK R{P, d_rgb, (long long)row_stride};
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)P.n_tiles), block, 0, stream, R);
    ++n_launch;
    WF_HIP(hipGetLastError());
    if (launches) *launches = n_launch;
    return RTR_OK;
}
