/*
 * rt_wavefront.h -- the wavefront pipeline: the integrator loop of
 * renderer/mis_path_integrator.h:25-150 (and the other four integrators of the reference CLI) cut at
 * its two ray casts into stage kernels that exchange path state through SoA arrays in HBM.
 *
 * Path pool.  Slot = (owned tile, spp chunk, pixel of the tile): the same numbering as the
 * megakernel's (blockIdx, threadIdx); 256 consecutive slots = one BLOCK = one tile's chunk.  A slot
 * runs the samples of its chunk back to back (in-slot regeneration), so per-pixel sums keep the
 * sample order of renderer.h:72-79 and the pool stays full until blocks run out of samples.  Every
 * array is indexed by slot: consecutive lanes touch consecutive 8-byte words.
 *
 * One iteration (one bounce of every live path), each stage over the list of LIVE blocks only:
 *   wf_extend   persistent threads: a wave walks its blocks, every lane takes the next slot as soon as
 *               its previous ray is finished (the resumable traversal machine of rt_machine.h: a lane
 *               never waits for the slowest ray of its wave).  Flush a finished sample into the
 *               pixel sum, start the next camera sample (renderer.h:73-75), cast the closest-hit ray
 *               (mis_path_integrator.h:37), handle the miss, else store the hit as 16 bytes:
 *               (t, primitive reference | medium step, instance).
 *   wf_shade    one workgroup per block; the block's hits are sorted by material class in LDS
 *               (material-sorted shading waves), the hit record is rebuilt from (t, reference,
 *               instance) with the reference's own arithmetic (fast_finish), then emission + MIS
 *               weight, light sample (:72-103,191-229), BSDF sample, throughput, Russian roulette
 *               (:105-146); writes the next ray and, if the light sample is usable, a shadow request.
 *   wf_connect  persistent threads over the shadow requests: occlusion ray (:210-213) on the same
 *               machine (first hit ends the cast once no medium is left to draw); unoccluded ->
 *               L += contrib.
 * Scenes with participating media draw random numbers INSIDE both ray casts
 * (constant_medium.h:85), so there the shade stage is split around the connect stage
 * (phase 1, wf_connect, phase 2) to keep the reference's draw order.
 *
 * Nothing in an iteration touches a shared address: blocks belong to one wave (extend / connect) or
 * one workgroup (shade) per launch; cast counts are summed per wave and added once per launch; the
 * number of live slots of a block is written by the wave that walked it.  wf_compact (one
 * workgroup) rewrites the live-block list and publishes its length to host-mapped memory; the host
 * enqueues iterations in batches, stays at most two batches ahead of the GPU (events) and stops
 * when a published length is zero -- it never drains the stream to look.
 *
 * The pipeline runs the compiled traversals (flat / box-tree / step-program scenes).  Graphs that
 * need the reference-order walk (a medium under a transform, hollow spheres) run on the megakernel.
 */
#pragma once

#include "rt_launch.h"
#include "rt_machine.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
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
    /* hit: t, reference into DScene::fprim (>= 0) or ~step of the medium that scattered the ray, instance */
    double* ht;
    int32_t *href, *hinst;
    /* shadow request: origin, direction, t_max, contribution if unoccluded */
    double *sox, *soy, *soz, *swx, *swy, *swz, *stmax, *scx, *scy, *scz;
    /* live blocks */
    int32_t* block_live; /* [n_blocks] slots of the block that are not WF_DONE (written by wf_extend) */
    int32_t* list[2];    /* ping-pong lists of live blocks */
    uint32_t* n_list;    /* [2] their lengths */
    uint32_t* host_live; /* host-mapped: length of the newest list */
    int n_slots, n_blocks;
};

void WavefrontPool::release() {
    if (slab) (void)hipFree(slab);
    if (h_live) (void)hipHostFree(h_live);
    for (hipEvent_t& e : ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    slab = nullptr, slab_bytes = 0, h_live = nullptr;
}

RT_DEV V3 ldv(const double* x, const double* y, const double* z, int i) { return mk(x[i], y[i], z[i]); }
RT_DEV void stv(double* x, double* y, double* z, int i, V3 v) { x[i] = v.x, y[i] = v.y, z[i] = v.z; }

/* slot -> pixel (same layout as k_mega) */
RT_DEV void slot_pixel(const RenderK& P, int slot, int& i, int& j, int& chunk, bool& active) {
    const int blk = slot / RTR_BLOCK, tid = slot % RTR_BLOCK;
    chunk = blk % P.chunks;
    tile_pixel(P, blk / P.chunks, tid, i, j, active);
}

__global__ void __launch_bounds__(RTR_BLOCK) wf_init(const WfState S, const RenderK P) {
    const int slot = blockIdx.x * RTR_BLOCK + threadIdx.x;
    int i, j, chunk;
    bool active;
    slot_pixel(P, slot, i, j, chunk, active);
    int s0, s1;
    chunk_range(P, chunk, s0, s1);
    active = active && s0 < s1;
    S.ax[slot] = 0, S.ay[slot] = 0, S.az[slot] = 0;
    S.lx[slot] = 0, S.ly[slot] = 0, S.lz[slot] = 0;
    S.samp[slot] = s0 - 1;
    S.flags[slot] = active ? (WF_NEED_SAMPLE | WF_FIRST) : WF_DONE;
    if (!active) { /* pixels outside the region still own a partial-sum cell */
        double* out = P.partial + (size_t)blockIdx.x * 3 * RTR_BLOCK + threadIdx.x;
        out[0] = 0, out[RTR_BLOCK] = 0, out[2 * RTR_BLOCK] = 0;
    }
    const int live = __syncthreads_count(active);
    if (threadIdx.x == 0) {
        P.done[blockIdx.x] = 1; /* k_resolve runs only after every slot has finished */
        S.block_live[blockIdx.x] = live;
        S.list[0][blockIdx.x] = blockIdx.x;
        if (blockIdx.x == 0) S.n_list[0] = (uint32_t)S.n_blocks, S.n_list[1] = 0;
    }
}

/* once per render: finished samples per slot into the render statistics */
__global__ void __launch_bounds__(RTR_BLOCK) wf_finish(const WfState S, const RenderK P) {
    const int slot = blockIdx.x * RTR_BLOCK + threadIdx.x;
    int i, j, chunk;
    bool active;
    slot_pixel(P, slot, i, j, chunk, active);
    int s0, s1;
    chunk_range(P, chunk, s0, s1);
    unsigned long long a = 0;
    if (active) a = (unsigned long long)(S.samp[slot] - s0); /* samp ends at s_end: finished samples */
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0 && a) atomicAdd(&P.stats[0], a);
}

/* keep the blocks that still hold live slots: list[from] -> list[from ^ 1], order preserved */
__global__ void __launch_bounds__(1024) wf_compact(const WfState S, const int from) {
    __shared__ uint32_t wave_total[16];
    __shared__ uint32_t base;
    const int32_t* in = S.list[from];
    int32_t* out = S.list[from ^ 1];
    const uint32_t n = S.n_list[from];
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 1024) {
        const uint32_t i = i0 + threadIdx.x;
        const int b = i < n ? in[i] : -1;
        const bool keep = b >= 0 && S.block_live[b] > 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) wave_total[wave] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t off = base;
        for (int w = 0; w < wave; ++w) off += wave_total[w];
        if (keep) out[off + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1))] = b;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int w = 0; w < 16; ++w) t += wave_total[w];
            base += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        S.n_list[from ^ 1] = base;
        __hip_atomic_store(S.host_live, base, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

/* ---- persistent-threads block cursor of one wave (lives in LDS: updated under divergent control flow) ---- */
struct WaveCursor {
    int cur, end;  /* slots [cur, end) of the current block are not handed out yet */
    int entry;     /* next list entry of this wave */
    int blk, done; /* current block and how many of its slots were found / became WF_DONE */
};
RT_DEV int lane_rank(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
struct BlockWalk {
    volatile WaveCursor* wc;
    const int32_t* list;
    int n_entries, stride;
    int32_t* block_live; /* nullptr: this stage does not count */
    /* Hand the next slots of this wave's blocks to the calling lanes (all lanes of the wave that are in
     * M_FETCH).  Returns the slot or -1 ("ask again"); `exhausted` = no block is left for this wave. */
    RT_DEV int take(bool& exhausted) const {
        exhausted = false;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(true);
        const int n = __builtin_popcountll(mask), rank = lane_rank(mask);
        int cur = wc->cur, end = wc->end;
        if (cur >= end) {
            const int entry = wc->entry, blk = wc->blk, done = wc->done;
            if (block_live && blk >= 0 && rank == 0) block_live[blk] = RTR_BLOCK - done;
            if (entry >= n_entries) {
                if (rank == 0) wc->blk = -1;
                exhausted = true;
                return -1;
            }
            const int nb = list[entry];
            cur = nb * RTR_BLOCK, end = cur + RTR_BLOCK;
            if (rank == 0) wc->entry = entry + stride, wc->blk = nb, wc->done = 0, wc->end = end;
        }
        const int avail = end - cur, take_n = n < avail ? n : avail;
        if (rank == 0) wc->cur = cur + take_n;
        return rank < take_n ? cur + rank : -1;
    }
    RT_DEV void count_done(bool dead) const {
        const unsigned long long d = __builtin_amdgcn_ballot_w64(dead);
        if (d && lane_rank(__builtin_amdgcn_ballot_w64(true)) == 0) wc->done = wc->done + __builtin_popcountll(d);
    }
};
RT_DEV BlockWalk block_walk(WaveCursor* cursors, const WfState& S, int parity, bool count) {
    const int wave = threadIdx.x >> 6, waves_per_block = RTR_BLOCK / 64;
    BlockWalk w;
    w.wc = cursors + wave;
    w.list = S.list[parity];
    w.n_entries = (int)S.n_list[parity];
    w.stride = gridDim.x * waves_per_block;
    w.block_live = count ? S.block_live : nullptr;
    if ((threadIdx.x & 63) == 0) {
        cursors[wave].cur = 0, cursors[wave].end = 0;
        cursors[wave].entry = blockIdx.x * waves_per_block + wave;
        cursors[wave].blk = -1, cursors[wave].done = 0;
    }
    return w;
}

/* ---- extend ------------------------------------------------------------------------------------------ */
/* What the extend stage does with a slot before the cast: flush the finished sample into the pixel sum and
 * start the next camera sample (renderer.h:73-79), or load the continued ray.  Returns false when the slot is
 * (or has just become) WF_DONE. */
RT_DEV bool wf_extend_load(const DScene& sc, const WfState& S, const RenderK& P, int slot, int& flags, V3& ro, V3& rd,
                           Real& tm, uint32_t& rng) {
    flags = S.flags[slot];
    const int status = WF_STATUS(flags);
    if (status == WF_DONE) return false;
    if (status == WF_NEED_SAMPLE) {
        int i, j, chunk;
        bool active;
        slot_pixel(P, slot, i, j, chunk, active);
        V3 acc = ldv(S.ax, S.ay, S.az, slot);
        if (!(flags & WF_FIRST)) acc = add(acc, ldv(S.lx, S.ly, S.lz, slot)); /* renderer.h:77-78 */
        const int s = S.samp[slot] + 1;
        S.samp[slot] = s;
        int s_begin, s_end;
        chunk_range(P, chunk, s_begin, s_end);
        if (s >= s_end) {
            S.flags[slot] = WF_DONE;
            double* out = P.partial + (size_t)(slot / RTR_BLOCK) * 3 * RTR_BLOCK + (slot % RTR_BLOCK);
            out[0] = acc.x, out[RTR_BLOCK] = acc.y, out[2 * RTR_BLOCK] = acc.z;
            return false;
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
        return true;
    }
    ro = ldv(S.ox, S.oy, S.oz, slot);
    rd = ldv(S.dx, S.dy, S.dz, slot);
    tm = S.tm[slot];
    rng = S.rng[slot];
    flags &= ~3;
    return true;
}
/* ... and after it: the miss (mis_path_integrator.h:37-67, rr_path_integrator.h:31-33) or the 16-byte hit.
 * `ref` < 0 and `med` < 0: nothing was hit.  RICH = false: scenes lit by QuadLights only, no environment-light
 * code on the miss branch. */
template <bool RICH>
RT_DEV void wf_extend_store(const DScene& sc, const WfState& S, const RenderK& P, int slot, int flags, V3 ro, V3 rd, Real t,
                            int ref, int inst, int med, uint32_t rng) {
    if (ref < 0 && med < 0) {
        const V3 thr = ldv(S.tx, S.ty, S.tz, slot);
        V3 add_l;
        if (RICH && P.integrator == RTR_INTEGRATOR_MIS)
            add_l = miss_radiance<RTR_INTEGRATOR_MIS, RT_MS_FULL>(sc, thr, ro, rd, flags >> 8, (flags & WF_SPEC) != 0, S.pdf[slot]);
        else if (RICH && P.integrator == RTR_INTEGRATOR_NEE)
            add_l = miss_radiance<RTR_INTEGRATOR_NEE, RT_MS_FULL>(sc, thr, ro, rd, flags >> 8, (flags & WF_SPEC) != 0, 0.0);
        else
            add_l = mul(thr, ld3(sc.background));
        stv(S.lx, S.ly, S.lz, slot, add(ldv(S.lx, S.ly, S.lz, slot), add_l));
        S.flags[slot] = flags | WF_NEED_SAMPLE;
    } else {
        S.ht[slot] = t;
        S.href[slot] = med >= 0 ? ~med : ref;
        S.hinst[slot] = inst;
        S.flags[slot] = flags | WF_HIT;
    }
    S.rng[slot] = rng;
}

/* persistent-threads form: the stage as a client of the traversal machine */
template <bool RICH>
struct ExtendClient {
    const DScene& sc;
    const WfState& S;
    const RenderK& P;
    BlockWalk walk;
    unsigned n_closest;

    RT_DEV void fetch(MLane& m) {
        bool exhausted;
        const int slot = walk.take(exhausted);
        if (exhausted) {
            m_set_phase(m, M_IDLE);
            return;
        }
        if (slot < 0) return;
        int flags;
        V3 ro = mk(0, 0, 0), rd = mk(0, 0, 0);
        Real tm = 0;
        uint32_t rng = 1;
        const bool live = wf_extend_load(sc, S, P, slot, flags, ro, rd, tm, rng);
        walk.count_done(!live);
        if (!live) return; /* ask for another slot */
        m.slot = slot;
        m.aux = flags;
        ++n_closest;
        m_begin(m, ro, rd, tm, RT_INF, rng);
    }
    RT_DEV void finish(MLane& m) {
        wf_extend_store<RICH>(sc, S, P, m.slot, m.aux, m.o, m.d, m.best_t, m.best_ref, m.best_inst, m.best_med, m.rng);
        m_set_phase(m, M_FETCH);
    }
};

#ifndef RTR_WF_EXTEND_WAVES
#define RTR_WF_EXTEND_WAVES 4
#endif

template <bool RICH>
__global__ void __launch_bounds__(RTR_BLOCK, RTR_WF_EXTEND_WAVES)
    wf_extend(const DScene* __restrict__ scp, const WfState S, const RenderK P, const int parity) {
    extern __shared__ int lds_stack[];
    __shared__ WaveCursor cursors[RTR_BLOCK / 64];
    const DScene& sc = *scp;
    const Stack st{lds_stack + threadIdx.x};
    ExtendClient<RICH> client{sc, S, P, block_walk(cursors, S, parity, true), 0u};
    MLane m;
    m_set_phase(m, M_FETCH);
    m_run<false>(sc, sc.fvisit, sc.n_fvisit, m, st, client);
    const unsigned long long c = wave_sum(client.n_closest);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&P.stats[1], c);
}

/* ---- connect: occlusion rays of the pending light connections (mis_path_integrator.h:210-213) ------------ */
template <bool MEDIA>
struct ConnectClient {
    const DScene& sc;
    const WfState& S;
    BlockWalk walk;
    unsigned n_shadow;

    RT_DEV void fetch(MLane& m) {
        bool exhausted;
        const int slot = walk.take(exhausted);
        if (exhausted) {
            m_set_phase(m, M_IDLE);
            return;
        }
        if (slot < 0) return;
        const int flags = S.flags[slot];
        if (!(flags & WF_SHADOW)) return;
        S.flags[slot] = flags & ~WF_SHADOW;
        m.slot = slot;
        ++n_shadow;
        /* shadow_ray = ray(rec.p, wi, time 0) over [0.001, dist - 0.001] (:210-213) */
        m_begin(m, ldv(S.sox, S.soy, S.soz, slot), ldv(S.swx, S.swy, S.swz, slot), 0.0, S.stmax[slot],
                MEDIA ? S.rng[slot] : 1u);
    }
    RT_DEV void finish(MLane& m) {
        const int slot = m.slot;
        if (MEDIA) S.rng[slot] = m.rng;
        if (!m_any_hit(m)) stv(S.lx, S.ly, S.lz, slot, add(ldv(S.lx, S.ly, S.lz, slot), ldv(S.scx, S.scy, S.scz, slot)));
        m_set_phase(m, M_FETCH);
    }
};

template <bool MEDIA>
__global__ void __launch_bounds__(RTR_BLOCK, RTR_WF_EXTEND_WAVES)
    wf_connect(const DScene* __restrict__ scp, const WfState S, const RenderK P, const int parity) {
    extern __shared__ int lds_stack[];
    __shared__ WaveCursor cursors[RTR_BLOCK / 64];
    const DScene& sc = *scp;
    const Stack st{lds_stack + threadIdx.x};
    ConnectClient<MEDIA> client{sc, S, block_walk(cursors, S, parity, false), 0u};
    MLane m;
    m_set_phase(m, M_FETCH);
    m_run<true>(sc, sc.fvisit, sc.n_fvisit, m, st, client);
    const unsigned long long c = wave_sum(client.n_shadow);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&P.stats[2], c);
}

/* ---- lockstep forms of the two casting stages ------------------------------------------------------------
 * One lane = one slot of the block, the whole wave runs run_program() / trace_fast() together: every
 * instance, step and primitive record comes through scalar loads, the non-tree part of a cast runs with all
 * lanes.  (The machine above keeps the lanes of a wave busy inside box trees, but splits them between its
 * two phases; which form is faster for a scene is measured, see DESIGN.md.) */
template <int TRAV, bool RICH>
__global__ void __launch_bounds__(RTR_BLOCK, 4)
    wf_extend_ls(const DScene* __restrict__ scp, const WfState S, const RenderK P, const int parity) {
    extern __shared__ int lds_stack[];
    const DScene& sc = *scp;
    const Stack st{lds_stack + threadIdx.x};
    const int32_t* list = S.list[parity];
    const int n_entries = (int)S.n_list[parity];
    unsigned n_closest = 0;
    for (int e = blockIdx.x; e < n_entries; e += gridDim.x) {
        const int blk = list[e], slot = blk * RTR_BLOCK + threadIdx.x;
        int flags;
        V3 ro = mk(0, 0, 0), rd = mk(0, 0, 0);
        Real tm = 0;
        uint32_t rng = 1;
        const bool live = wf_extend_load(sc, S, P, slot, flags, ro, rd, tm, rng);
        const int n_live = __syncthreads_count(live);
        if (threadIdx.x == 0) S.block_live[blk] = n_live;
        if (!live) continue;
        ++n_closest;
        Real t = RT_INF;
        int ref = -1, inst = -1, med = -1;
        if (rt_is_program(TRAV))
            run_program<false>(sc, ro, rd, tm, 0.001, t, ref, inst, med, rng, st);
        else
            trace_fast<false, TRAV == RT_TRAV_FAST>(sc, sub_scene0(sc), ro, rd, tm, 0.001, t, ref, inst, st, 0);
        wf_extend_store<RICH>(sc, S, P, slot, flags, ro, rd, t, ref, inst, med, rng);
    }
    const unsigned long long c = wave_sum(n_closest);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&P.stats[1], c);
}

template <int TRAV>
__global__ void __launch_bounds__(RTR_BLOCK, 4)
    wf_connect_ls(const DScene* __restrict__ scp, const WfState S, const RenderK P, const int parity) {
    extern __shared__ int lds_stack[];
    const DScene& sc = *scp;
    const Stack st{lds_stack + threadIdx.x};
    const int32_t* list = S.list[parity];
    const int n_entries = (int)S.n_list[parity];
    unsigned n_shadow = 0;
    for (int e = blockIdx.x; e < n_entries; e += gridDim.x) {
        const int slot = list[e] * RTR_BLOCK + threadIdx.x;
        const int flags = S.flags[slot];
        if (!(flags & WF_SHADOW)) continue;
        S.flags[slot] = flags & ~WF_SHADOW;
        ++n_shadow;
        uint32_t rng = rt_is_program(TRAV) ? S.rng[slot] : 1u;
        const bool blocked = cast_shadow<TRAV>(sc, ldv(S.sox, S.soy, S.soz, slot), ldv(S.swx, S.swy, S.swz, slot),
                                               S.stmax[slot], rng, st);
        if (rt_is_program(TRAV)) S.rng[slot] = rng;
        if (!blocked) stv(S.lx, S.ly, S.lz, slot, add(ldv(S.lx, S.ly, S.lz, slot), ldv(S.scx, S.scy, S.scz, slot)));
    }
    const unsigned long long c = wave_sum(n_shadow);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&P.stats[2], c);
}

/* ---- shade ----------------------------------------------------------------------------------------------- */
/* the hit record of the reference, rebuilt from what wf_extend stored */
template <bool UV_POSSIBLE>
RT_DEV void wf_hit_record(const DScene& sc, const WfState& S, int slot, V3 o, V3 d, Real tm, Hit& rec) {
    const Real t = S.ht[slot];
    const int ref = S.href[slot];
    rec.u = 0, rec.v = 0;
    if (ref < 0) {
        medium_finish(sc, ~ref, o, d, t, rec);
    } else if (UV_POSSIBLE && sc.needs_uv) {
        fast_finish<true>(sc, o, d, tm, t, ref, S.hinst[slot], rec);
    } else {
        fast_finish<false>(sc, o, d, tm, t, ref, S.hinst[slot], rec);
    }
}

/* after the second half of a bounce: store the continued path or mark the sample finished */
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

/* PHASE 0: whole shading (no media).  PHASE 1: first half only (emission + light sample).
 * PHASE 2: second half only (BSDF sample + roulette), after the connect stage.
 * SORT: the block's hits are shaded in material-class order (scenes that mix classes). */
template <int INTEG, int PHASE, int MS, bool SORT>
__global__ void __launch_bounds__(RTR_BLOCK, MS == RT_MS_LEAN ? 4 : 2)
    wf_shade(const DScene* __restrict__ scp, const WfState S, const RenderK P, const int parity) {
    const DScene& sc = *scp;
    const int32_t* list = S.list[parity];
    const int n_entries = (int)S.n_list[parity];
    __shared__ int key_count[WF_NTYPES];
    __shared__ short order[RTR_BLOCK];
    /* one workgroup per list entry: the launch covers the longest possible list, surplus workgroups leave at once */
    for (int e = blockIdx.x; e < n_entries; e += gridDim.x) {
        const int slot0 = list[e] * RTR_BLOCK;
        int slot = slot0 + threadIdx.x;
        int flags = S.flags[slot];
        bool work = WF_STATUS(flags) == WF_HIT;
        if (SORT) { /* counting sort of the block's hits by material class */
            if (threadIdx.x < WF_NTYPES) key_count[threadIdx.x] = 0;
            __syncthreads();
            int key = -1, pos = 0;
            if (work) {
                const int ref = S.href[slot];
                const int mat = ref < 0 ? as_const(sc.fstep)[~ref].mat : as_const(sc.fprim)[ref].a;
                key = as_const(sc.materials)[mat].type;
                pos = atomicAdd(&key_count[key], 1);
            }
            __syncthreads();
            int total = 0;
            for (int q = 0; q < WF_NTYPES; ++q) {
                const int c = key_count[q];
                if (q < key) pos += c;
                total += c;
            }
            if (work) order[pos] = (short)threadIdx.x;
            __syncthreads();
            work = (int)threadIdx.x < total;
            if (work) {
                slot = slot0 + order[threadIdx.x];
                flags = S.flags[slot];
            }
            __syncthreads(); /* order[] is rewritten by the next block of this workgroup */
        }
        if (!work) continue;
        PathState ps;
        ps.ro = ldv(S.ox, S.oy, S.oz, slot);
        ps.rd = ldv(S.dx, S.dy, S.dz, slot);
        ps.tm = S.tm[slot];
        ps.thr = ldv(S.tx, S.ty, S.tz, slot);
        ps.L = mk(0.0, 0.0, 0.0); /* at most one term (the emission) is added here: L + e is the rounding of L += e */
        ps.prev_bsdf_pdf = S.pdf[slot];
        ps.depth = flags >> 8;
        ps.specular_bounce = (flags & WF_SPEC) != 0;
        Hit rec;
        wf_hit_record<MS == RT_MS_FULL>(sc, S, slot, ps.ro, ps.rd, ps.tm, rec);
        uint32_t rng = S.rng[slot];
        bool go;
        int extra = flags & WF_SHADOW; /* phase 2 keeps what phase 1 requested */
        if (INTEG == RTR_INTEGRATOR_RR) {
            go = shade_rr<MS>(sc, ps, rec, rng, P.rr_start);
            extra = 0;
        } else if (INTEG == RTR_INTEGRATOR_PATH) {
            go = shade_path<MS>(sc, ps, rec, rng);
            extra = 0;
        } else {
            const V3 wo = neg(unit(ps.rd));
            const MatCtx mc = mat_prepare<MS>(sc, rec);
            if (PHASE != 2) {
                ShadowReq rq;
                shade_a_mis<MS, INTEG>(sc, ps, rec, mc, wo, rng, rq);
                extra = 0;
                if (rq.valid) {
                    stv(S.sox, S.soy, S.soz, slot, rec.p);
                    stv(S.swx, S.swy, S.swz, slot, rq.wi);
                    S.stmax[slot] = rq.tmax;
                    stv(S.scx, S.scy, S.scz, slot, rq.contrib);
                    extra = WF_SHADOW;
                }
            }
            if (PHASE == 1) {
                if (ps.L.x != 0.0 || ps.L.y != 0.0 || ps.L.z != 0.0)
                    stv(S.lx, S.ly, S.lz, slot, add(ldv(S.lx, S.ly, S.lz, slot), ps.L));
                S.rng[slot] = rng;
                if (extra) S.flags[slot] = flags | WF_SHADOW;
                continue;
            }
            go = shade_b_mis<MS, INTEG>(sc, ps, rec, mc, wo, rng, P.rr_start);
        }
        if (ps.L.x != 0.0 || ps.L.y != 0.0 || ps.L.z != 0.0)
            stv(S.lx, S.ly, S.lz, slot, add(ldv(S.lx, S.ly, S.lz, slot), ps.L));
        wf_store_path(S, slot, ps, go, P.max_depth, PHASE == 2 ? 0 : extra);
        S.rng[slot] = rng;
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

inline int wf_alloc(WavefrontPool& pool, WfState& S, int n_slots, std::string& err) {
    const size_t n = (size_t)n_slots, nb = n / RTR_BLOCK;
    const size_t n_f64 = 28; /* ray 7, path 10, hit t 1, shadow 10 */
    const size_t bytes = n_f64 * n * 8 + 5 * n * 4 /* rng samp flags href hinst */ + 3 * nb * 4 + 256;
    if (pool.slab_bytes < bytes) {
        if (pool.slab) WF_HIP(hipFree(pool.slab));
        pool.slab = nullptr, pool.slab_bytes = 0;
        hipError_t e = hipMalloc(&pool.slab, bytes);
        if (e != hipSuccess) return wf_fail(err, RTR_ERR_NOMEM, std::string("hipMalloc(path pool): ") + hipGetErrorString(e));
        pool.slab_bytes = bytes;
    }
    if (!pool.h_live) {
        WF_HIP(hipHostMalloc(reinterpret_cast<void**>(&pool.h_live), 64, hipHostMallocMapped));
        for (hipEvent_t& e : pool.ev) WF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    char* p = static_cast<char*>(pool.slab);
    auto f64 = [&]() {
        double* r = reinterpret_cast<double*>(p);
        p += n * 8;
        return r;
    };
    double** arrs[] = {&S.ox, &S.oy, &S.oz, &S.dx, &S.dy, &S.dz, &S.tm, &S.tx, &S.ty, &S.tz, &S.lx,
                       &S.ly, &S.lz, &S.ax, &S.ay, &S.az, &S.pdf, &S.ht, &S.sox, &S.soy, &S.soz, &S.swx,
                       &S.swy, &S.swz, &S.stmax, &S.scx, &S.scy, &S.scz};
    static_assert(sizeof(arrs) / sizeof(arrs[0]) == 28, "array count");
    for (double** a : arrs) *a = f64();
    auto i32 = [&](size_t count) {
        int32_t* r = reinterpret_cast<int32_t*>(p);
        p += count * 4;
        return r;
    };
    S.rng = reinterpret_cast<uint32_t*>(i32(n));
    S.samp = i32(n);
    S.flags = i32(n);
    S.href = i32(n);
    S.hinst = i32(n);
    S.block_live = i32(nb);
    S.list[0] = i32(nb);
    S.list[1] = i32(nb);
    S.n_list = reinterpret_cast<uint32_t*>(i32(2));
    void* dev_live = nullptr;
    WF_HIP(hipHostGetDevicePointer(&dev_live, pool.h_live, 0));
    S.host_live = static_cast<uint32_t*>(dev_live);
    S.n_slots = n_slots;
    S.n_blocks = (int)nb;
    return RTR_OK;
}

template <typename K>
inline int wf_lds_attr(K kernel, size_t bytes, std::string& err) {
    hipFuncAttributes fa{};
    WF_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel)));
    if (bytes + fa.sharedSizeBytes > 160 * 1024) /* static LDS of the kernel counts against the same 160 KiB */
        return wf_fail(err, RTR_ERR_UNSUPPORTED, "this traversal of the scene needs a deeper stack than 160 KiB of LDS holds");
    if (bytes > 64 * 1024)
        WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)bytes));
    return RTR_OK;
}

/* Enqueues the whole render on `stream`.  The host stays at most two batches of iterations ahead of
 * the GPU and returns once the published number of live blocks is zero (everything it enqueued past that
 * point finds empty lists); it does not wait for the stream to drain. */
int wavefront_render(WavefrontPool& pool, const DScene* sc, const WavefrontPlan& plan, const RenderK& Pin, int integrator,
                     double* d_rgb, int64_t row_stride, unsigned char* tile_done, hipStream_t stream,
                     std::atomic<uint32_t>* cancelled_upto, int* launches, std::string& err) {
    RenderK P = Pin;
    const long long n_slots_ll = (long long)P.n_tiles * P.chunks * RTR_BLOCK;
    if (n_slots_ll > (1ll << 30)) return wf_fail(err, RTR_ERR_UNSUPPORTED, "path pool larger than 2^30 slots");
    WfState S{};
    int rc = wf_alloc(pool, S, (int)n_slots_ll, err);
    if (rc) return rc;
    const bool shadows = plan.has_lights && (integrator == RTR_INTEGRATOR_MIS || integrator == RTR_INTEGRATOR_NEE);
    const bool split = plan.media && shadows;
    const dim3 block(RTR_BLOCK);
    const dim3 grid_all((unsigned)S.n_blocks);
    const dim3 grid_cast((unsigned)std::max(1, std::min(S.n_blocks, plan.n_cus * RTR_WF_EXTEND_WAVES)));
    /* workgroups per CU of the block-per-workgroup stages; each walks several blocks of the list.  Measured on
     * MI355X: the flat Cornell stages are fastest with few, long-lived workgroups (4 per CU: 1 150 Msamples/s
     * against 900 at 16 and 620 at 32), scenes with box trees with many (32 per CU) */
    const int grid_mult = plan.trav == RT_TRAV_FLAT ? 4 : 32;
    const uint32_t grid_cap = (uint32_t)plan.n_cus * (uint32_t)grid_mult;
    dim3 grid_shade(std::min((uint32_t)S.n_blocks, grid_cap)); /* shrinks with the published live count */
    if (plan.lds > 64 * 1024) /* deep box trees only: every casting kernel gets the attribute */
        if ((rc = wf_lds_attr(wf_extend<true>, plan.lds, err)) || (rc = wf_lds_attr(wf_extend<false>, plan.lds, err)) ||
            (rc = wf_lds_attr(wf_connect<true>, plan.lds, err)) || (rc = wf_lds_attr(wf_connect<false>, plan.lds, err)) ||
            (rc = wf_lds_attr(wf_extend_ls<RT_TRAV_FAST, true>, plan.lds, err)) || (rc = wf_lds_attr(wf_extend_ls<RT_TRAV_FAST, false>, plan.lds, err)) ||
            (rc = wf_lds_attr(wf_extend_ls<RT_TRAV_PROGRAM_EXT, true>, plan.lds, err)) || (rc = wf_lds_attr(wf_extend_ls<RT_TRAV_PROGRAM_EXT, false>, plan.lds, err)) ||
            (rc = wf_lds_attr(wf_connect_ls<RT_TRAV_FAST>, plan.lds, err)) || (rc = wf_lds_attr(wf_connect_ls<RT_TRAV_PROGRAM_EXT>, plan.lds, err)))
            return rc;
    if (plan.lds > 160 * 1024) return wf_lds_attr(wf_extend<true>, plan.lds, err);
    const bool rich = !(plan.lean || plan.quadlit);
    int n_launch = 0;
    *pool.h_live = (uint32_t)S.n_blocks;
    hipLaunchKernelGGL(wf_init, grid_all, block, 0, stream, S, P);
    ++n_launch;

#define WF_SHADE_S(I, PH, M)                                                                             \
    do {                                                                                                 \
        if (plan.sort)                                                                                   \
            hipLaunchKernelGGL((wf_shade<I, PH, M, true>), grid_shade, block, 0, stream, sc, S, P, par); \
        else                                                                                             \
            hipLaunchKernelGGL((wf_shade<I, PH, M, false>), grid_shade, block, 0, stream, sc, S, P, par); \
        ++n_launch;                                                                                      \
    } while (0)
/* material-set variants as in the megakernel: lean (RR / MIS), QuadLights only (MIS), everything */
#define WF_SHADE(PH)                                                                    \
    do {                                                                                \
        switch (integrator) {                                                           \
        case RTR_INTEGRATOR_RR:                                                         \
            if (plan.lean)                                                              \
                WF_SHADE_S(RTR_INTEGRATOR_RR, 0, RT_MS_LEAN);                           \
            else                                                                        \
                WF_SHADE_S(RTR_INTEGRATOR_RR, 0, RT_MS_FULL);                           \
            break;                                                                      \
        case RTR_INTEGRATOR_PATH: WF_SHADE_S(RTR_INTEGRATOR_PATH, 0, RT_MS_FULL); break; \
        case RTR_INTEGRATOR_PBR: WF_SHADE_S(RTR_INTEGRATOR_PBR, 0, RT_MS_FULL); break;  \
        case RTR_INTEGRATOR_NEE: WF_SHADE_S(RTR_INTEGRATOR_NEE, PH, RT_MS_FULL); break; \
        default:                                                                        \
            if (plan.lean && PH == 0)                                                   \
                WF_SHADE_S(RTR_INTEGRATOR_MIS, 0, RT_MS_LEAN);                          \
            else if (plan.quadlit)                                                      \
                WF_SHADE_S(RTR_INTEGRATOR_MIS, PH, RT_MS_QUADLIT);                      \
            else                                                                        \
                WF_SHADE_S(RTR_INTEGRATOR_MIS, PH, RT_MS_FULL);                         \
        }                                                                               \
    } while (0)

    const int batch = 8;
    int par = 0, iter = 0, n_batches = 0;
    bool cancelled = false;
    /* every iteration advances every live slot by one path segment: a slot holds at most ceil(spp / chunks) samples
     * of at most max_depth segments (+ one iteration each to start the next sample) -- more iterations than that mean a
     * slot that never finishes (a bug in the live-block accounting, a NaN in the path state), not work */
    const long long iter_bound = ((long long)P.spp / P.chunks + 1) * ((long long)P.max_depth + 2) + 4 * batch;
    for (;;) {
        if (iter > iter_bound) {
            (void)hipStreamSynchronize(stream);
            err = "wavefront pipeline: live blocks left after the largest possible number of iterations";
            return RTR_ERR_DEVICE;
        }
        for (int b = 0; b < batch; ++b, ++iter) {
#define WF_EXTEND_LS(T)                                                                                      \
    do {                                                                                                     \
        if (rich)                                                                                            \
            hipLaunchKernelGGL((wf_extend_ls<T, true>), grid_shade, block, plan.lds, stream, sc, S, P, par);  \
        else                                                                                                 \
            hipLaunchKernelGGL((wf_extend_ls<T, false>), grid_shade, block, plan.lds, stream, sc, S, P, par); \
    } while (0)
            if (plan.machine) {
                if (rich)
                    hipLaunchKernelGGL(wf_extend<true>, grid_cast, block, plan.lds, stream, sc, S, P, par);
                else
                    hipLaunchKernelGGL(wf_extend<false>, grid_cast, block, plan.lds, stream, sc, S, P, par);
            } else if (plan.trav == RT_TRAV_PROGRAM) {
                WF_EXTEND_LS(RT_TRAV_PROGRAM_EXT);
            } else if (plan.trav == RT_TRAV_FAST) {
                WF_EXTEND_LS(RT_TRAV_FAST);
            } else {
                WF_EXTEND_LS(RT_TRAV_FLAT);
            }
#undef WF_EXTEND_LS
            ++n_launch;
            if (!split) {
                WF_SHADE(0);
            } else {
                WF_SHADE(1);
            }
            if (shadows) {
                if (plan.machine) {
                    if (plan.media)
                        hipLaunchKernelGGL(wf_connect<true>, grid_cast, block, plan.lds, stream, sc, S, P, par);
                    else
                        hipLaunchKernelGGL(wf_connect<false>, grid_cast, block, plan.lds, stream, sc, S, P, par);
                } else if (plan.trav == RT_TRAV_PROGRAM) {
                    hipLaunchKernelGGL(wf_connect_ls<RT_TRAV_PROGRAM_EXT>, grid_shade, block, plan.lds, stream, sc, S, P, par);
                } else if (plan.trav == RT_TRAV_FAST) {
                    hipLaunchKernelGGL(wf_connect_ls<RT_TRAV_FAST>, grid_shade, block, plan.lds, stream, sc, S, P, par);
                } else {
                    hipLaunchKernelGGL(wf_connect_ls<RT_TRAV_FLAT>, grid_shade, block, plan.lds, stream, sc, S, P, par);
                }
                ++n_launch;
            }
            if (split) WF_SHADE(2);
            hipLaunchKernelGGL(wf_compact, dim3(1), dim3(1024), 0, stream, S, par);
            ++n_launch;
            par ^= 1;
        }
        WF_HIP(hipGetLastError());
        WF_HIP(hipEventRecord(pool.ev[n_batches & 1], stream));
        ++n_batches;
        if (n_batches >= 2) { /* wait for the batch before the one just enqueued: the GPU still has a full batch queued */
            WF_HIP(hipEventSynchronize(pool.ev[n_batches & 1]));
            const uint32_t live = __atomic_load_n(pool.h_live, __ATOMIC_ACQUIRE);
            if (live == 0) break;
            /* blocks only die: the newest published count bounds every later list, so the launches shrink with it */
            grid_shade = dim3(std::min(std::min(live, (uint32_t)S.n_blocks), grid_cap));
        }
        if (cancelled_upto && cancelled_upto->load() >= P.render_id) {
            cancelled = true;
            break;
        }
    }
#undef WF_SHADE
#undef WF_SHADE_S
    hipLaunchKernelGGL(wf_finish, grid_all, block, 0, stream, S, P);
    ++n_launch;
#ifdef RTR_MACHINE_STATS
    {
        (void)hipStreamSynchronize(stream);
        MStats h{};
        (void)hipMemcpyFromSymbol(&h, HIP_SYMBOL(g_mstats), sizeof h);
        auto line = [](const char* n, unsigned long long t, unsigned long long l) {
            std::fprintf(stderr, "  %-6s trips %.4g  lanes/trip %.1f\n", n, (double)t, t ? (double)l / t : 0.0);
        };
        std::fprintf(stderr, "[machine stats, extend + connect]\n");
        line("tree", h.tree_trips, h.tree_lanes), line("inner", h.inner_trips, h.inner_lanes), line("leaf", h.leaf_trips, h.leaf_lanes);
        line("adv", h.adv_trips, h.adv_lanes), line("fin", h.fin_trips, h.fin_lanes), line("fetch", h.fetch_trips, h.fetch_lanes);
        MStats z{};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mstats), &z, sizeof z);
    }
#endif
    if (launches) *launches = n_launch;
    if (cancelled) return RTR_ERR_CANCELLED; /* unfinished pixels have no sum yet: the caller's buffer stays untouched */
    ResolveK R{P, d_rgb, (long long)row_stride, tile_done};
    rtr_launch_resolve(R, stream);
    ++n_launch;
    WF_HIP(hipGetLastError());
    if (launches) *launches = n_launch;
    return RTR_OK;
}
