/*
 * rt_machine.h -- the ray cast of the compiled scene (run_program / trace_fast of rt_device.h) as a
 * RESUMABLE per-lane state machine, for the persistent-threads stages of the wavefront pipeline.
 *
 * Why.  run_program() keeps the 64 lanes of a wave in lockstep through the steps of the ray-cast
 * program; inside a box tree every lane then waits for the slowest one (scene 9: 4.9 inner nodes and
 * 1.7 primitive tests per lane and tree, but 33 loop iterations per wave -- lane utilisation 26 %).
 * Here every lane carries its own position in the program and the wave alternates
 *
 *   tree phase     all lanes that are inside SOME box tree (not necessarily the same one) run the
 *                  while-while traversal loop together; the phase ends when enough lanes have left
 *                  their tree and wait for something else to do;
 *   service phase  the waiting lanes move on.  The program is flattened at upload into a list of
 *                  VISITS (struct FVisit: one instance of one step, in execution order) and advancing
 *                  is a data-driven loop -- "do the next visit" is the same instruction stream for
 *                  every lane wherever it stands in the program, the visit record comes through a
 *                  per-lane load -- that runs until each lane is inside a tree again or has finished
 *                  its cast.  Finished lanes hand their result to the stage (client.finish) and take
 *                  the NEXT RAY of the stage's queue (client.fetch): persistent threads.
 *
 * Lanes that are still inside a tree when the tree phase ends simply pause; their traversal state
 * (MLane) stays in registers.  A lane's arithmetic and its sequence of random draws are those of
 * run_program() / trace_fast(): same primitive tests in the same per-ray order, the constant_medium
 * draw under the same condition (constant_medium.h:62-103), so results are bit-identical
 * (tests: wavefront == megakernel == reference-order walk on every golden scene).
 *
 * Shadow rays (ANYHIT) may stop at their first hit once no medium is left to draw in the program
 * (step >= fstep_tail): nothing after that point can change the random sequence or the answer.
 */
#pragma once

#include "rt_device.h"

/* One instance visit of the ray-cast program, in execution order (built by rtr_upload_scene from
 * FStep / FSub / FInst).  A medium step is walked twice (constant_medium.h:62-66): pass 1 restarts at
 * `step_first`. */
struct FVisit {
    int32_t flags;
    int32_t step;       /* index into DScene::fstep (a medium hit is reported as its step) */
    int32_t inst;       /* index into DScene::finst (hit_inst of trace_fast; its box when FV_BOXES) */
    int32_t step_first; /* first visit of this step */
    int32_t xf_first, n_xf;
    int32_t ref_first, n_ref;
    int32_t bvh_root;   /* -1: scan the references linearly */
    float bound;
    double neg_inv_density; /* medium steps */
    int32_t pad[4];
};
static_assert(sizeof(FVisit) == 64, "FVisit is one 64-byte record");
#define FV_FIRST 1  /* first visit of its step */
#define FV_LAST 2   /* last visit of its step */
#define FV_MEDIUM 4 /* the step is a constant_medium (its sub-scene is the boundary) */
#define FV_BOXES 8  /* the step's sub-scene has more than RT_FAST_NO_BOX_MAX instances: test instance boxes */
#define FV_TAIL 16  /* step >= fstep_tail: no medium at or after this step */

enum { M_IDLE = 0, M_FETCH = 1, M_ADV = 2, M_TREE = 3, M_FIN = 4 };
/* mode bits next to the phase */
#define M_PASS1 8       /* second boundary cast of a medium step */
#define M_PASS_START 16 /* the next visit opens a pass (= one trace_fast() call of run_program()) */
#define M_AFTER_TREE 32 /* the visit's box tree has just been left */
#define M_EARLY 64      /* a shadow ray found its blocker where the first hit suffices */

struct MLane {
    /* the ray (world frame) and the closest hit of the program so far */
    V3 o, d;
    Real time;
    Real best_t; /* run_program()'s tmax */
    int best_ref, best_inst, best_med;
    uint32_t rng;
    int state; /* phase | mode bits */
    int j;     /* current visit */
    /* current pass = one trace_fast() call of run_program() */
    Real pass_lo, t1, t;
    int hit_ref, hit_inst, order;
    /* current instance with a box tree */
    V3 lo, ld;
    BoxRay br;
    float tmin_f, tmax_f;
    int sp, node;
    int slot, aux; /* what the client is working on (opaque to the machine) */
};
RT_DEV int m_phase(const MLane& m) { return m.state & 7; }
RT_DEV void m_set_phase(MLane& m, int phase) { m.state = phase; }
RT_DEV bool m_any_hit(const MLane& m) { return m.best_ref >= 0 || m.best_med >= 0; }

/* start a cast: ray (o, d, time) over [0.001, tmax] */
RT_DEV void m_begin(MLane& m, V3 o, V3 d, Real time, Real tmax, uint32_t rng) {
    m.o = o, m.d = d, m.time = time;
    m.best_t = tmax;
    m.best_ref = -1, m.best_inst = -1, m.best_med = -1;
    m.rng = rng;
    m.j = 0;
    m.state = M_ADV | M_PASS_START;
}

#ifdef RTR_MACHINE_STATS /* debug build: wave-level loop trips and the lanes that took part in them */
struct MStats {
    unsigned long long tree_trips, tree_lanes, inner_trips, inner_lanes, leaf_trips, leaf_lanes, adv_trips, adv_lanes,
        fin_trips, fin_lanes, fetch_trips, fetch_lanes;
};
__device__ MStats g_mstats;
#define M_STAT(field, pred)                                                                         \
    do {                                                                                            \
        const unsigned long long b_ = __builtin_amdgcn_ballot_w64(pred);                            \
        if (b_ && (threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) {       \
            atomicAdd(&g_mstats.field##_trips, 1ull);                                               \
            atomicAdd(&g_mstats.field##_lanes, (unsigned long long)__builtin_popcountll(b_));       \
        }                                                                                           \
    } while (0)
#else
#define M_STAT(field, pred) do { } while (0)
#endif

#define M_TMIN 0.001 /* both ray casts of the integrators start there (mis_path_integrator.h:37,213) */

/* a record of a read-only scene array through per-lane (divergent) loads */
template <class T>
RT_DEV T ld_lane(const T* base, int idx) {
    static_assert(sizeof(T) % 8 == 0, "records are multiples of 8 bytes");
    const __attribute__((address_space(1))) unsigned long long* src =
        (const __attribute__((address_space(1))) unsigned long long*)(unsigned long long)(base + idx);
    unsigned long long w[sizeof(T) / 8];
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 8; ++k) w[k] = src[k];
    T v;
    __builtin_memcpy(&v, w, sizeof(T));
    return v;
}

/* ---- service phase: advance every lane in M_ADV until it is inside a tree or its cast is finished ---- */
template <bool ANYHIT>
RT_DEV void m_advance(const DScene& sc, const FVisit* visits, const int n_visits, MLane& m) {
    while (m_phase(m) == M_ADV) {
        M_STAT(adv, true);
        if (m.j >= n_visits) { /* end of the program */
            m_set_phase(m, M_FIN);
            break;
        }
        const FVisit v = ld_lane(visits, m.j);
        const bool medium = (v.flags & FV_MEDIUM) != 0;
        /* first hit suffices: shadow ray in a geometry step with no medium at or after it */
        const bool first_hit_ends = ANYHIT && (v.flags & FV_TAIL) != 0;
        bool ended = (m.state & M_EARLY) != 0;
        if (!(m.state & M_AFTER_TREE)) {
            if (m.state & M_PASS_START) {
                /* a shadow ray that is blocked may stop once no medium is left to draw (run_program: `ANY && any`) */
                if (first_hit_ends && m_any_hit(m)) {
                    m_set_phase(m, M_FIN);
                    break;
                }
                if (!(m.state & M_PASS1)) m.pass_lo = medium ? -RT_INF : M_TMIN;
                m.t = medium ? RT_INF : m.best_t;
                m.hit_ref = -1, m.hit_inst = -1, m.order = -1;
                m.state &= ~M_PASS_START;
            }
            V3 lo = m.o, ld = m.d;
            bool skip = false;
            if (v.n_xf) {
                if (v.flags & FV_BOXES) {
                    const FInst I = ld_lane(sc.finst, v.inst);
                    const V3 inv = mk(1.0 / m.d.x, 1.0 / m.d.y, 1.0 / m.d.z);
                    Real tn;
                    skip = !box_enter(I.bmin, I.bmax, m.o, inv, m.pass_lo, m.t, tn);
                }
                for (int x = 0; x < v.n_xf; ++x) {
                    const FXf xf = ld_lane(sc.fxf, v.xf_first + x);
                    wrapper_enter(xf.type, xf.f, lo, ld);
                }
            }
            if (v.bvh_root < 0) {
                for (int q = 0; q < v.n_ref && !skip; ++q) {
                    Real t;
                    if (fast_ref_hit<true>(sc, v.ref_first + q, lo, ld, m.time, m.pass_lo, m.t, t, m.order)) {
                        m.t = t;
                        m.hit_ref = v.ref_first + q;
                        m.hit_inst = v.inst;
                        if (first_hit_ends) skip = true, ended = true;
                    }
                }
            } else if (!skip) {
                m.lo = lo, m.ld = ld;
                m.br = boxray_make(lo, ld, v.bound);
                m.tmin_f = float_below(m.pass_lo);
                m.tmax_f = float_above(m.t);
                m.sp = 0;
                m.node = v.bvh_root;
                m.state = (m.state & ~7) | M_TREE | M_AFTER_TREE;
                break;
            }
        }
        m.state &= ~(M_AFTER_TREE | M_EARLY);
        if (!(v.flags & FV_LAST) && !ended) {
            ++m.j;
            continue;
        }
        /* the pass is over (run_program(): what follows a trace_fast() call) */
        const bool h = m.hit_ref >= 0;
        int next = m.j + 1, mode = M_PASS_START;
        if (!medium) {
            if (h) m.best_t = m.t, m.best_ref = m.hit_ref, m.best_inst = m.hit_inst, m.best_med = -1;
            if (ended) { /* blocked, and nothing left to draw */
                m_set_phase(m, M_FIN);
                break;
            }
        } else if (h) {
            if (!(m.state & M_PASS1)) {
                m.t1 = m.t;
                m.pass_lo = m.t + 0.0001;
                next = v.step_first;
                mode = M_PASS_START | M_PASS1;
            } else { /* constant_medium.h:68-103 */
                Real t1 = m.t1, t2 = m.t;
                if (t1 < M_TMIN) t1 = M_TMIN;
                if (t2 > m.best_t) t2 = m.best_t;
                if (!(t1 >= t2)) {
                    if (t1 < 0) t1 = 0;
                    const Real ray_length = len(m.d);
                    const Real distance_inside_boundary = (t2 - t1) * ray_length;
                    const Real hit_distance = v.neg_inv_density * log(rng_next(m.rng));
                    if (!(hit_distance > distance_inside_boundary)) {
                        m.best_t = t1 + hit_distance / ray_length;
                        m.best_med = v.step, m.best_ref = -1, m.best_inst = -1;
                    }
                }
            }
        }
        m.j = next;
        m.state = M_ADV | mode;
    }
}

/* ---- tree phase --------------------------------------------------------------------------------
 * While-while traversal (trace_fast) for every lane whose phase is M_TREE, each in its own tree.
 * Ends when no lane is inside a tree any more, or when at least `wait_limit` lanes of the wave are
 * waiting for the service phase. */
template <bool ANYHIT>
RT_DEV void m_tree_phase(const DScene& sc, const FVisit* visits, MLane& m, const Stack st, const int wait_limit) {
    const bool mine = m_phase(m) == M_TREE;
    if (!__builtin_amdgcn_ballot_w64(mine)) return;
    const int vflags = mine ? as_const(&visits->flags)[(size_t)m.j * (sizeof(FVisit) / 4)] : 0;
    const int inst = mine ? as_const(&visits->inst)[(size_t)m.j * (sizeof(FVisit) / 4)] : 0;
    const bool first_hit_ends = ANYHIT && (vflags & FV_TAIL) != 0;
    int node = mine ? m.node : RT_BVH_DONE;
    int sp = m.sp;
    /* with few lanes left at the end of a queue the limit follows their number */
    const int busy = __builtin_popcountll(__builtin_amdgcn_ballot_w64(m_phase(m) != M_IDLE));
    const int limit = wait_limit < (busy + 1) / 2 ? wait_limit : (busy + 1) / 2;
    bool ended_early = false;
    while (true) {
        M_STAT(tree, node != RT_BVH_DONE);
        /* Descend.  A lane that has reached a leaf waits there; the leaves are served once no lane is at an
         * inner node any more or half of the lanes inside a tree wait at one -- so neither the box tests nor the
         * (expensive, double precision) primitive tests below run for a handful of lanes while the others idle. */
        for (;;) {
            const unsigned long long at_inner = __builtin_amdgcn_ballot_w64(node >= 0);
            if (!at_inner) break;
            const int n_leaf = __builtin_popcountll(__builtin_amdgcn_ballot_w64(node < 0 && node != RT_BVH_DONE));
            if (2 * n_leaf > __builtin_popcountll(at_inner) + n_leaf) break;
            if (node >= 0) {
                M_STAT(inner, true);
                const NodeRegs b = load_node(sc.fbvh, node);
                float tl, tr;
                const bool hl = boxray_hit(m.br, b.lmin, b.lmax, m.tmin_f, m.tmax_f, tl);
                const bool hr = boxray_hit(m.br, b.rmin, b.rmax, m.tmin_f, m.tmax_f, tr);
                const int cl = b.left, cr = b.right;
                if (hl && hr) { /* nearer child first */
                    const bool left_first = tl <= tr;
                    st.put(sp++, left_first ? cr : cl);
                    node = left_first ? cl : cr;
                } else if (hl) {
                    node = cl;
                } else if (hr) {
                    node = cr;
                } else {
                    node = sp > 0 ? st.get(--sp) : RT_BVH_DONE;
                }
            }
        }
        if (node < 0 && node != RT_BVH_DONE) {
            const int code = -1 - node;
            const int r0 = code >> 3, r1 = r0 + (code & 7) + 1;
            for (int r = r0; r < r1; ++r) {
                M_STAT(leaf, true);
                Real t;
                if (fast_ref_hit<true>(sc, r, m.lo, m.ld, m.time, m.pass_lo, m.t, t, m.order)) {
                    m.t = t;
                    m.tmax_f = float_above(t);
                    m.hit_ref = r;
                    m.hit_inst = inst;
                    if (first_hit_ends) ended_early = true;
                }
            }
            node = (sp > 0 && !ended_early) ? st.get(--sp) : RT_BVH_DONE;
        }
        const unsigned long long active = __builtin_amdgcn_ballot_w64(node != RT_BVH_DONE);
        if (!active) break;
        const int left_tree = __builtin_popcountll(__builtin_amdgcn_ballot_w64(mine && node == RT_BVH_DONE));
        if (left_tree >= limit) break;
    }
    if (mine) {
        m.node = node, m.sp = sp;
        if (node == RT_BVH_DONE) m.state = (m.state & ~7) | M_ADV | (ended_early ? M_EARLY : 0);
    }
}

/* ---- the machine ---------------------------------------------------------------------------------
 * Client (the stage kernel) supplies
 *     void finish(MLane& m)   called with the lanes in M_FIN active: consume the result, go M_FETCH
 *     void fetch(MLane& m)    called with the lanes in M_FETCH active: take the next ray of the queue and
 *                             m_begin() it, stay in M_FETCH to be asked again, or go M_IDLE (queue empty)
 * The loop returns when every lane is M_IDLE. */
#ifndef RTR_MACHINE_WAIT
#define RTR_MACHINE_WAIT 20 /* lanes of 64 that must be waiting before a tree phase is cut short */
#endif

template <bool ANYHIT, class Client>
RT_DEV void m_run(const DScene& sc, const FVisit* visits, const int n_visits, MLane& m, const Stack st, Client& client) {
    for (;;) {
        m_tree_phase<ANYHIT>(sc, visits, m, st, RTR_MACHINE_WAIT);
        for (;;) {
            m_advance<ANYHIT>(sc, visits, n_visits, m);
            if (m_phase(m) == M_FIN) {
                M_STAT(fin, true);
                client.finish(m);
            }
            while (__builtin_amdgcn_ballot_w64(m_phase(m) == M_FETCH))
                if (m_phase(m) == M_FETCH) {
                    M_STAT(fetch, true);
                    client.fetch(m);
                }
            if (!__builtin_amdgcn_ballot_w64(m_phase(m) == M_ADV)) break;
        }
        if (!__builtin_amdgcn_ballot_w64(m_phase(m) != M_IDLE)) break;
    }
}
