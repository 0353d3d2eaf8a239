/*
 * rt_machine.h -- the ray cast of the compiled scene (run_program / trace_fast of rt_device.h) as a
 * RESUMABLE per-lane state machine, for the persistent-threads stages of the wavefront pipeline.
 *
 * Why.  run_program() keeps the 64 lanes of a wave in lockstep through the steps of the ray-cast
 * program; inside a box tree every lane then waits for the slowest one (scene 9: 4.9 inner nodes and
 * 1.7 primitive tests per lane and tree, but 33 loop iterations per wave -- lane utilisation 26 %).
 * Here every lane carries its own program position (step, pass, instance) and the wave alternates
 *
 *   tree phase     all lanes that are inside SOME box tree (not necessarily the same one) run the
 *                  while-while traversal loop together; the phase ends when enough lanes have left
 *                  their tree and wait for something else to do;
 *   advance phase  the waiting lanes move on -- next instance, next pass, next step, result written,
 *                  NEXT RAY FETCHED from the stage's queue (persistent threads) -- until each of them
 *                  is inside a tree again or the queue is empty.  Lanes at the same position are
 *                  served together (a waterfall over positions), so instance / step records still
 *                  come in through scalar loads exactly as in run_program().
 *
 * Lanes that are still inside a tree when the tree phase ends simply pause; their traversal state
 * (MLane) stays in registers.  A lane's arithmetic and its sequence of random draws are those of
 * run_program() / trace_fast(): same primitive tests in the same per-ray order, the constant_medium
 * draw under the same condition (constant_medium.h:62-103), so results are bit-identical
 * (tests: machine == run_program == reference-order walk on the golden hit vectors).
 *
 * Shadow rays (ANYHIT) may stop at their first hit once no medium is left to draw in the program
 * (step >= fstep_tail): nothing after that point can change the random sequence or the answer.
 */
#pragma once

#include "rt_device.h"

enum { M_IDLE = 0, M_FETCH = 1, M_PASS = 2, M_INST = 3, M_TREE = 4, M_END = 5, M_FINISH = 6 };
/* position word: phase (3 bits) | pass (1 bit) | step (12 bits) | instance index into DScene::finst (16 bits) */
RT_DEV int m_pos(int phase, int k, int pass, int ii) { return phase | (pass << 3) | (k << 4) | (ii << 16); }
RT_DEV int m_phase(int pos) { return pos & 7; }
RT_DEV int m_pass(int pos) { return (pos >> 3) & 1; }
RT_DEV int m_step(int pos) { return (pos >> 4) & 0xFFF; }
RT_DEV int m_inst(int pos) { return (pos >> 16) & 0xFFFF; }
#define M_MAX_STEPS 4096
#define M_MAX_INSTANCES 65536

struct MLane {
    /* the ray (world frame) and the closest hit of the program so far */
    V3 o, d;
    Real time;
    Real best_t; /* run_program()'s tmax */
    int best_ref, best_inst, best_med;
    uint32_t rng;
    int pos;
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

RT_DEV bool m_any_hit(const MLane& m) { return m.best_ref >= 0 || m.best_med >= 0; }

/* start a cast: ray (o, d, time) over [0.001, tmax] */
RT_DEV void m_begin(MLane& m, V3 o, V3 d, Real time, Real tmax, uint32_t rng) {
    m.o = o, m.d = d, m.time = time;
    m.best_t = tmax;
    m.best_ref = -1, m.best_inst = -1, m.best_med = -1;
    m.rng = rng;
    m.pos = m_pos(M_PASS, 0, 0, 0);
}

#define M_TMIN 0.001 /* both ray casts of the integrators start there (mis_path_integrator.h:37,213) */

/* ---- advance phase: one position, uniform over the participating lanes ---------------------- */
template <bool ANYHIT>
RT_DEV void m_step_pass(const DScene& sc, MLane& m, const int k, const int pass) {
    if (k >= sc.n_fstep) {
        m.pos = M_FINISH;
        return;
    }
    /* a shadow ray that is blocked may stop once no medium is left to draw (run_program: `ANY && any`) */
    if (ANYHIT && k >= sc.fstep_tail && m_any_hit(m)) {
        m.pos = M_FINISH;
        return;
    }
    const FStep step = ld_const(sc.fstep, k);
    const FSub sub = ld_const(sc.fsub, step.sub);
    const bool medium = step.kind != 0;
    if (pass == 0) m.pass_lo = medium ? -RT_INF : M_TMIN;
    m.t = medium ? RT_INF : m.best_t;
    m.hit_ref = -1, m.hit_inst = -1, m.order = -1;
    m.pos = m_pos(M_INST, k, pass, sub.inst_first);
}

template <bool ANYHIT>
RT_DEV void m_step_inst(const DScene& sc, MLane& m, const int k, const int pass, const int ii) {
    const FStep step = ld_const(sc.fstep, k);
    const FSub sub = ld_const(sc.fsub, step.sub);
    if (ii >= sub.inst_first + sub.n_inst) {
        m.pos = m_pos(M_END, k, pass, 0);
        return;
    }
    /* first hit suffices: shadow ray, geometry step, no medium at or after this step */
    const bool first_hit_ends = ANYHIT && step.kind == 0 && k >= sc.fstep_tail;
    const FInst I = ld_const(sc.finst, ii);
    const bool use_boxes = sub.n_inst > RT_FAST_NO_BOX_MAX;
    V3 lo = m.o, ld = m.d;
    bool skip = false;
    const int n_xf = I.n_xf;
    if (n_xf) {
        if (use_boxes) {
            const V3 inv = mk(1.0 / m.d.x, 1.0 / m.d.y, 1.0 / m.d.z);
            Real tn;
            skip = !box_enter(I.bmin, I.bmax, m.o, inv, m.pass_lo, m.t, tn);
        }
        for (int x = 0; x < n_xf; ++x) {
            const FXf xf = ld_const(sc.fxf, I.xf_first + x);
            wrapper_enter(xf.type, xf.f, lo, ld);
        }
    }
    int next = m_pos(M_INST, k, pass, ii + 1);
    if (I.bvh_root < 0) {
        const int r0 = I.ref_first, r1 = r0 + I.n_ref;
        for (int r = r0; r < r1; ++r) {
            Real t;
            if (!skip && fast_ref_hit<true>(sc, r, lo, ld, m.time, m.pass_lo, m.t, t, m.order)) {
                m.t = t;
                m.hit_ref = r;
                m.hit_inst = ii;
                if (first_hit_ends) skip = true, next = m_pos(M_END, k, pass, 0);
            }
        }
    } else if (!skip) {
        m.lo = lo, m.ld = ld;
        m.br = boxray_make(lo, ld, I.bound);
        m.tmin_f = float_below(m.pass_lo);
        m.tmax_f = float_above(m.t);
        m.sp = 0;
        m.node = I.bvh_root;
        next = m_pos(M_TREE, k, pass, ii);
    }
    m.pos = next;
}

RT_DEV void m_step_end(const DScene& sc, MLane& m, const int k, const int pass) {
    const FStep step = ld_const(sc.fstep, k);
    const bool medium = step.kind != 0;
    const bool h = m.hit_ref >= 0;
    int next = m_pos(M_PASS, k + 1, 0, 0);
    if (!medium) {
        if (h) m.best_t = m.t, m.best_ref = m.hit_ref, m.best_inst = m.hit_inst, m.best_med = -1;
    } else if (h) {
        if (pass == 0) {
            m.t1 = m.t;
            m.pass_lo = m.t + 0.0001;
            next = m_pos(M_PASS, k, 1, 0);
        } else { /* constant_medium.h:68-103 */
            Real t1 = m.t1, t2 = m.t;
            if (t1 < M_TMIN) t1 = M_TMIN;
            if (t2 > m.best_t) t2 = m.best_t;
            if (!(t1 >= t2)) {
                if (t1 < 0) t1 = 0;
                const Real ray_length = len(m.d);
                const Real distance_inside_boundary = (t2 - t1) * ray_length;
                const Real hit_distance = step.neg_inv_density * log(rng_next(m.rng));
                if (!(hit_distance > distance_inside_boundary)) {
                    m.best_t = t1 + hit_distance / ray_length;
                    m.best_med = k, m.best_ref = -1, m.best_inst = -1;
                }
            }
        }
    }
    m.pos = next;
}

/* ---- tree phase --------------------------------------------------------------------------------
 * While-while traversal (trace_fast) for every lane whose phase is M_TREE, each in its own tree.
 * Ends when no lane is inside a tree any more, or when at least `wait_limit` lanes of the wave are
 * waiting for the advance phase (lanes that have left their tree + those that were waiting before). */
template <bool ANYHIT>
RT_DEV void m_tree_phase(const DScene& sc, MLane& m, const Stack st, const int wait_limit) {
    const bool mine = m_phase(m.pos) == M_TREE;
    if (!__builtin_amdgcn_ballot_w64(mine)) return;
    const int k = m_step(m.pos), ii = m_inst(m.pos);
    /* a medium's step index is below fstep_tail, so `k >= fstep_tail` also says "geometry step" */
    const bool first_hit_ends = ANYHIT && k >= sc.fstep_tail;
    int node = mine ? m.node : RT_BVH_DONE;
    int sp = m.sp;
    const int idle_before = __builtin_popcountll(__builtin_amdgcn_ballot_w64(!mine && m_phase(m.pos) != M_IDLE));
    /* with few lanes left at the end of a queue the limit follows their number */
    const int busy = __builtin_popcountll(__builtin_amdgcn_ballot_w64(m_phase(m.pos) != M_IDLE));
    const int limit = wait_limit < (busy + 2) / 3 ? wait_limit : (busy + 2) / 3;
    bool ended_early = false;
    while (true) {
        while (node >= 0) {
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
        if (node != RT_BVH_DONE) {
            const int code = -1 - node;
            const int r0 = code >> 3, r1 = r0 + (code & 7) + 1;
            for (int r = r0; r < r1; ++r) {
                Real t;
                if (fast_ref_hit<true>(sc, r, m.lo, m.ld, m.time, m.pass_lo, m.t, t, m.order)) {
                    m.t = t;
                    m.tmax_f = float_above(t);
                    m.hit_ref = r;
                    m.hit_inst = ii;
                    if (first_hit_ends) ended_early = true;
                }
            }
            node = (sp > 0 && !ended_early) ? st.get(--sp) : RT_BVH_DONE;
        }
        const unsigned long long active = __builtin_amdgcn_ballot_w64(node != RT_BVH_DONE);
        if (!active) break;
        const int left_tree = __builtin_popcountll(__builtin_amdgcn_ballot_w64(mine && node == RT_BVH_DONE));
        if (idle_before + left_tree >= limit) break;
    }
    if (mine) {
        m.node = node, m.sp = sp;
        if (node == RT_BVH_DONE)
            m.pos = ended_early ? m_pos(M_END, k, m_pass(m.pos), 0) : m_pos(M_INST, k, m_pass(m.pos), ii + 1);
    }
}

/* ---- the machine ---------------------------------------------------------------------------------
 * Client (the stage kernel) supplies
 *     void fetch(MLane& m)    lanes in M_FETCH: take the next ray of the queue and m_begin() it, stay in
 *                             M_FETCH to be asked again, or go M_IDLE when the queue is empty (uniform call)
 *     void finish(MLane& m)   lanes in M_FINISH: consume the result, then M_FETCH
 * The loop returns when every lane is M_IDLE. */
#ifndef RTR_MACHINE_WAIT
#define RTR_MACHINE_WAIT 20 /* lanes of 64 that must be waiting before a tree phase is cut short */
#endif

template <bool ANYHIT, class Client>
RT_DEV void m_run(const DScene& sc, MLane& m, const Stack st, Client& client) {
    for (;;) {
        m_tree_phase<ANYHIT>(sc, m, st, RTR_MACHINE_WAIT);
        /* advance phase: serve positions one at a time until every lane is in a tree or idle */
        for (;;) {
            const int ph = m_phase(m.pos);
            const bool waiting = ph != M_TREE && ph != M_IDLE;
            const unsigned long long todo = __builtin_amdgcn_ballot_w64(waiting);
            if (!todo) break;
            const int upos = __builtin_amdgcn_readlane(m.pos, __builtin_ctzll(todo));
            if (m.pos == upos) { /* uniform position: the records below come through scalar loads */
                const int uph = m_phase(upos), k = m_step(upos), pass = m_pass(upos), ii = m_inst(upos);
                if (uph == M_INST)
                    m_step_inst<ANYHIT>(sc, m, k, pass, ii);
                else if (uph == M_PASS)
                    m_step_pass<ANYHIT>(sc, m, k, pass);
                else if (uph == M_END)
                    m_step_end(sc, m, k, pass);
                else if (uph == M_FINISH)
                    client.finish(m);
                else
                    client.fetch(m);
            }
        }
        if (!__builtin_amdgcn_ballot_w64(m_phase(m.pos) != M_IDLE)) break;
    }
}
