/*
 * rt_compile.h -- host side: lower a validated hittable graph (no constant_medium) to the
 * compiled scene of rt_device.h (instances / references / wrapper epilogue lists / box trees).
 * See the comment above `struct FInst` for why this is result-preserving.
 */
#pragma once

#include "rt_device.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <vector>

struct CompiledScene {
    std::vector<FSub> subs;
    std::vector<rtr_node> dev_nodes; /* node array the device walks: subtrees replaced by RT_NODE_COMPILED */
    int n_compiled_subtrees = 0;
    std::vector<FInst> inst;
    std::vector<FXf> xf;
    std::vector<FRef> ref;
    std::vector<int32_t> exits;
    std::vector<FBvh> bvh;
    int stack_words = 1;
    bool ok = false;
};

namespace rtc {

constexpr int kLinearMax = 12; /* instances with more references get a box tree */
constexpr int kLeafMax = 4;
constexpr int kMaxInstances = 64;

struct Box {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    void grow(const double* p) {
        for (int c = 0; c < 3; ++c) lo[c] = std::min(lo[c], p[c]), hi[c] = std::max(hi[c], p[c]);
    }
    void grow(const Box& b) {
        grow(b.lo);
        grow(b.hi);
    }
    void pad() { /* conservative: the boxes only prune */
        for (int c = 0; c < 3; ++c) {
            const double e = 1e-4 + 1e-9 * std::max(std::fabs(lo[c]), std::fabs(hi[c]));
            lo[c] -= e, hi[c] += e;
        }
    }
};

struct PendingRef {
    int node;
    std::vector<int> wrappers; /* outermost first, as met on the way down */
    Box local;
    int order; /* position of the primitive's LAST visit in the reference's walk (tie-breaks equal t) */
};

struct Builder {
    const rtr_scene_desc* s;
    CompiledScene& out;
    int inst_base = 0;
    explicit Builder(CompiledScene& o) : out(o) {}
    double t_lo, t_hi; /* ray times the camera and the shadow rays can carry */
    std::vector<int> chain;    /* translate / rotate_y node indices, outermost first */
    std::vector<int> wrappers; /* the same plus flip_face */
    std::map<std::vector<uint64_t>, int> inst_of_chain;
    std::vector<std::vector<PendingRef>> pending; /* per instance */
    std::map<std::pair<int, std::vector<uint64_t>>, std::pair<int, int>> seen; /* -> (instance, slot) */
    int visit_counter = 0;
    bool too_complex = false;

    static uint64_t bits(double v) {
        uint64_t u;
        std::memcpy(&u, &v, 8);
        return u;
    }
    std::vector<uint64_t> chain_key() const {
        std::vector<uint64_t> k;
        for (int w : chain) {
            const rtr_node& n = s->nodes[w];
            k.push_back((uint64_t)n.type);
            k.push_back(bits(n.f[0])), k.push_back(bits(n.f[1])), k.push_back(bits(n.f[2]));
        }
        return k;
    }
    std::vector<uint64_t> full_key() const { /* flips matter for the hit record */
        std::vector<uint64_t> k = chain_key();
        k.push_back(~0ull);
        for (int w : wrappers) k.push_back((uint64_t)s->nodes[w].type);
        return k;
    }

    Box prim_box(const rtr_node& n) const {
        Box b;
        switch (n.type) {
        case RTR_NODE_SPHERE: {
            const double r = std::fabs(n.f[3]);
            double lo[3] = {n.f[0] - r, n.f[1] - r, n.f[2] - r}, hi[3] = {n.f[0] + r, n.f[1] + r, n.f[2] + r};
            b.grow(lo), b.grow(hi);
            break;
        }
        case RTR_NODE_MOVING_SPHERE: {
            const double r = std::fabs(n.f[8]);
            for (double t : {t_lo, t_hi}) {
                const double a = (t - n.f[6]) / (n.f[7] - n.f[6]);
                for (int c = 0; c < 3; ++c) {
                    const double ctr = n.f[c] + a * (n.f[3 + c] - n.f[c]);
                    b.lo[c] = std::min(b.lo[c], ctr - r), b.hi[c] = std::max(b.hi[c], ctr + r);
                }
            }
            break;
        }
        default: { /* rects: f = a0 a1 b0 b1 k */
            const int ka = n.type == RTR_NODE_XY_RECT ? 2 : (n.type == RTR_NODE_XZ_RECT ? 1 : 0);
            const int aa = n.type == RTR_NODE_YZ_RECT ? 1 : 0;
            const int ba = n.type == RTR_NODE_XY_RECT ? 1 : 2;
            b.lo[ka] = n.f[4], b.hi[ka] = n.f[4];
            b.lo[aa] = std::min(n.f[0], n.f[1]), b.hi[aa] = std::max(n.f[0], n.f[1]);
            b.lo[ba] = std::min(n.f[2], n.f[3]), b.hi[ba] = std::max(n.f[2], n.f[3]);
        }
        }
        b.pad();
        return b;
    }

    void walk(int ix) {
        if (too_complex) return;
        const rtr_node& n = s->nodes[ix];
        switch (n.type) {
        case RTR_NODE_BVH:
            walk(n.a);
            if (n.b != n.a) walk(n.b);
            break;
        case RTR_NODE_LIST:
            for (int k = 0; k < n.b; ++k) walk(s->list_children[n.a + k]);
            break;
        case RTR_NODE_TRANSLATE:
        case RTR_NODE_ROTATE_Y:
            chain.push_back(ix), wrappers.push_back(ix);
            walk(n.a);
            chain.pop_back(), wrappers.pop_back();
            break;
        case RTR_NODE_FLIP_FACE:
            wrappers.push_back(ix);
            walk(n.a);
            wrappers.pop_back();
            break;
        case RTR_NODE_MEDIUM: too_complex = true; break;
        default: {
            const int order = visit_counter++;
            auto sk = std::make_pair(ix, full_key());
            auto seen_it = seen.find(sk);
            if (seen_it != seen.end()) { /* same primitive, same frame: one test is enough; the later visit wins ties */
                pending[seen_it->second.first][seen_it->second.second].order = order;
                break;
            }
            const std::vector<uint64_t> key = chain_key();
            auto it = inst_of_chain.find(key);
            int ii;
            if (it == inst_of_chain.end()) {
                ii = (int)out.inst.size();
                if (ii - inst_base >= kMaxInstances) {
                    too_complex = true;
                    break;
                }
                inst_of_chain[key] = ii;
                FInst I{};
                I.xf_first = (int)out.xf.size();
                I.n_xf = (int)chain.size();
                for (int w : chain) {
                    const rtr_node& wn = s->nodes[w];
                    FXf x{};
                    x.type = wn.type;
                    x.f[0] = wn.f[0], x.f[1] = wn.f[1], x.f[2] = wn.f[2];
                    out.xf.push_back(x);
                }
                I.bvh_root = -1;
                out.inst.push_back(I);
                pending.emplace_back();
            } else {
                ii = it->second;
            }
            seen[sk] = {ii - inst_base, (int)pending[ii - inst_base].size()};
            pending[ii - inst_base].push_back(PendingRef{ix, wrappers, prim_box(n), order});
        }
        }
    }

    /* local point -> world: the wrappers' epilogues on points, innermost first */
    void to_world(const FInst& I, double* p) const {
        for (int k = I.n_xf - 1; k >= 0; --k) {
            const FXf& x = out.xf[I.xf_first + k];
            if (x.type == RTR_NODE_TRANSLATE) {
                p[0] += x.f[0], p[1] += x.f[1], p[2] += x.f[2];
            } else {
                const double s_ = x.f[0], c = x.f[1], px = p[0], pz = p[2];
                p[0] = c * px + s_ * pz;
                p[2] = -s_ * px + c * pz;
            }
        }
    }

    int build_tree(std::vector<PendingRef>& refs, int lo, int hi, int depth, int& max_depth) {
        max_depth = std::max(max_depth, depth);
        Box b;
        for (int k = lo; k < hi; ++k) b.grow(refs[k].local);
        const int me = (int)out.bvh.size();
        out.bvh.push_back(FBvh{});
        FBvh node{};
        for (int c = 0; c < 3; ++c) node.bmin[c] = b.lo[c], node.bmax[c] = b.hi[c];
        if (hi - lo <= kLeafMax) {
            node.left = lo; /* patched to the global reference index by the caller */
            node.right = -(hi - lo);
        } else {
            int axis = 0;
            double ext = -1;
            Box cb;
            for (int k = lo; k < hi; ++k) {
                double ctr[3];
                for (int c = 0; c < 3; ++c) ctr[c] = 0.5 * (refs[k].local.lo[c] + refs[k].local.hi[c]);
                cb.grow(ctr);
            }
            for (int c = 0; c < 3; ++c)
                if (cb.hi[c] - cb.lo[c] > ext) ext = cb.hi[c] - cb.lo[c], axis = c;
            const int mid = (lo + hi) / 2;
            std::nth_element(refs.begin() + lo, refs.begin() + mid, refs.begin() + hi,
                             [axis](const PendingRef& x, const PendingRef& y) {
                                 return x.local.lo[axis] + x.local.hi[axis] < y.local.lo[axis] + y.local.hi[axis];
                             });
            node.left = build_tree(refs, lo, mid, depth + 1, max_depth);
            node.right = build_tree(refs, mid, hi, depth + 1, max_depth);
        }
        out.bvh[me] = node;
        return me;
    }

    /* compile the subtree under `root`; returns the sub-scene index or -1 (media / too fragmented) */
    int run(int root) {
        /* camera rays carry a time in [time0, time1], shadow rays time 0 (mis_path_integrator.h:210) */
        t_lo = std::min(0.0, std::min(s->camera.time0, s->camera.time1));
        t_hi = std::max(0.0, std::max(s->camera.time0, s->camera.time1));
        const size_t mark[6] = {out.inst.size(), out.xf.size(), out.ref.size(), out.exits.size(), out.bvh.size(), 0};
        inst_base = (int)out.inst.size();
        walk(root);
        if (too_complex || (int)out.inst.size() == inst_base) { /* undo */
            out.inst.resize(mark[0]), out.xf.resize(mark[1]), out.ref.resize(mark[2]), out.exits.resize(mark[3]);
            out.bvh.resize(mark[4]);
            return -1;
        }
        int stack = 1;
        for (size_t ii = inst_base; ii < out.inst.size(); ++ii) {
            FInst& I = out.inst[ii];
            std::vector<PendingRef>& refs = pending[ii - inst_base];
            I.ref_first = (int)out.ref.size();
            I.n_ref = (int)refs.size();
            const int first_bvh = (int)out.bvh.size();
            if ((int)refs.size() > kLinearMax) {
                int depth = 0;
                I.bvh_root = build_tree(refs, 0, (int)refs.size(), 1, depth);
                for (size_t b = first_bvh; b < out.bvh.size(); ++b)
                    if (out.bvh[b].right < 0) out.bvh[b].left += I.ref_first;
                stack = std::max(stack, depth + 2);
            }
            Box local;
            for (const PendingRef& r : refs) {
                FRef fr{};
                fr.node = r.node;
                fr.exit_first = (int)out.exits.size();
                fr.n_exit = (int)r.wrappers.size();
                fr.pad = r.order;
                for (auto it = r.wrappers.rbegin(); it != r.wrappers.rend(); ++it) out.exits.push_back(*it);
                out.ref.push_back(fr);
                local.grow(r.local);
            }
            Box world;
            for (int corner = 0; corner < 8; ++corner) {
                double p[3] = {corner & 1 ? local.hi[0] : local.lo[0], corner & 2 ? local.hi[1] : local.lo[1],
                               corner & 4 ? local.hi[2] : local.lo[2]};
                to_world(I, p);
                world.grow(p);
            }
            world.pad();
            for (int c = 0; c < 3; ++c) I.bmin[c] = world.lo[c], I.bmax[c] = world.hi[c];
        }
        out.stack_words = std::max(out.stack_words, stack);
        FSub sub{};
        sub.inst_first = inst_base;
        sub.n_inst = (int)out.inst.size() - inst_base;
        out.subs.push_back(sub);
        return (int)out.subs.size() - 1;
    }
};

/* per node: primitives below it and whether a constant_medium is below it (DAG-aware) */
struct SubtreeFacts {
    std::vector<int> prims;
    std::vector<char> media, done;
    const rtr_scene_desc* s;
    void visit(int ix) {
        if (done[ix]) return;
        done[ix] = 1;
        const rtr_node& n = s->nodes[ix];
        int p = 0;
        char m = 0;
        auto child = [&](int c) {
            visit(c);
            p += prims[c];
            m |= media[c];
        };
        switch (n.type) {
        case RTR_NODE_BVH:
            child(n.a);
            if (n.b != n.a) child(n.b);
            break;
        case RTR_NODE_LIST:
            for (int k = 0; k < n.b; ++k) child(s->list_children[n.a + k]);
            break;
        case RTR_NODE_TRANSLATE:
        case RTR_NODE_ROTATE_Y:
        case RTR_NODE_FLIP_FACE: child(n.a); break;
        case RTR_NODE_MEDIUM:
            child(n.a);
            m = 1;
            break;
        default: p = 1;
        }
        prims[ix] = p;
        media[ix] = m;
    }
};

} // namespace rtc

/* `scene` must have passed validation.
 *  - no media: sub-scene 0 = the whole scene (ok = true), device nodes = the scene's nodes;
 *  - media: ok = false (the reference-order walk stays in charge), but every media-free subtree
 *    with at least kMinCompiled primitives met on the way down from the root is compiled and its
 *    node replaced by RT_NODE_COMPILED in the device copy of the node array. */
inline CompiledScene compile_scene(const rtr_scene_desc* scene, bool has_media) {
    CompiledScene cs;
    cs.dev_nodes.assign(scene->nodes, scene->nodes + scene->n_nodes);
    if (!has_media) {
        rtc::Builder b(cs);
        b.s = scene;
        cs.ok = b.run(scene->root) == 0;
        if (!cs.ok) {
            cs.subs.clear();
            cs.stack_words = 1;
        }
        return cs;
    }
    constexpr int kMinCompiled = 8;
    rtc::SubtreeFacts facts;
    facts.s = scene;
    facts.prims.assign(scene->n_nodes, 0);
    facts.media.assign(scene->n_nodes, 0);
    facts.done.assign(scene->n_nodes, 0);
    facts.visit(scene->root);
    std::vector<char> seen(scene->n_nodes, 0);
    std::vector<int> todo{scene->root};
    while (!todo.empty()) {
        const int ix = todo.back();
        todo.pop_back();
        if (seen[ix]) continue;
        seen[ix] = 1;
        const rtr_node& n = scene->nodes[ix];
        if (!facts.media[ix]) {
            if (facts.prims[ix] >= kMinCompiled) {
                rtc::Builder b(cs);
                b.s = scene;
                const int sub = b.run(ix);
                if (sub >= 0) {
                    rtr_node c{};
                    c.type = RT_NODE_COMPILED;
                    c.a = sub;
                    cs.dev_nodes[ix] = c;
                    ++cs.n_compiled_subtrees;
                }
            }
            continue; /* small media-free subtrees stay interpreted */
        }
        switch (n.type) {
        case RTR_NODE_BVH: todo.push_back(n.a), todo.push_back(n.b); break;
        case RTR_NODE_LIST:
            for (int k = 0; k < n.b; ++k) todo.push_back(scene->list_children[n.a + k]);
            break;
        case RTR_NODE_TRANSLATE:
        case RTR_NODE_ROTATE_Y:
        case RTR_NODE_FLIP_FACE: todo.push_back(n.a); break;
        default: break; /* a medium: its boundary is cast t-only by the walk */
        }
    }
    cs.ok = false;
    return cs;
}
