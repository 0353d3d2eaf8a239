/*
 * rt_compile.h -- host side: lower a validated hittable graph to the compiled scene of rt_device.h
 * (instances / references / wrapper epilogue lists / SAH box trees with single-precision boxes).
 *   - no order-sensitive part: sub-scene 0 = the whole graph (RT_TRAV_FAST / RT_TRAV_FLAT);
 *   - constant_media under bvh_nodes / lists: the step program (FStep, RT_TRAV_PROGRAM);
 *   - anything else that is order-sensitive (a medium under a transform, a sphere with a negative
 *     radius): the reference-order walk with every large media-free subtree compiled (RT_NODE_COMPILED).
 * See the comments above `struct FInst`, `struct FStep` and `struct FBvh` for why each is result-preserving.
 */
#pragma once

#include "rt_device.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
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
    std::vector<double> scan; /* packed geometry of the linearly scanned instances (FInst::scan_first / run) */
    std::vector<FStep> steps; /* ray-cast program of a scene with media (empty: none could be built) */
    std::vector<FGuard> guards; /* bvh_node boxes above the guarded primitives (FStep kind 3, RT_GUARD_FLAG references) */
    std::map<int, std::pair<int, int>> guard_of_ref; /* reference -> (first guard, count): RT_GUARD_FLAG references */
    int step_tail = 0;
    int stack_words = 1;
    bool ok = false;
};

namespace rtc {

/* FInst::scan_first / run[] of every linearly scanned instance; `prims` = the per-reference node records with their tie
 * flags (rtr_upload_scene).  Instances whose references do not fit RT_INST_RUNS_MAX runs, or that hold a tie-capable
 * reference (its visiting position takes part in the test), keep the generic loop. */
inline void build_scan_runs(CompiledScene& cs, const std::vector<rtr_node>& prims);
/* FLeaf record of every reference */
inline std::vector<FLeaf> build_leaf_records(const CompiledScene& cs, const std::vector<rtr_node>& prims);

constexpr int kLinearMax = 12; /* instances with more references get a box tree */
constexpr int kLeafMax = 4;
constexpr int kGroupMax = 8; /* = the most references a leaf link can name (FBvh) */
constexpr int kMaxInstances = 1 << 18; /* a cost guard, not a correctness limit (a top tree makes many instances cheap;
                                          the traversal machine's visit list is still one entry per instance) */
/* transformed instances from which a sub-scene gets a top tree (FSub::top_root): below, the wave-uniform scan of every
 * instance behind its world box is as fast or faster than a per-lane walk (tools/time_random.py, one MI355X, top tree
 * against scan: RR integrator 1.03x at 4 instances, 1.14x at 30, 1.29x at 52, 2.1x at 305; MIS 0.74x at 6, 0.85x at 30,
 * 1.0x at 52, 1.43x at 305 -- the MIS kernel of a scene without box trees is the flat one with shared reciprocals).
 * RTR_TOP_MIN, read at every upload, overrides it (sweeps, tests). */
inline int top_tree_min() {
    const char* e = std::getenv("RTR_TOP_MIN");
    return e && *e ? std::atoi(e) : 40;
}
constexpr int kMaxTreeDepth = 56; /* bounds the per-lane LDS stack of the box-tree traversal */

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
    int group; /* primitives of one small hittable_list (a `box` = 6 rects) stay together in one tree leaf; -1: none */
    std::vector<int> guards; /* guard mode, hollow spheres: the bvh_nodes above it, root first */
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
    int current_group = -1, group_counter = 0;
    bool too_complex = false;
    /* Guard mode: a graph with hollow spheres (negative radius: inverted bounding box, sphere.h:62-66) and nothing else
     * that depends on the visiting order.  The reference reaches such a sphere only through rays that pass the boxes of
     * the bvh_nodes above it within [t_min, closest t so far] (bvh.h:40-50).  If everything sits under ONE transform
     * chain (the empty one) and is scanned linearly, the scan order IS the reference's visiting order, so the running
     * t_max at the sphere's reference is the reference's "closest so far" and the box tests can be run right there
     * (RT_GUARD_FLAG).  Anything else (several instances, too many references for a linear scan, a shared object whose
     * visits lie on both sides of a hollow sphere) makes run() fail: the caller falls back to the step program. */
    bool guard_mode = false, guard_fail = false;
    int hollows_seen = 0;
    std::vector<int> bvh_stack;
    std::map<std::pair<int, std::vector<uint64_t>>, int> hollows_at_first_visit;
    static constexpr int kGuardLinearMax = 256;

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
            bvh_stack.push_back(ix);
            walk(n.a);
            if (n.b != n.a) walk(n.b);
            bvh_stack.pop_back();
            break;
        case RTR_NODE_LIST: {
            /* a short list of primitives (geometry/box.h: 6 rects) is one object for the box tree */
            bool small = n.b >= 2 && n.b <= kGroupMax && current_group < 0;
            for (int k = 0; k < n.b && small; ++k) small = s->nodes[s->list_children[n.a + k]].type >= RTR_NODE_SPHERE;
            if (small) current_group = group_counter++;
            for (int k = 0; k < n.b; ++k) walk(s->list_children[n.a + k]);
            if (small) current_group = -1;
            break;
        }
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
            const bool hollow = (n.type == RTR_NODE_SPHERE && n.f[3] < 0) || (n.type == RTR_NODE_MOVING_SPHERE && n.f[8] < 0);
            if (seen_it != seen.end()) { /* same primitive, same frame: one test is enough; the later visit wins ties */
                pending[seen_it->second.first][seen_it->second.second].order = order;
                /* (scanned at its first visit: a hollow sphere between its visits, or a hollow sphere visited twice
                 * under different boxes, would see another "closest so far" than the reference) */
                if (guard_mode && (hollow || hollows_at_first_visit[sk] != hollows_seen)) guard_fail = true;
                break;
            }
            if (guard_mode) {
                hollows_at_first_visit[sk] = hollows_seen;
                if (hollow) {
                    ++hollows_seen;
                    if (!chain.empty() || n.type != RTR_NODE_SPHERE) guard_fail = true; /* boxes above a transform; f[] full */
                }
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
            pending[ii - inst_base].push_back(PendingRef{ix, wrappers, prim_box(n), order, current_group,
                                                         guard_mode && hollow ? bvh_stack : std::vector<int>()});
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

    /* outward rounding of a box plane to single precision */
    static float float_down(double v) {
        float f = (float)v;
        if ((double)f > v) f = std::nextafterf(f, -INFINITY);
        return std::nextafterf(f, -INFINITY);
    }
    static float float_up(double v) {
        float f = (float)v;
        if ((double)f < v) f = std::nextafterf(f, INFINITY);
        return std::nextafterf(f, INFINITY);
    }
    static double half_area(const Box& b) {
        const double x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
        return x * y + y * z + z * x;
    }

    /* One unit of a sub-scene's top tree: a transformed instance, or one leaf of the untransformed instance's tree */
    struct TopItem {
        Box box;
        int link;
        double weight; /* expected cost of entering it, in primitive tests */
    };
    std::vector<TopItem>* leaves_out = nullptr; /* build_tree: the leaves it makes, with their boxes */
    /* One unit of the tree build: a single reference, or the references of one small list (a box). */
    struct Item {
        Box box;
        std::vector<int> members; /* indices into the instance's pending references */
    };
    /* Box tree over items[lo, hi); `ref_off` = references laid out before them (instance-local; `ref_base` makes
     * it global).  Binned surface-area heuristic, median split of the widest centroid axis when the bins do not
     * separate anything; an item is never split, so a box's six faces share one leaf and the tree above them
     * separates whole boxes (no overlapping siblings, a third of the depth-first backtracking).  Returns the link of
     * the subtree (see FBvh) and its box. */
    int build_tree(std::vector<Item>& items, int lo, int hi, int ref_off, int ref_base, int depth, int& max_depth, Box& box) {
        box = Box();
        int n_refs = 0;
        for (int k = lo; k < hi; ++k) box.grow(items[k].box), n_refs += (int)items[k].members.size();
        const int n = hi - lo;
        if (n_refs <= kLeafMax || (n == 1 && n_refs <= kGroupMax)) {
            const int link = -1 - (((ref_base + ref_off) << 3) | (n_refs - 1));
            if (leaves_out) leaves_out->push_back(TopItem{box, link, (double)n_refs});
            return link;
        }
        max_depth = std::max(max_depth, depth);
        auto centre = [&](int k, int c) { return 0.5 * (items[k].box.lo[c] + items[k].box.hi[c]); };
        Box cb;
        for (int k = lo; k < hi; ++k) {
            const double ctr[3] = {centre(k, 0), centre(k, 1), centre(k, 2)};
            cb.grow(ctr);
        }
        constexpr int kBins = 16;
        /* bin of a centroid; total for any input (an extent that overflowed to inf gives NaN: bin 0) */
        auto bin_of = [](double ctr, double c0, double ext) {
            const double f = kBins * ((ctr - c0) / ext);
            return f >= 0 ? (f < kBins ? (int)f : kBins - 1) : 0;
        };
        int best_axis = -1, best_bin = -1;
        double best_cost = INFINITY;
        if (depth < kMaxTreeDepth - 8) { /* deep, badly separable sets fall back to balanced median splits */
            for (int axis = 0; axis < 3; ++axis) {
                const double c0 = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
                if (!(ext > 0)) continue;
                Box bb[kBins];
                int cnt[kBins] = {0};
                for (int k = lo; k < hi; ++k) {
                    const int b = bin_of(centre(k, axis), c0, ext);
                    bb[b].grow(items[k].box);
                    cnt[b] += (int)items[k].members.size();
                }
                Box right_box[kBins];
                int right_cnt[kBins];
                Box acc;
                int c = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    if (cnt[b]) acc.grow(bb[b]);
                    c += cnt[b];
                    right_box[b] = acc, right_cnt[b] = c;
                }
                acc = Box(), c = 0;
                for (int b = 0; b + 1 < kBins; ++b) { /* split between bin b and b + 1 */
                    if (cnt[b]) acc.grow(bb[b]);
                    c += cnt[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    const double cost = half_area(acc) * c + half_area(right_box[b + 1]) * right_cnt[b + 1];
                    if (cost < best_cost) best_cost = cost, best_axis = axis, best_bin = b;
                }
            }
        }
        int mid;
        if (best_axis >= 0) {
            const double c0 = cb.lo[best_axis], ext = cb.hi[best_axis] - cb.lo[best_axis];
            auto it = std::partition(items.begin() + lo, items.begin() + hi, [&](const Item& r) {
                const double ctr = 0.5 * (r.box.lo[best_axis] + r.box.hi[best_axis]);
                return bin_of(ctr, c0, ext) <= best_bin;
            });
            mid = (int)(it - items.begin());
        } else {
            int axis = 0;
            for (int c = 1; c < 3; ++c)
                if (cb.hi[c] - cb.lo[c] > cb.hi[axis] - cb.lo[axis]) axis = c;
            mid = (lo + hi) / 2;
            std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi,
                             [axis](const Item& x, const Item& y) {
                                 return x.box.lo[axis] + x.box.hi[axis] < y.box.lo[axis] + y.box.hi[axis];
                             });
        }
        int left_refs = 0;
        for (int k = lo; k < mid; ++k) left_refs += (int)items[k].members.size();
        const int me = (int)out.bvh.size();
        out.bvh.push_back(FBvh{});
        Box bl, br;
        const int l = build_tree(items, lo, mid, ref_off, ref_base, depth + 1, max_depth, bl);
        const int r = build_tree(items, mid, hi, ref_off + left_refs, ref_base, depth + 1, max_depth, br);
        FBvh node{};
        for (int c = 0; c < 3; ++c) {
            node.lmin[c] = float_down(bl.lo[c]), node.lmax[c] = float_up(bl.hi[c]);
            node.rmin[c] = float_down(br.lo[c]), node.rmax[c] = float_up(br.hi[c]);
        }
        node.left = l, node.right = r;
        out.bvh[me] = node;
        return me;
    }

    /* Box tree over items[lo, hi): full-sweep surface-area heuristic over the three centroid orders (the item counts
     * are instance counts, not primitive counts); balanced median splits once the tree is deep.  A single item is
     * linked directly: every leaf of this tree is one instance. */
    int build_top(std::vector<TopItem>& items, int lo, int hi, int depth, int& max_depth, Box& box) {
        box = Box();
        for (int k = lo; k < hi; ++k) box.grow(items[k].box);
        if (hi - lo == 1) return items[lo].link;
        max_depth = std::max(max_depth, depth);
        const int n = hi - lo;
        int best_axis = -1, best_at = -1;
        double best_cost = INFINITY;
        auto by_axis = [](int axis) {
            return [axis](const TopItem& x, const TopItem& y) {
                return x.box.lo[axis] + x.box.hi[axis] < y.box.lo[axis] + y.box.hi[axis];
            };
        };
        if (depth < 24) {
            std::vector<double> right_cost(n);
            for (int axis = 0; axis < 3; ++axis) {
                std::sort(items.begin() + lo, items.begin() + hi, by_axis(axis));
                Box acc;
                double w = 0;
                for (int k = n - 1; k > 0; --k) {
                    acc.grow(items[lo + k].box), w += items[lo + k].weight;
                    right_cost[k] = half_area(acc) * w;
                }
                acc = Box(), w = 0;
                for (int k = 0; k + 1 < n; ++k) { /* split after item k */
                    acc.grow(items[lo + k].box), w += items[lo + k].weight;
                    const double cost = half_area(acc) * w + right_cost[k + 1];
                    if (cost < best_cost) best_cost = cost, best_axis = axis, best_at = k + 1;
                }
            }
        }
        int mid;
        if (best_axis >= 0) {
            std::sort(items.begin() + lo, items.begin() + hi, by_axis(best_axis));
            mid = lo + best_at;
        } else {
            Box cb;
            for (int k = lo; k < hi; ++k) {
                const double ctr[3] = {0.5 * (items[k].box.lo[0] + items[k].box.hi[0]), 0.5 * (items[k].box.lo[1] + items[k].box.hi[1]),
                                       0.5 * (items[k].box.lo[2] + items[k].box.hi[2])};
                cb.grow(ctr);
            }
            int axis = 0;
            for (int c = 1; c < 3; ++c)
                if (cb.hi[c] - cb.lo[c] > cb.hi[axis] - cb.lo[axis]) axis = c;
            mid = (lo + hi) / 2;
            std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi, by_axis(axis));
        }
        const int me = (int)out.bvh.size();
        out.bvh.push_back(FBvh{});
        Box bl, br;
        const int l = build_top(items, lo, mid, depth + 1, max_depth, bl);
        const int r = build_top(items, mid, hi, depth + 1, max_depth, br);
        FBvh node{};
        for (int c = 0; c < 3; ++c) {
            node.lmin[c] = float_down(bl.lo[c]), node.lmax[c] = float_up(bl.hi[c]);
            node.rmin[c] = float_down(br.lo[c]), node.rmax[c] = float_up(br.hi[c]);
        }
        node.left = l, node.right = r;
        out.bvh[me] = node;
        return me;
    }

    int run(int root) { return run(std::vector<int>{root}); }
    /* compile the subtrees under `roots` into one sub-scene; returns its index or -1 (media / too fragmented) */
    int run(const std::vector<int>& roots) {
        /* camera rays carry a time in [time0, time1], shadow rays time 0 (mis_path_integrator.h:210) */
        t_lo = std::min(0.0, std::min(s->camera.time0, s->camera.time1));
        t_hi = std::max(0.0, std::max(s->camera.time0, s->camera.time1));
        const size_t mark[6] = {out.inst.size(), out.xf.size(), out.ref.size(), out.exits.size(), out.bvh.size(), 0};
        inst_base = (int)out.inst.size();
        for (int root : roots) walk(root);
        if (guard_mode && !too_complex) {
            size_t n_refs = 0;
            for (const auto& refs : pending) n_refs += refs.size();
            if (guard_fail || (int)out.inst.size() - inst_base != 1 || n_refs > (size_t)kGuardLinearMax) too_complex = true;
        }
        if (too_complex || (int)out.inst.size() == inst_base) { /* undo */
            out.inst.resize(mark[0]), out.xf.resize(mark[1]), out.ref.resize(mark[2]), out.exits.resize(mark[3]);
            out.bvh.resize(mark[4]);
            return -1;
        }
        int stack = 1;
        std::vector<TopItem> world_leaves; /* of the untransformed instance's tree, if it has one */
        for (size_t ii = inst_base; ii < out.inst.size(); ++ii) {
            FInst& I = out.inst[ii];
            std::vector<PendingRef>& refs = pending[ii - inst_base];
            I.ref_first = (int)out.ref.size();
            I.n_ref = (int)refs.size();
            if ((int)refs.size() > kLinearMax && !guard_mode) {
                int depth = 0;
                Box all;
                std::vector<Item> items;
                std::map<int, int> item_of_group;
                for (int k = 0; k < (int)refs.size(); ++k) {
                    int it = -1;
                    if (refs[k].group >= 0) {
                        auto f = item_of_group.find(refs[k].group);
                        if (f != item_of_group.end()) it = f->second;
                    }
                    if (it < 0) {
                        it = (int)items.size();
                        items.emplace_back();
                        if (refs[k].group >= 0) item_of_group[refs[k].group] = it;
                    }
                    items[it].box.grow(refs[k].local);
                    items[it].members.push_back(k);
                }
                std::vector<TopItem> leaves;
                leaves_out = I.n_xf == 0 ? &leaves : nullptr;
                I.bvh_root = build_tree(items, 0, (int)items.size(), 0, I.ref_first, 1, depth, all);
                leaves_out = nullptr;
                if (I.n_xf == 0) world_leaves.swap(leaves);
                std::vector<PendingRef> ordered; /* references in leaf order; a group keeps the reference's visiting order */
                ordered.reserve(refs.size());
                for (const Item& it : items)
                    for (int k : it.members) ordered.push_back(refs[k]);
                refs.swap(ordered);
                double bound = 0;
                for (int c = 0; c < 3; ++c) bound = std::max(bound, std::max(std::fabs(all.lo[c]), std::fabs(all.hi[c])));
                I.bound = float_up(bound);
                stack = std::max(stack, depth + 2);
            }
            I.flags = 0;
            for (int k = 0; k < I.n_xf; ++k)
                if (out.xf[I.xf_first + k].type == RTR_NODE_ROTATE_Y) I.flags |= RT_INST_ROTATED;
            for (int k = 0; k < RT_INST_XF_INLINE; ++k) {
                I.xf_type[k] = k < I.n_xf ? out.xf[I.xf_first + k].type : 0;
                for (int c = 0; c < 3; ++c) I.xf_f[k][c] = k < I.n_xf ? out.xf[I.xf_first + k].f[c] : 0.0;
            }
            for (const PendingRef& r : refs)
                if (s->nodes[r.node].type < RTR_NODE_XY_RECT) I.flags |= RT_INST_SPHERES;
            Box local;
            for (const PendingRef& r : refs) {
                FRef fr{};
                fr.node = r.node;
                fr.exit_first = (int)out.exits.size();
                fr.n_exit = (int)r.wrappers.size();
                fr.pad = r.order;
                for (auto it = r.wrappers.rbegin(); it != r.wrappers.rend(); ++it) out.exits.push_back(*it);
                if (guard_mode && s->nodes[r.node].type == RTR_NODE_SPHERE && s->nodes[r.node].f[3] < 0) {
                    out.guard_of_ref[(int)out.ref.size()] = {(int)out.guards.size(), (int)r.guards.size()};
                    for (int g : r.guards) {
                        FGuard gb;
                        for (int c = 0; c < 6; ++c) gb.b[c] = s->nodes[g].f[c];
                        out.guards.push_back(gb);
                    }
                }
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
        FSub sub{};
        sub.inst_first = inst_base;
        sub.n_inst = (int)out.inst.size() - inst_base;
        sub.top_root = -1, sub.world_inst = -1;
        {
            std::vector<TopItem> items;
            double bound = 0;
            auto grow_bound = [&bound](const FInst& I) {
                for (int c = 0; c < 3; ++c) bound = std::max(bound, std::max(std::fabs(I.bmin[c]), std::fabs(I.bmax[c])));
            };
            int world = -1;
            for (size_t ii = inst_base; ii < out.inst.size(); ++ii) {
                const FInst& I = out.inst[ii];
                if (I.n_xf == 0) {
                    world = (int)ii; /* (one per sub-scene: instances are keyed by their chain) */
                    continue;
                }
                TopItem it;
                for (int c = 0; c < 3; ++c) it.box.lo[c] = I.bmin[c], it.box.hi[c] = I.bmax[c];
                it.link = -1 - (RT_TOP_INST | (int)ii);
                it.weight = I.bvh_root >= 0 ? 4.0 + 2.0 * std::log2((double)I.n_ref) : 1.0 + I.n_ref;
                items.push_back(it);
                grow_bound(I);
            }
            if ((int)items.size() >= top_tree_min() && out.inst.size() < (size_t)RT_TOP_INST && out.ref.size() < (1u << 26)) {
                sub.world_inst = world;
                if (world >= 0) {
                    const FInst& I = out.inst[world];
                    sub.world_linear = I.bvh_root < 0;
                    if (I.bvh_root >= 0) { /* its frame is this tree's frame: its leaves are items like the instances,
                                              one tree separates both (the instance's own tree stays for the kernels
                                              that scan instances in order) */
                        items.insert(items.end(), world_leaves.begin(), world_leaves.end());
                        grow_bound(I);
                        bound = std::max(bound, (double)I.bound);
                    }
                }
                int depth = 0;
                Box all;
                sub.top_root = build_top(items, 0, (int)items.size(), 1, depth, all);
                sub.top_bound = float_up(bound);
                stack += depth + 2; /* an instance's own tree is walked on top of its stack */
            }
        }
        out.stack_words = std::max(out.stack_words, stack);
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
        default:
            p = 1;
            /* a sphere with a negative radius has an inverted bounding box in the reference
             * (sphere.h:62-66): the bvh_node boxes above it do not enclose it, so whether a ray reaches
             * it is decided by those boxes and the running t_max -- keep the reference's walk there */
            m = (n.type == RTR_NODE_SPHERE && n.f[3] < 0) || (n.type == RTR_NODE_MOVING_SPHERE && n.f[8] < 0);
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
/* `has_media`: the graph holds a constant_medium or an inverted box (order-sensitive parts) */
/* `hollow`: spheres with a negative radius, no constant_medium: first tried as ONE sub-scene with guarded references
 * (rtc::Builder::guard_mode), which keeps the scene on the order-free kernels; the step program below otherwise */
inline CompiledScene compile_scene(const rtr_scene_desc* scene, bool has_media, bool hollow = false) {
    CompiledScene cs;
    cs.dev_nodes.assign(scene->nodes, scene->nodes + scene->n_nodes);
    if (!has_media || hollow) {
        rtc::Builder b(cs);
        b.s = scene;
        b.guard_mode = hollow;
        cs.ok = b.run(scene->root) == 0;
        if (cs.ok || !hollow) {
            if (!cs.ok) {
                cs.subs.clear();
                cs.stack_words = 1;
            }
            return cs;
        }
        cs.subs.clear(), cs.guards.clear(), cs.guard_of_ref.clear();
        cs.stack_words = 1;
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
    /* the step program (struct FStep): the reference's visiting order (bvh_node: left then right,
     * also when both are the same object, bvh.h:56-58; hittable_list: in order) over the part of
     * the graph that holds media; media-free subtrees met on the way are the geometry items */
    {
        std::vector<int> items;
        std::vector<std::vector<int>> item_wrappers; /* per item (a medium): translate / rotate_y / flip_face above it, outermost first */
        std::vector<int> item_chain; /* per item: its innermost bvh_node ancestor as an index into `chain` (-1: none) */
        struct ChainLink {
            int node, up;
        };
        std::vector<ChainLink> chain;
        bool possible = true;
        std::vector<std::pair<int, int>> walk_stack{{scene->root, -1}};
        while (!walk_stack.empty() && possible) {
            const int ix = walk_stack.back().first, up = walk_stack.back().second;
            walk_stack.pop_back();
            const rtr_node& n = scene->nodes[ix];
            const bool hollow = (n.type == RTR_NODE_SPHERE && n.f[3] < 0) || (n.type == RTR_NODE_MOVING_SPHERE && n.f[8] < 0);
            if (!facts.media[ix] || n.type == RTR_NODE_MEDIUM || hollow) {
                items.push_back(ix);
                item_chain.push_back(up);
                item_wrappers.emplace_back();
            } else if (n.type == RTR_NODE_TRANSLATE || n.type == RTR_NODE_ROTATE_Y || n.type == RTR_NODE_FLIP_FACE) {
                /* wrappers straight above ONE medium: a step with a transform chain (FStep::xf_first) */
                std::vector<int> above;
                int cur = ix;
                while (scene->nodes[cur].type == RTR_NODE_TRANSLATE || scene->nodes[cur].type == RTR_NODE_ROTATE_Y ||
                       scene->nodes[cur].type == RTR_NODE_FLIP_FACE) {
                    above.push_back(cur);
                    cur = scene->nodes[cur].a;
                }
                if (scene->nodes[cur].type != RTR_NODE_MEDIUM || above.size() > 30) {
                    possible = false; /* media (or a hollow sphere) deeper under a transform: the reference-order walk */
                } else {
                    items.push_back(cur);
                    item_chain.push_back(up);
                    item_wrappers.push_back(above);
                }
            } else if (n.type == RTR_NODE_LIST) {
                for (int k = n.b - 1; k >= 0; --k) walk_stack.push_back({scene->list_children[n.a + k], up});
            } else if (n.type == RTR_NODE_BVH) {
                chain.push_back({ix, up});
                const int me = (int)chain.size() - 1;
                walk_stack.push_back({n.b, me}), walk_stack.push_back({n.a, me});
            } else {
                possible = false; /* a medium or a hollow sphere under a transform: the reference-order walk handles it */
            }
        }
        const size_t mark[7] = {cs.inst.size(), cs.xf.size(), cs.ref.size(), cs.exits.size(), cs.bvh.size(), cs.subs.size(),
                                (size_t)cs.stack_words};
        std::vector<int> group;
        auto flush = [&]() {
            if (group.empty() || !possible) return;
            rtc::Builder b(cs);
            b.s = scene;
            const int sub = b.run(group);
            group.clear();
            if (sub < 0) {
                possible = false;
                return;
            }
            FStep st{};
            st.kind = 0, st.sub = sub;
            cs.steps.push_back(st);
        };
        const size_t guard_mark = cs.guards.size();
        for (size_t k = 0; k < items.size() && possible; ++k) {
            const rtr_node& n = scene->nodes[items[k]];
            if (n.type == RTR_NODE_SPHERE || n.type == RTR_NODE_MOVING_SPHERE) {
                const bool hollow = (n.type == RTR_NODE_SPHERE && n.f[3] < 0) || (n.type == RTR_NODE_MOVING_SPHERE && n.f[8] < 0);
                if (hollow) { /* a guarded step of its own (FStep kind 3) */
                    flush();
                    if (!possible) break;
                    rtc::Builder b(cs);
                    b.s = scene;
                    const int sub = b.run(items[k]);
                    if (sub < 0) {
                        possible = false;
                        break;
                    }
                    std::vector<int> above; /* innermost first */
                    for (int c = item_chain[k]; c >= 0; c = chain[c].up) above.push_back(chain[c].node);
                    FStep st{};
                    st.kind = 3, st.sub = sub, st.pad = (int)cs.guards.size(), st.mat = (int)above.size();
                    for (auto it = above.rbegin(); it != above.rend(); ++it) {
                        FGuard g;
                        for (int c = 0; c < 6; ++c) g.b[c] = scene->nodes[*it].f[c];
                        cs.guards.push_back(g);
                    }
                    cs.steps.push_back(st);
                    continue;
                }
            }
            if (n.type != RTR_NODE_MEDIUM) {
                group.push_back(items[k]);
                continue;
            }
            flush();
            if (!possible) break;
            rtc::Builder b(cs);
            b.s = scene;
            for (int w : item_wrappers[k]) { /* the boundary's references sit under the medium's wrappers */
                b.wrappers.push_back(w);
                if (scene->nodes[w].type != RTR_NODE_FLIP_FACE) b.chain.push_back(w);
            }
            const int sub = b.run(n.a);
            if (sub < 0) {
                possible = false;
                break;
            }
            FStep st{};
            st.kind = 1, st.sub = sub, st.mat = n.b, st.neg_inv_density = n.f[0];
            st.xf_first = (int)cs.xf.size();
            for (int w : item_wrappers[k]) {
                const rtr_node& wn = scene->nodes[w];
                if (wn.type == RTR_NODE_FLIP_FACE) continue;
                FXf x{};
                x.type = wn.type;
                x.f[0] = wn.f[0], x.f[1] = wn.f[1], x.f[2] = wn.f[2];
                cs.xf.push_back(x);
            }
            st.n_xf = (int)cs.xf.size() - st.xf_first;
            st.exit_first = (int)cs.exits.size();
            for (auto it = item_wrappers[k].rbegin(); it != item_wrappers[k].rend(); ++it) cs.exits.push_back(*it);
            st.n_exit = (int)item_wrappers[k].size();
            const rtr_node& bn = scene->nodes[n.a];
            const FSub& bs = cs.subs[sub];
            if (bn.type == RTR_NODE_SPHERE && bn.f[3] > 0 && bs.n_inst == 1 && cs.inst[bs.inst_first].n_xf == 0 &&
                cs.inst[bs.inst_first].n_ref == 1 && cs.inst[bs.inst_first].bvh_root < 0)
                st.kind = 2, st.pad = cs.inst[bs.inst_first].ref_first;
            cs.steps.push_back(st);
            cs.step_tail = (int)cs.steps.size();
        }
        flush();
        if (!possible) { /* undo */
            cs.inst.resize(mark[0]), cs.xf.resize(mark[1]), cs.ref.resize(mark[2]), cs.exits.resize(mark[3]);
            cs.bvh.resize(mark[4]), cs.subs.resize(mark[5]);
            cs.stack_words = (int)mark[6];
            cs.steps.clear();
            cs.step_tail = 0;
            cs.guards.resize(guard_mark);
        }
    }
    return cs;
}

inline void rtc::build_scan_runs(CompiledScene& cs, const std::vector<rtr_node>& prims) {
    cs.scan.clear();
    for (FInst& I : cs.inst) {
        I.flags &= ~RT_INST_RUNS;
        I.scan_first = 0;
        I.runs = 0;
        if (I.bvh_root >= 0 || I.n_ref == 0) continue;
        std::vector<std::pair<int, int>> runs; /* type, count */
        std::vector<double> data;
        bool ok = true;
        const int kBox = RTR_NODE_SPHERE + RT_RUN_BOX;
        auto same = [](double a, double b) { return std::memcmp(&a, &b, 8) == 0; };
        /* the six references from r on are the sides of one box, in box.h's order and with its extents */
        auto box_at = [&](int r) {
            if (r + 6 > I.ref_first + I.n_ref) return false;
            const rtr_node* p = &prims[r];
            static const int want_type[6] = {RTR_NODE_XY_RECT, RTR_NODE_XY_RECT, RTR_NODE_XZ_RECT,
                                             RTR_NODE_XZ_RECT, RTR_NODE_YZ_RECT, RTR_NODE_YZ_RECT};
            for (int k = 0; k < 6; ++k)
                if (p[k].type != want_type[k] || (p[k].reserved & RT_TIE_FLAG)) return false;
            const double x0 = p[0].f[0], x1 = p[0].f[1], y0 = p[0].f[2], y1 = p[0].f[3], z1 = p[0].f[4], z0 = p[1].f[4];
            const double want[6][5] = {{x0, x1, y0, y1, z1}, {x0, x1, y0, y1, z0}, {x0, x1, z0, z1, y1},
                                       {x0, x1, z0, z1, y0}, {y0, y1, z0, z1, x1}, {y0, y1, z0, z1, x0}};
            for (int k = 0; k < 6; ++k)
                for (int c = 0; c < 5; ++c)
                    if (!same(p[k].f[c], want[k][c])) return false;
            return true;
        };
        for (int r = I.ref_first; r < I.ref_first + I.n_ref && ok;) {
            const rtr_node& n = prims[r];
            if (n.reserved & RT_TIE_FLAG) ok = false;
            const bool box = box_at(r);
            /* (a guarded run is read by the kernels of guarded scenes only -- RT_TRAV_FLAT_GUARD --, the others that meet
             * such an instance scan it through the generic loop) */
            const bool guarded = (n.reserved & RT_GUARD_FLAG) != 0;
            const int type = box ? kBox : (guarded ? RTR_NODE_SPHERE + RT_RUN_GUARDED : n.type);
            if (runs.empty() || runs.back().first != type || runs.back().second == RT_RUN_COUNT_MAX) runs.push_back({type, 0});
            ++runs.back().second;
            if (guarded) { /* centre, radius, first guard and guard count (as the integers' bits): six words */
                data.insert(data.end(), n.f, n.f + 6);
                r += 1;
            } else if (box) {
                const double rec[6] = {n.f[0], n.f[1], n.f[2], n.f[3], prims[r + 1].f[4], n.f[4]}; /* x0 x1 y0 y1 z0 z1 */
                data.insert(data.end(), rec, rec + 6);
                r += 6;
            } else {
                const int nf = n.type == RTR_NODE_SPHERE ? 4 : (n.type == RTR_NODE_MOVING_SPHERE ? 9 : 5);
                data.insert(data.end(), n.f, n.f + nf);
                r += 1;
            }
        }
        if (!ok || (int)runs.size() > RT_INST_RUNS_MAX) continue;
        I.scan_first = (int32_t)cs.scan.size();
        for (size_t k = 0; k < runs.size(); ++k)
            I.runs |= (uint64_t)((runs[k].first - RTR_NODE_SPHERE) << RT_RUN_COUNT_BITS | runs[k].second) << (RT_RUN_BITS * k);
        cs.scan.insert(cs.scan.end(), data.begin(), data.end());
        for (size_t k = 0; k < 6; ++k) I.head[k] = k < data.size() ? data[k] : 0.0;
        I.flags |= RT_INST_RUNS;
    }
    cs.scan.resize(cs.scan.size() + 16, 0.0); /* the two-records-per-trip loads never leave the array */
}

inline std::vector<FLeaf> rtc::build_leaf_records(const CompiledScene& cs, const std::vector<rtr_node>& prims) {
    std::vector<FLeaf> out(prims.size());
    for (size_t r = 0; r < prims.size(); ++r) {
        FLeaf L{};
        const rtr_node& n = prims[r];
        const int nf = n.type == RTR_NODE_SPHERE ? 4 : (n.type == RTR_NODE_MOVING_SPHERE ? 0 : 5);
        for (int k = 0; k < nf; ++k) L.f[k] = n.f[k];
        L.type = n.type, L.tag = n.reserved;
        out[r] = L;
    }
    return out;
}
