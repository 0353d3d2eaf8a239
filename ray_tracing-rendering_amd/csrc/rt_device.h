/*
 * rt_device.h -- gfx950 device functions of the path tracer: vector math, xorshift32,
 * hittable-graph traversal with an LDS stack, textures, materials, lights, camera and the
 * two integrator bounce loops.  Shared by the megakernel and the wavefront stage kernels
 * (rt_kernels.hip).
 *
 * Numerics contract ("exact" build, -ffp-contract=off): every expression keeps the
 * reference's operation order in IEEE binary64 (division and sqrt correctly rounded), so
 * results are bit-identical to the reference wherever no libm transcendental is involved
 * (all of the Cornell scenes); sin/cos/log/acos/atan2 come from OCML and may differ from
 * glibc in the last ulp (scenes 23, 9, 22); pow(x, 5) is a correctly rounded x^5 (pow5).  RNG draws are sequenced in the order
 * g++ evaluates the reference's argument lists.
 *
 * Citations are reference paths relative to /root/reference/src.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtr_hip.h"
#include "rtr_seed.h"

#ifndef RTR_BLOCK
#define RTR_BLOCK 256 /* threads per workgroup = one 16x16 tile (renderer/renderer.h:40) */
#endif

#define RT_DEV __device__ __forceinline__

typedef double Real;

/* ---- region profile (debug builds: -DRTR_REGION_PROFILE; tools/region_profile.sh) ---------------------------------
 * Where a wave's cycles go: every marker charges the shader cycles (s_memtime) since the wave's previous marker to
 * the region that was current, then makes `id` current.  Counters live in LDS per wave and are added to
 * RenderK::stats[RT_PROF_BASE ...] when the workgroup ends.  The markers cost about a tenth of the wave's time
 * themselves: read the table as proportions.  In product builds RT_REGION() expands to nothing. */
#define RT_STATS_WORDS 80 /* u64 words of RenderK::stats: 8 of the product + 2 x 32 region words */
#define RT_PROF_BASE 8
#define RT_PROF_REGIONS 32
enum { RG_OTHER = 0, RG_SETUP = 1, RG_RECTS = 2, RG_SPHERES = 3, RG_GENERIC = 4, RG_TREE = 5, RG_LEAVES = 6, RG_FINISH = 7,
       RG_SH_SETUP = 8, RG_SH_RECTS = 9, RG_SH_SPHERES = 10, RG_SH_GENERIC = 11, RG_SH_TREE = 12, RG_SH_LEAVES = 13,
       RG_MEDIA = 14, RG_MATPREP = 15, RG_SHADE_A = 16, RG_SHADE_B = 17, RG_MISS = 18, RG_REGEN = 19, RG_SHADE_RR = 20,
       RG_PARK = 21, RG_BARRIER = 22, RG_EXCHANGE = 23, RG_N = 24 };
#ifdef RTR_REGION_PROFILE
__shared__ unsigned long long rt_prof_lds[4 * 2 * RT_PROF_REGIONS + 4 * 2]; /* [wave][cycles | visits][region], then [wave][last, current] */
RT_DEV void rt_region(int id) {
    const unsigned long long m = __ballot(1);
    if ((int)__lane_id() == __builtin_ctzll(m)) {
        const int w = threadIdx.x >> 6;
        unsigned long long* acc = rt_prof_lds + w * 2 * RT_PROF_REGIONS;
        unsigned long long* st = rt_prof_lds + 4 * 2 * RT_PROF_REGIONS + w * 2;
        const unsigned long long now = __builtin_readcyclecounter();
        acc[st[1]] += now - st[0];
        acc[RT_PROF_REGIONS + id] += 1;
        st[0] = now, st[1] = (unsigned long long)id;
    }
}
#define RT_REGION(id) rt_region(id)
#else
#define RT_REGION(id) do { } while (0)
#endif

#define RT_INF (__builtin_huge_val())
#define RT_PI 3.1415926535897932385 /* core/rtweekend.h:18 */

/* ---- core/vec3.h:12-224 --------------------------------------------------------------- */
struct V3 {
    Real x, y, z;
};
RT_DEV V3 mk(Real x, Real y, Real z) {
    V3 r;
    r.x = x, r.y = y, r.z = z;
    return r;
}
RT_DEV V3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }
RT_DEV V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
RT_DEV V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 mul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 scl(Real t, V3 v) { return mk(t * v.x, t * v.y, t * v.z); }
RT_DEV V3 divs(V3 v, Real t) { return scl(1 / t, v); } /* vec3.h:208-210: multiply by 1/t */
RT_DEV Real dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV V3 cross(V3 u, V3 v) { return mk(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }
RT_DEV Real len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
RT_DEV Real len(V3 a) { return __builtin_sqrt(len2(a)); }
RT_DEV V3 unit(V3 v) { return divs(v, len(v)); }
RT_DEV bool near_zero(V3 a) { /* vec3.h:81-85 */
    const Real s = 1e-8;
    return (__builtin_fabs(a.x) < s) && (__builtin_fabs(a.y) < s) && (__builtin_fabs(a.z) < s);
}
RT_DEV V3 reflect(V3 v, V3 n) { return sub(v, scl(2 * dot(v, n), n)); } /* vec3.h:239-241 */
RT_DEV V3 refract(V3 uv, V3 n, Real etai_over_etat) {                    /* vec3.h:243-248 */
    Real cos_theta = __builtin_fmin(dot(neg(uv), n), 1.0);
    V3 r_out_perp = scl(etai_over_etat, add(uv, scl(cos_theta, n)));
    V3 r_out_parallel = scl(-__builtin_sqrt(__builtin_fabs(1.0 - len2(r_out_perp))), n);
    return add(r_out_perp, r_out_parallel);
}
RT_DEV Real clampd(Real x, Real lo, Real hi) { /* rtweekend.h:40-46 */
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
RT_DEV Real max3(V3 a) { /* std::max({x,y,z}) */
    Real m = a.x;
    if (m < a.y) m = a.y;
    if (m < a.z) m = a.z;
    return m;
}
RT_DEV Real maxd(Real a, Real b) { return a < b ? b : a; } /* std::max(a,b) */

/* ---- core/rtweekend.h:24-50 ------------------------------------------------------------ */
RT_DEV Real rng_next(uint32_t& s) {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s * 2.3283064365386963e-10;
}
RT_DEV Real rng_range(uint32_t& s, Real lo, Real hi) { return lo + (hi - lo) * rng_next(s); }
RT_DEV int rng_int(uint32_t& s, int lo, int hi) { return (int)rng_range(s, lo, hi + 1); }

/* random_double(-1, 1) = -1 + (1 - -1) * (s * 2^-32) (rtweekend.h:36-38).  Doubling is exact, so
 * this equals -1 + s * 2^-31 bit for bit, with one multiply less. */
RT_DEV Real rng_sym(uint32_t& s) {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return -1.0 + s * 4.656612873077392578125e-10;
}

/* vec3::random(-1,1) accepted into the unit ball (vec3.h:226-233); z takes the first draw */
RT_DEV V3 random_in_unit_sphere(uint32_t& s) {
    for (;;) {
        Real z = rng_sym(s);
        Real y = rng_sym(s);
        Real x = rng_sym(s);
        V3 p = mk(x, y, z);
        if (len2(p) >= 1) continue;
        return p;
    }
}
RT_DEV V3 random_unit_vector(uint32_t& s) { return unit(random_in_unit_sphere(s)); }
RT_DEV V3 random_in_unit_disk(uint32_t& s) { /* vec3.h:250-257; y first */
    for (;;) {
        Real y = rng_sym(s);
        Real x = rng_sym(s);
        V3 p = mk(x, y, 0);
        if (len2(p) >= 1) continue;
        return p;
    }
}
RT_DEV V3 random_cosine_direction(uint32_t& s) { /* vec3.h:261-269 */
    Real r1 = rng_next(s);
    Real r2 = rng_next(s);
    Real z = __builtin_sqrt(1 - r2);
    Real phi = 2 * RT_PI * r1;
    /* cos(phi) and sin(phi) from ONE sincos(): the same bits as the two calls for every phi = 2 pi s 2^-32 the
     * generator can produce (rtr_test_sincos_exhaustive walks all 2^32 of them), one argument reduction less */
    Real sphi, cphi;
    sincos(phi, &sphi, &cphi);
    Real x = cphi * __builtin_sqrt(r2);
    Real y = sphi * __builtin_sqrt(r2);
    return mk(x, y, z);
}

/* ---- IEEE division with a shared divisor ------------------------------------------------------
 * hipcc expands a double division n / d into eleven VALU instructions:
 *     ds = v_div_scale(d), ns = v_div_scale(n), r0 = v_rcp_f64(ds),
 *     r1 = fma(r0, fma(-ds, r0, 1), r0), r2 = fma(r1, fma(-ds, r1, 1), r1),        <- depends on d only
 *     q0 = ns * r2, q1 = v_div_fmas(fma(-ds, q0, ns), r2, q0), v_div_fixup(q1, d, n)
 * v_div_scale returns its operand unchanged, v_div_fmas is a plain fma and v_div_fixup returns q1 unless an
 * operand is zero / inf / NaN / denormal or the exponents are extreme (difference >= 768, |n| < 2^-969,
 * |d| > 2^1021, |n / d| < 2^-1022).  A ray divides many numerators by the same few numbers -- the three
 * components of its direction (x?_rect::hit: t = (k - o) / d, aarect.h:80,99,118) and |d|^2 (sphere::hit:
 * root = (-half_b -+ sqrtd) / a, sphere.h:43-47) -- so r2 is computed once per ray and frame and a division costs
 * the last three instructions.  While |d| is in [2^-100, 2^100], |n| in [2^-300, 2^200] that IS the compiler's
 * sequence, operand for operand: the same bits as n / d (tests/: rtr_test_shared_division on 2^32 pairs).
 * Outside: an unsafe divisor (or a ray origin beyond 2^80, which bounds n) switches the whole wave to plain
 * divisions (`fast`, wave-uniform); a numerator below 2^-300 (zero included: the sign of a zero quotient) gives a
 * quotient below 2^-200 either way, which every caller with t_min >= 2^-100 rejects by `t < t_min` -- the
 * others (medium boundaries: t_min = -inf) take the plain division for such numerators (`guard`). */
RT_DEV Real rcp_refined(Real d) {
    Real r = __builtin_amdgcn_rcp(d);
    Real e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}
RT_DEV bool rcp_safe(Real d) { return (__builtin_fabs(d) >= 0x1p-100) & (__builtin_fabs(d) <= 0x1p100); }
/* n / d given r = rcp_refined(d).  SHARED is decided once per ray and frame (RayDiv::fast, wave-uniform), outside the
 * loops over primitives: every instruction counts in them, a scalar branch as much as an FP64 one */
template <bool SHARED>
RT_DEV Real div_shared(Real n, Real d, Real r, bool guard) {
    if (!SHARED) return n / d;
    const Real q = n * r;
    const Real e = __builtin_fma(-d, q, n);
    Real t = __builtin_fma(e, r, q);
    if (guard && !(__builtin_fabs(n) >= 0x1p-300)) t = n / d;
    return t;
}
/* what the primitive tests of one ray in one frame share */
struct RayDiv {
    Real rx, ry, rz; /* rcp_refined of the direction's components */
    Real a, ra;      /* |d|^2 and its refined reciprocal (frames that hold spheres) */
    bool fast;       /* wave-uniform: every lane's direction components (and a, where the frame holds spheres), origin and
                        interval allow div_shared's short form */
    bool guard;      /* t_min < 2^-100: tiny numerators take the plain division */
};
RT_DEV RayDiv raydiv_none() {
    RayDiv q;
    q.rx = q.ry = q.rz = q.a = q.ra = 0;
    q.fast = q.guard = false;
    return q;
}
/* every lane's origin is small enough for div_shared's numerators (checked once per cast, in the world frame: the
 * transforms of a chain add scene coordinates below 2^60 to it at most 30 times) */
RT_DEV bool raydiv_origin_ok(V3 o) {
    return __all((__builtin_fabs(o.x) <= 0x1p80) & (__builtin_fabs(o.y) <= 0x1p80) & (__builtin_fabs(o.z) <= 0x1p80));
}
/* one frame of a cast; `origin_ok` = raydiv_origin_ok of the world ray and DScene::shared_div */
RT_DEV RayDiv raydiv_make(V3 d, Real tmin, bool origin_ok) {
    RayDiv q;
    q.rx = rcp_refined(d.x), q.ry = rcp_refined(d.y), q.rz = rcp_refined(d.z);
    q.a = q.ra = 0;
    q.guard = !(tmin >= 0x1p-100);
    q.fast = origin_ok && __all(rcp_safe(d.x) & rcp_safe(d.y) & rcp_safe(d.z));
    return q;
}
RT_DEV void raydiv_spheres(RayDiv& q, V3 d) {
    q.a = len2(d);
    q.ra = rcp_refined(q.a);
    q.fast = q.fast && __all(rcp_safe(q.a));
}
/* 1 / d for the conservative box tests: the refined reciprocal is within an ulp of it */
RT_DEV V3 raydiv_inv(const RayDiv& q, V3 d) {
    if (q.fast) return mk(q.rx, q.ry, q.rz);
    return mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
}

/* ---- device scene ------------------------------------------------------------------------ */
/* Scene arrays are immutable during a render.  Reading them through the constant address space
 * tells the compiler so: wave-uniform accesses (instance / reference / primitive loops of the
 * compiled scene) become scalar loads into SGPRs instead of 64 identical vector loads. */
#define RT_CONST_AS __attribute__((address_space(4)))
template <class T>
RT_DEV const RT_CONST_AS T* as_const(const T* p) {
    return (const RT_CONST_AS T*)(unsigned long long)p;
}
/* copy record `idx` of a POD array out of the constant address space (unused fields fold away) */
template <class T>
RT_DEV T ld_const(const T* base, int idx) {
    static_assert(sizeof(T) % 8 == 0, "records are multiples of 8 bytes");
    const RT_CONST_AS unsigned long long* src = as_const(reinterpret_cast<const unsigned long long*>(base + idx));
    unsigned long long w[sizeof(T) / 8];
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 8; ++k) w[k] = src[k];
    T v;
    __builtin_memcpy(&v, w, sizeof(T));
    return v;
}

struct DScene {
    const rtr_node* nodes;
    const int32_t* list_children;
    const rtr_material* materials;
    const rtr_texture* textures;
    const rtr_perlin* perlin;
    const rtr_image* images;
    const uint8_t* image_bytes;
    const rtr_light* lights;
    rtr_camera camera;
    double background[3];
    int32_t root;
    int32_t n_nodes;
    int32_t n_lights;
    int32_t needs_uv; /* some texture reads (u,v): image textures */
    int32_t shared_div; /* every coordinate of the scene is below 2^60 and no transform chain is deeper than 30: the primitive
                           tests may divide through RayDiv (div_shared's range argument) */
    int32_t pad_;
    /* compiled scene for the order-free traversal (trace_fast); scenes with media use it per step (FStep) */
    const struct FInst* finst;
    const struct FXf* fxf;
    const struct FRef* fref;
    const rtr_node* fprim; /* copy of the primitive's node record per reference (one load hop less) */
    const double* fscan;   /* packed geometry of linearly scanned references (see FInst) */
    const struct FMat* fmat; /* per material: its record with the values of its solid_color textures inline (see FMat) */
    const struct FLeaf* fleaf; /* leaf records of the box trees, one per reference (see FLeaf) */
    const int32_t* fexit;
    const struct FBvh* fbvh;
    const struct FSub* fsub; /* compiled sub-scenes: [0] = whole scene when it has no media */
    int32_t n_finst;         /* instances of sub-scene 0 (0 = none) */
    int32_t top_root0, world_inst0, world_linear0; /* sub-scene 0's top tree (struct FSub) */
    float top_bound0;
    int32_t n_fstep;         /* steps of the ray-cast program of a scene with media (0 = none) */
    const struct FStep* fstep;
    int32_t fstep_tail;      /* first step after the last medium */
    const struct FGuard* fguard; /* boxes of the guarded steps (FStep kind 3) */
    int32_t n_fvisit;        /* visits of the traversal machine's flattened program (rt_machine.h) */
    const struct FVisit* fvisit;
};

/*
 * Compiled scene.  Where no constant_medium exists, a ray cast's result does not depend on
 * the order objects are visited in (only exact ties in t could tell, and the reference resolves
 * those by its arbitrary BVH order), so upload lowers the hittable graph to:
 *   instance  = all primitives under the same chain of translate / rotate_y wrappers, with the
 *               world-space box of the set; the ray is moved into the instance's frame once;
 *   reference = one primitive of an instance + the wrappers (translate, rotate_y, flip_face) it
 *               sits under, innermost first, whose hit() epilogues are replayed on the hit record;
 *   BVH       = SAH box tree (struct FBvh) over an instance's references when there are many.
 * Every primitive test and every epilogue uses the same arithmetic as the reference's hit()
 * functions, so the hit record is bit-identical to the reference-order traversal.
 */
struct FInst {
    double bmin[3], bmax[3]; /* world box of the instance (used when n_xf > 0) */
    int32_t xf_first, n_xf;  /* transform ops, outermost first */
    int32_t ref_first, n_ref;
    int32_t bvh_root;        /* -1: scan the references linearly */
    float bound;             /* box tree: largest |coordinate| of any of its boxes (instance frame) */
    int32_t flags;           /* RT_INST_* */
    /* Packed scan records of a linearly scanned instance (bvh_root < 0): its references in visiting order, cut into
     * runs of one primitive type, the geometry of each reference as bare doubles in DScene::fscan from `scan_first` on
     * -- x?_rect: a0 a1 b0 b1 k (aarect.h:31,53,75), sphere: centre radius, moving_sphere: c0 c1 t0 t1 radius.  A
     * run's loop knows its type at compile time and fetches two records per trip through one scalar-load round trip.
     * `runs` holds up to eight runs of eight bits each, first run lowest: (type - RTR_NODE_SPHERE) << 5 | count (a
     * longer run is cut), 0 ends the list (one scalar register pair: shifting it out needs no indexed access to this
     * record).  Instances with more runs than fit, or with tie-capable references, keep the generic loop over fprim
     * (RT_INST_RUNS clear).  Type codes beyond the node types: RT_RUN_BOX, and RT_RUN_GUARDED in scenes with guarded
     * references (read by their own kernel variant, RT_TRAV_FLAT_GUARD). */
    int32_t scan_first;
    int32_t pad;
    uint64_t runs;
    /* the first two transform ops of the chain, inline (a copy of fxf[xf_first ...]): translate(rotate_y(...)) of the
     * reference's scenes comes in with the record itself instead of through two more dependent loads */
    int32_t xf_type[2];
    double xf_f[2][3];
    /* the first six words of the instance's packed scan data (a copy of fscan[scan_first ...]): an instance that is ONE
     * `box` -- translate(rotate_y(box)), the reference's favourite object -- is then complete in this record, and the
     * per-lane walk of a top tree (trace_top) tests it after one fetch instead of two dependent ones */
    double head[6];
};
#define RT_INST_RUNS_MAX 8
#define RT_RUN_BITS 8       /* per run: count in the low RT_RUN_COUNT_BITS bits, type code above */
#define RT_RUN_COUNT_BITS 5
#define RT_RUN_COUNT_MAX 31
#define RT_INST_XF_INLINE 2
#define RT_INST_ROTATED 1 /* a rotate_y in the chain: the direction's x and z differ from the frame above */
#define RT_INST_SPHERES 2 /* holds sphere / moving_sphere references: the frame needs |d|^2 and its reciprocal */
#define RT_INST_RUNS 4    /* scan_first / run[] describe the references (see FInst) */
struct FXf {
    int32_t type; /* RTR_NODE_TRANSLATE (f = offset) or RTR_NODE_ROTATE_Y (f[0] = sin, f[1] = cos) */
    int32_t pad;
    double f[3];
};
/* fprim[ref].f[9] carries, as an integer's bits, the wrappers the reference sits under, innermost first, two bits
 * each: 1 = the next translate / rotate_y of its instance's chain (from the inside), 2 = flip_face, 0 = end;
 * RT_EXIT_LONG in place of it all = more than 31 wrappers or a moving_sphere-sized record: use the FRef list. */
#define RT_EXIT_LONG (~0ull)
struct FRef {
    int32_t node;       /* primitive node index */
    int32_t exit_first; /* into fexit: wrapper node indices, innermost first */
    int32_t n_exit;
    int32_t pad;        /* host side: visiting order (copied into fprim[].reserved) */
};
/* One INNER node of an instance's box tree: the boxes of both children next to their links, so a
 * traversal step costs one 64-byte record.  A link >= 0 is another inner node; a negative link is a
 * leaf, -1 - ((first_reference << 3) | (count - 1)), count <= 8.  The tree is built with a binned
 * surface-area heuristic (rt_compile.h).  Boxes are SINGLE precision, rounded outward: they only
 * prune, the primitive tests behind them stay in double and exact (see struct BoxRay). */
struct FBvh {
    float lmin[3], lmax[3], rmin[3], rmax[3];
    int32_t left, right;
    int32_t pad[2];
};
#define RT_BVH_DONE (-2147483647 - 1) /* traversal sentinel: nothing left on the stack */
/* One record per material for mat_prepare: the rtr_material plus, where every texture it reads is a solid_color
 * (texture.h:46-48: value() returns the colour whatever (u, v, p) are), those values -- one fetch per hit instead of the
 * material and then up to three texture records behind it.  `solid` = 0: mat_prepare takes the textures' own path. */
struct FMat {
    int32_t type, solid;
    int32_t tex[4];
    int32_t pad[2];
    double f[4];
    double albedo[3];    /* value of tex[0] */
    double rough, metal; /* PBR: tex[1] clamped to [0.01, 1] (material.h:264,327,368), tex[2] */
};
/* What a lane fetches per reference of a tree leaf: 64 bytes instead of the 96-byte node record.  (A leaf holding the six
 * sides of a `box` as ONE record -- one fetch, three shared reciprocals, the six rectangle tests in list order -- was built
 * and measured: half the leaf instructions, bit-identical, and 3 % SLOWER on scenes 9 / 22 (1 459 vs 1 510 Msamples/s): the
 * tree kernels are bound by the latency of dependent fetches at three waves per SIMD and by their spills, not by
 * instruction count; profiles/README.md.) */
struct FLeaf {
    double f[6];  /* x?_rect: a0 a1 b0 b1 k; sphere: centre radius */
    int32_t type; /* RTR_NODE_*; a moving_sphere is tested from its fprim record */
    int32_t tag;  /* fprim[].reserved: tie flag | visiting position */
    int32_t pad[2];
};
/* A compiled sub-scene = a range of instances.  Scenes WITH media keep the reference-order walk
 * for the (small) part of the graph the media live in, and every large media-free subtree under
 * it is compiled on its own: the walk meets it as one node of type RT_NODE_COMPILED at the place
 * the reference would enter that subtree, with the same t_max, so RNG draws and results of the
 * media around it are unchanged while the bulk of the geometry is traversed order-free. */
struct FSub {
    int32_t inst_first, n_inst;
    /* Sub-scenes with many instances (rt_compile.h: kTopTreeMin): one box tree in the sub-scene's own ("world") frame
     * over the world boxes of the transformed instances, walked PER LANE (trace_top) instead of the wave-uniform scan
     * of every instance.  Its leaves are single instances (link code RT_TOP_INST | index) and, where the untransformed
     * instance `world_inst` (whose frame is the world frame) has a box tree, the leaves of that tree: one tree
     * separates both kinds, a cast is one walk.
     * top_root < 0: no such tree, the instances are scanned in order. */
    int32_t top_root;
    int32_t world_inst;   /* the instance without a transform chain, or -1 */
    int32_t world_linear; /* 1: `world_inst` has no box tree: scanned (wave-uniformly) before the walk */
    float top_bound;      /* largest |coordinate| of any box of the walk (see FInst::bound) */
    int32_t pad[2];
};
#define RT_TOP_INST (1 << 30) /* leaf code of a top tree: RT_TOP_INST | instance index */
#define RT_NODE_COMPILED 11 /* device-only node type: a = sub-scene index */
/* Ray-cast program of a scene whose constant_media sit under bvh_nodes / hittable_lists only: the
 * reference's visiting order cut into steps -- a medium is a step of its own, every run of
 * media-free objects between two media is merged into ONE compiled sub-scene.  Both containers hand
 * each member the closest t so far and keep the closest record (hittable_list.h:33-47,
 * bvh.h:40-50), so merging neighbours changes nothing, and a medium sees the t_max the reference
 * passes to it: its random draw (constant_medium.h:88) happens under the same condition, in the
 * same sequence.  What the program drops are the box tests of the bvh_nodes above a medium; a box
 * the ray misses within [t_min, t_max] contains no boundary segment there either, so the medium
 * returns before drawing -- the two can differ only when a boundary hit and a box face coincide to
 * the last bit (tests/: program == reference-order walk on every golden scene).  All lanes of a
 * wave run the same steps; there is no per-node interpretation left on this path. */
struct FStep {
    int32_t kind; /* 0: geometry sub-scene `sub`; 1: constant_medium with boundary sub-scene `sub`; 2: constant_medium whose
                   * boundary is one plain sphere in the world frame, reference `pad` (run_program's short cut; `sub` as for 1);
                   * 3: a GUARDED primitive (sub-scene `sub` holds it alone): a sphere with a negative radius -- hollow
                   * glass, scenes.cpp:903 -- whose bounding box is inverted (sphere.h:62-66), so the boxes of the
                   * bvh_nodes above it do not enclose it and the reference reaches it only through rays that pass
                   * every one of those boxes within [t_min, closest t so far] (bvh.h:40-50, aabb.h:31-48): the step
                   * runs exactly those box tests, `mat` of them from DScene::fguard[`pad`] on, root first */
    int32_t sub;
    int32_t mat;  /* medium: phase function material */
    int32_t pad;
    double neg_inv_density;
    /* a constant_medium UNDER translate / rotate_y / flip_face wrappers (scenes.cpp:214-217 wraps the boundary; a
     * wrapped medium is what tests/_randscene.py's moved_media builds): the transform ops of that chain, outermost
     * first, in DScene::fxf -- constant_medium::hit measures the ray's length in ITS frame (constant_medium.h:84) --
     * and all wrappers, innermost first, in DScene::fexit for the epilogues on its hit record (hittable.h:58-61,
     * 142-155, 168).  The boundary sub-scene `sub` carries the same chain in its instances, so it is cast with the
     * outer ray like every sub-scene. */
    int32_t xf_first, n_xf;
    int32_t exit_first, n_exit;
};
struct FGuard {
    double b[6]; /* a bvh_node's box: min, max (geometry/bvh.h:12-31) */
};

struct Hit { /* geometry/hittable.h:10-23 */
    V3 p, n;
    Real t, u, v;
    int mat;
    bool front;
};

RT_DEV void set_face_normal(Hit& rec, V3 rd, V3 outward) { /* hittable.h:19-22 */
    rec.front = dot(rd, outward) < 0;
    rec.n = rec.front ? outward : neg(outward);
}

/* ---- per-lane traversal stack in LDS, interleaved so lane l owns bank l%32 ------------------ */
struct Stack {
    int* col; /* &lds[threadIdx.x]; entry k lives at col[k*RTR_BLOCK] */
    RT_DEV void put(int k, int v) const { col[k * RTR_BLOCK] = v; }
    RT_DEV int get(int k) const { return col[k * RTR_BLOCK]; }
    RT_DEV void putd(int k, Real v) const {
        put(k, __double2loint(v));
        put(k + 1, __double2hiint(v));
    }
    RT_DEV Real getd(int k) const { return __hiloint2double(get(k + 1), get(k)); }
};

/* stack words used by the wrapper frames (must match the analysis in rtr_capi.hip) */
#define RT_FRAME_TRANSLATE 8 /* o.xyz (6) + hits + marker */
#define RT_FRAME_ROTATE 14   /* o.x o.z d.x d.z inv.x inv.z (12) + hits + marker */
#define RT_FRAME_FLIP 2      /* hits + marker */
/* a hittable_list with more than RT_LIST_BULK children is walked through a two-word continuation (next child
 * index + marker) instead of one stack word per child: the stack then grows with the depth of the graph, not
 * with the length of a flat list (a world of hundreds of objects without a bvh_node around it) */
#define RT_LIST_BULK 8
#define RT_LIST_MARK 0x40000000 /* stack word: continuation of list node (word & ~RT_LIST_MARK); node indices stay below */

/* geometry/aabb.h:31-48 with ray.h's cached inv_dir / dir_sign */
RT_DEV bool aabb_hit(const double* b, V3 o, V3 inv, Real t_min, Real t_max) {
    {
        Real t0 = (b[0] - o.x) * inv.x, t1 = (b[3] - o.x) * inv.x;
        if (inv.x < 0) {
            Real s = t0;
            t0 = t1, t1 = s;
        }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return false;
    }
    {
        Real t0 = (b[1] - o.y) * inv.y, t1 = (b[4] - o.y) * inv.y;
        if (inv.y < 0) {
            Real s = t0;
            t0 = t1, t1 = s;
        }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return false;
    }
    {
        Real t0 = (b[2] - o.z) * inv.z, t1 = (b[5] - o.z) * inv.z;
        if (inv.z < 0) {
            Real s = t0;
            t0 = t1, t1 = s;
        }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return false;
    }
    return true;
}

RT_DEV void sphere_uv(V3 p, Real& u, Real& v) { /* geometry/sphere.h:24-30 */
    Real theta = acos(-p.y);
    Real phi = atan2(-p.z, p.x) + RT_PI;
    u = phi / (2 * RT_PI);
    v = theta / RT_PI;
}

/* ---- primitive tests shared by both traversals ------------------------------------------------ */
/* x?_rect::hit (geometry/aarect.h:79-135): t and the in-plane coordinates (a, b) */
template <bool WAVE_EXIT = false, bool SHARED = false>
RT_DEV bool rect_hit_axes(const rtr_node& n, Real ok, Real dk, Real rk, const RayDiv& q, Real oa, Real da, Real ob, Real db,
                          Real tmin, Real tmax, Real& t, Real& a, Real& b) {
    /* same tests as aarect.h:80-88 (a NaN fails none of the rejects there, nor here), evaluated
     * without per-lane early exits: lanes of a wave diverge on them anyway */
    t = div_shared<SHARED>(n.f[4] - ok, dk, rk, q.guard);
    const bool off = (t < tmin) | (t > tmax);
    /* WAVE_EXIT (shadow rays): the one exit that costs no divergence -- no lane of the wave reaches the plane
     * inside its interval, e.g. every shadow ray against the walls of the room it starts in */
    if (WAVE_EXIT && !__any(!off)) return false;
    a = oa + t * da;
    b = ob + t * db;
    const bool out = off | (a < n.f[0]) | (a > n.f[1]) | (b < n.f[2]) | (b > n.f[3]);
    return !out;
}
template <bool WAVE_EXIT = false, bool SHARED = false>
RT_DEV bool rect_hit_t(const rtr_node& n, int type, V3 o, V3 d, const RayDiv& q, Real tmin, Real tmax, Real& t, Real& a,
                       Real& b) {
    /* one copy of the test per orientation: `type` is wave-uniform wherever the record came through
     * scalar loads, so this is a scalar branch, and no copy shuffles ray components through moves */
    /* (the empty asm statements differ, which keeps the optimiser from folding the copies back into one
     * body behind a chain of selects) */
    bool hit;
    if (type == RTR_NODE_XY_RECT) {
        hit = rect_hit_axes<WAVE_EXIT, SHARED>(n, o.z, d.z, q.rz, q, o.x, d.x, o.y, d.y, tmin, tmax, t, a, b);
        asm volatile("; xy_rect" : "+v"(t));
    } else if (type == RTR_NODE_XZ_RECT) {
        hit = rect_hit_axes<WAVE_EXIT, SHARED>(n, o.y, d.y, q.ry, q, o.x, d.x, o.z, d.z, tmin, tmax, t, a, b);
        asm volatile("; xz_rect" : "+v"(t));
    } else {
        hit = rect_hit_axes<WAVE_EXIT, SHARED>(n, o.x, d.x, q.rx, q, o.y, d.y, o.z, d.z, tmin, tmax, t, a, b);
        asm volatile("; yz_rect" : "+v"(t));
    }
    return hit;
}
RT_DEV void rect_fill(const rtr_node& n, int type, V3 o, V3 d, Real t, Real a, Real b, bool needs_uv, Hit& rec) {
    if (needs_uv) {
        rec.u = (a - n.f[0]) / (n.f[1] - n.f[0]);
        rec.v = (b - n.f[2]) / (n.f[3] - n.f[2]);
    }
    rec.t = t;
    V3 outward = mk(type == RTR_NODE_YZ_RECT ? 1.0 : 0.0, type == RTR_NODE_XZ_RECT ? 1.0 : 0.0,
                    type == RTR_NODE_XY_RECT ? 1.0 : 0.0);
    set_face_normal(rec, d, outward);
    rec.mat = n.a;
    rec.p = add(o, scl(t, d));
}
/* sphere::hit / moving_sphere::hit (geometry/sphere.h:33-60, moving_sphere.h:32-62) */
RT_DEV void sphere_geom(const rtr_node& n, int type, Real time, V3& center, Real& radius) {
    if (type == RTR_NODE_SPHERE) {
        center = ld3(n.f);
        radius = n.f[3];
    } else {
        V3 c0 = ld3(n.f), c1 = ld3(n.f + 3);
        center = add(c0, scl((time - n.f[6]) / (n.f[7] - n.f[6]), sub(c1, c0)));
        radius = n.f[8];
    }
}
template <bool SHARED>
RT_DEV bool sphere_hit_t(V3 center, Real radius, V3 o, V3 d, const RayDiv& q, Real tmin, Real tmax, Real& t) {
    V3 oc = sub(o, center);
    Real a = SHARED ? q.a : len2(d); /* the same sum either way: |d|^2 of this frame, computed once */
    Real half_b = dot(oc, d);
    Real c = len2(oc) - radius * radius;
    Real discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return false;
    Real sqrtd = __builtin_sqrt(discriminant);
    t = div_shared<SHARED>(-half_b - sqrtd, a, q.ra, q.guard);
    if (t < tmin || t > tmax) {
        t = div_shared<SHARED>(-half_b + sqrtd, a, q.ra, q.guard);
        if (t < tmin || t > tmax) return false;
    }
    return true;
}
RT_DEV bool sphere_hit_t(V3 center, Real radius, V3 o, V3 d, Real tmin, Real tmax, Real& t) {
    return sphere_hit_t<false>(center, radius, o, d, raydiv_none(), tmin, tmax, t);
}
RT_DEV void sphere_fill(const rtr_node& n, int type, V3 center, Real radius, V3 o, V3 d, Real t, bool needs_uv,
                        Hit& rec) {
    rec.t = t;
    rec.p = add(o, scl(t, d));
    V3 outward = divs(sub(rec.p, center), radius);
    set_face_normal(rec, d, outward);
    if (type == RTR_NODE_SPHERE && needs_uv) sphere_uv(outward, rec.u, rec.v); /* moving_sphere sets no u,v */
    rec.mat = n.a;
}
/* the part of translate::hit / rotate_y::hit / flip_face::hit that runs after the child hit
 * (geometry/hittable.h:58-61,142-155,168); `d` is the direction of the ray the wrapper passed down */
RT_DEV void wrapper_epilogue(const rtr_node& n, V3 d, Hit& rec) {
    const int type = n.type;
    if (type == RTR_NODE_TRANSLATE) {
        rec.p = add(rec.p, ld3(n.f));
        set_face_normal(rec, d, rec.n);
    } else if (type == RTR_NODE_ROTATE_Y) {
        const Real s = n.f[0], c = n.f[1];
        V3 p = rec.p, nn = rec.n;
        p.x = c * rec.p.x + s * rec.p.z;
        p.z = -s * rec.p.x + c * rec.p.z;
        nn.x = c * rec.n.x + s * rec.n.z;
        nn.z = -s * rec.n.x + c * rec.n.z;
        rec.p = p;
        set_face_normal(rec, d, nn);
    } else {
        rec.front = !rec.front;
    }
}
/* the ray translate::hit / rotate_y::hit pass down (hittable.h:53,128-138) */
RT_DEV void wrapper_enter(int type, const double* f, V3& o, V3& d) {
    if (type == RTR_NODE_TRANSLATE) {
        o = sub(o, ld3(f));
    } else {
        const Real s = f[0], c = f[1];
        const Real ox = c * o.x - s * o.z, oz = s * o.x + c * o.z;
        const Real dx = c * d.x - s * d.z, dz = s * d.x + c * d.z;
        o.x = ox, o.z = oz, d.x = dx, d.z = dz;
    }
}

/* defined below (order-free traversal of compiled sub-scenes) */
template <bool ANY, bool TREES = true, bool WEXIT = ANY, bool TOP = false, bool GUARD = false>
__device__ __forceinline__ bool trace_fast(const DScene& sc, const FSub sub, V3 o, V3 d, Real time, Real tmin, Real& tmax,
                                           int& hit_ref, int& hit_inst, const Stack st, const int sp0);
/* sub-scene 0 (the whole scene where it has no media) out of the scene record itself */
RT_DEV FSub sub_scene0(const DScene& sc) {
    FSub s;
    s.inst_first = 0, s.n_inst = sc.n_finst;
    s.top_root = sc.top_root0, s.world_inst = sc.world_inst0, s.world_linear = sc.world_linear0, s.top_bound = sc.top_bound0;
    s.pad[0] = s.pad[1] = 0;
    return s;
}
template <bool UV>
RT_DEV void fast_finish(const DScene& sc, V3 o, V3 d, Real time, Real t, int ref, int inst, Hit& rec);

/*
 * Closest hit of the hittable graph under `root`: an iterative restatement of the
 * reference's recursive virtual hit() calls that visits objects in the same order
 * (bvh_node: left then right, right limited by the left's t, geometry/bvh.h:40-50;
 * hittable_list in order, hittable_list.h:33-47), so ties resolve the same way and the
 * RNG draws of constant_medium::hit happen in the same sequence.
 *
 * FULL = false computes only whether/where (t) something is hit (shadow rays, medium
 * boundaries); MEDIA enables constant_medium nodes (never nested: checked at upload).  Returns whether anything was hit; `tmax` returns the hit t.
 */
template <bool FULL, bool MEDIA>
__device__ __forceinline__ bool traverse(const DScene& sc, int root, V3 o, V3 d, Real time, Real tmin, Real& tmax,
                                         Hit& rec, uint32_t& rng, const Stack st, const int sp0) {
    V3 inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z); /* core/ray.h:11-12 */
    int hits = 0;
    int sp = sp0;
    st.put(sp++, root);
    while (sp > sp0) {
        const int e = st.get(--sp);
        if (e < 0) { /* leaving wrapper node -(e+1) */
            const rtr_node n = ld_const(sc.nodes, -(e + 1));
            const bool inside = st.get(--sp) != hits;
            const int type = n.type;
            if (FULL && inside) wrapper_epilogue(n, d, rec);
            if (type == RTR_NODE_TRANSLATE) {
                sp -= 6;
                o = mk(st.getd(sp), st.getd(sp + 2), st.getd(sp + 4));
            } else if (type == RTR_NODE_ROTATE_Y) {
                sp -= 12;
                o.x = st.getd(sp), o.z = st.getd(sp + 2);
                d.x = st.getd(sp + 4), d.z = st.getd(sp + 6);
                inv.x = st.getd(sp + 8), inv.z = st.getd(sp + 10);
            }
            continue;
        }
        if (e >= RT_LIST_MARK) { /* next child of a long hittable_list (hittable_list.h:38-44: in order) */
            const int list = e & ~RT_LIST_MARK;
            const rtr_node ln = ld_const(sc.nodes, list);
            const int k = st.get(--sp);
            if (k + 1 < ln.b) {
                st.put(sp++, k + 1);
                st.put(sp++, e);
            }
            st.put(sp++, as_const(sc.list_children)[ln.a + k]);
            continue;
        }
        const rtr_node n = ld_const(sc.nodes, e);
        const int type = n.type;
        if (type == RTR_NODE_BVH) {
            if (aabb_hit(n.f, o, inv, tmin, tmax)) {
                st.put(sp++, n.b);
                st.put(sp++, n.a);
            }
        } else if (type >= RTR_NODE_XY_RECT && type <= RTR_NODE_YZ_RECT) {
            Real t, a, b;
            if (rect_hit_t(n, type, o, d, raydiv_none(), tmin, tmax, t, a, b)) {
                tmax = t;
                ++hits;
                if (FULL) rect_fill(n, type, o, d, t, a, b, sc.needs_uv != 0, rec);
            }
        } else if (type == RTR_NODE_SPHERE || type == RTR_NODE_MOVING_SPHERE) {
            V3 center;
            Real radius, t;
            sphere_geom(n, type, time, center, radius);
            if (sphere_hit_t(center, radius, o, d, tmin, tmax, t)) {
                tmax = t;
                ++hits;
                if (FULL) sphere_fill(n, type, center, radius, o, d, t, sc.needs_uv != 0, rec);
            }
        } else if (type == RTR_NODE_LIST) {
            if (n.b > RT_LIST_BULK) {
                st.put(sp++, 0);
                st.put(sp++, e | RT_LIST_MARK);
            } else {
                for (int k = n.b - 1; k >= 0; --k) st.put(sp++, as_const(sc.list_children)[n.a + k]);
            }
        } else if (type == RTR_NODE_TRANSLATE) { /* geometry/hittable.h:51-56 */
            st.putd(sp, o.x), st.putd(sp + 2, o.y), st.putd(sp + 4, o.z);
            sp += 6;
            st.put(sp++, hits);
            st.put(sp++, -(e + 1));
            wrapper_enter(type, n.f, o, d);
            st.put(sp++, n.a);
        } else if (type == RTR_NODE_ROTATE_Y) { /* geometry/hittable.h:127-140 */
            st.putd(sp, o.x), st.putd(sp + 2, o.z), st.putd(sp + 4, d.x), st.putd(sp + 6, d.z);
            st.putd(sp + 8, inv.x), st.putd(sp + 10, inv.z);
            sp += 12;
            st.put(sp++, hits);
            st.put(sp++, -(e + 1));
            wrapper_enter(type, n.f, o, d);
            inv.x = 1.0 / d.x, inv.z = 1.0 / d.z;
            st.put(sp++, n.a);
        } else if (type == RTR_NODE_FLIP_FACE) {
            st.put(sp++, hits);
            st.put(sp++, -(e + 1));
            st.put(sp++, n.a);
        } else if (MEDIA && type == RT_NODE_COMPILED) { /* a media-free subtree, compiled (see FSub) */
            const FSub sub = ld_const(sc.fsub, n.a);
            int ref, inst;
            if (trace_fast<false>(sc, sub, o, d, time, tmin, tmax, ref, inst, st, sp)) {
                ++hits;
                if (FULL) {
                    if (sc.needs_uv)
                        fast_finish<true>(sc, o, d, time, tmax, ref, inst, rec);
                    else
                        fast_finish<false>(sc, o, d, time, tmax, ref, inst, rec);
                }
            }
        } else if (MEDIA && type == RTR_NODE_MEDIUM) { /* geometry/constant_medium.h:55-104 */
            Hit dummy;
            Real t1 = RT_INF;
            if (traverse<false, false>(sc, n.a, o, d, time, -RT_INF, t1, dummy, rng, st, sp)) {
                Real t2 = RT_INF;
                if (traverse<false, false>(sc, n.a, o, d, time, t1 + 0.0001, t2, dummy, rng, st, sp)) {
                    if (t1 < tmin) t1 = tmin;
                    if (t2 > tmax) t2 = tmax;
                    if (!(t1 >= t2)) {
                        if (t1 < 0) t1 = 0;
                        const Real ray_length = len(d);
                        const Real distance_inside_boundary = (t2 - t1) * ray_length;
                        const Real hit_distance = n.f[0] * log(rng_next(rng));
                        if (!(hit_distance > distance_inside_boundary)) {
                            tmax = t1 + hit_distance / ray_length;
                            ++hits;
                            if (FULL) {
                                rec.t = tmax;
                                rec.p = add(o, scl(tmax, d));
                                rec.n = mk(1, 0, 0);
                                rec.front = true;
                                rec.mat = n.b;
                            }
                        }
                    }
                }
            }
        }
    }
    return hits > 0;
}

/* ---- order-free traversal of the compiled scene ---------------------------------------------- */
#define RT_FAST_NO_BOX_MAX 4
/* slab test returning the entry distance (any conservative box test is valid here: the boxes
 * are padded at build time and only prune work) */
RT_DEV bool box_enter(const double* bmin, const double* bmax, V3 o, V3 inv, Real tmin, Real tmax, Real& tnear) {
    Real x0 = (bmin[0] - o.x) * inv.x, x1 = (bmax[0] - o.x) * inv.x;
    Real y0 = (bmin[1] - o.y) * inv.y, y1 = (bmax[1] - o.y) * inv.y;
    Real z0 = (bmin[2] - o.z) * inv.z, z1 = (bmax[2] - o.z) * inv.z;
    Real lo = __builtin_fmax(__builtin_fmax(__builtin_fmin(x0, x1), __builtin_fmin(y0, y1)),
                             __builtin_fmax(__builtin_fmin(z0, z1), tmin));
    Real hi = __builtin_fmin(__builtin_fmin(__builtin_fmax(x0, x1), __builtin_fmax(y0, y1)),
                             __builtin_fmin(__builtin_fmax(z0, z1), tmax));
    tnear = lo;
    return lo <= hi; /* NaNs (0 * inf) drop out of fmin/fmax, which keeps the test conservative */
}

/* Conservative single-precision slab test for the box trees.  For axis k the entry / exit
 * parameters are fma(plane_k, idir_k, c_k) with idir = float(1/d), c_near = float(-o/d - slack),
 * c_far = float(-o/d + slack) and slack = 2^-21 |1/d| (|o| + bound): that covers the rounding of
 * idir (2^-24 |plane/d|), of c (2^-24 |o/d|) and of the fma itself with a factor 4 to spare, so the
 * computed interval always CONTAINS the exact one and no box the double-precision ray enters is
 * ever skipped; it widens a box by ~5e-7 of the scene size.  An axis the ray is (nearly) parallel to
 * (|1/d| >= 1e30, inf or NaN) is switched off (idir = 0, c = -/+inf), which is conservative too. */
struct BoxRay {
    float idx, idy, idz;
    float cnx, cny, cnz;
    float cfx, cfy, cfz;
    bool sx, sy, sz; /* direction negative: the box's max plane is the near one */
};
RT_DEV void boxray_axis(Real o, Real inv, float bound, float& idir, float& cn, float& cf, bool& neg_dir) {
    const Real ainv = __builtin_fabs(inv);
    const bool usable = ainv < 1e30; /* false for inf and NaN as well */
    const Real slack = 0x1p-21 * ainv * (__builtin_fabs(o) + (Real)bound);
    const Real mid = -o * inv;
    idir = usable ? (float)inv : 0.0f;
    cn = usable ? (float)(mid - slack) : -__builtin_huge_valf();
    cf = usable ? (float)(mid + slack) : __builtin_huge_valf();
    neg_dir = inv < 0;
}
/* `inv` = 1 / d up to a few ulps (the slack covers 2^-24 of it) */
RT_DEV BoxRay boxray_make_inv(V3 o, V3 inv, float bound) {
    BoxRay r;
    boxray_axis(o.x, inv.x, bound, r.idx, r.cnx, r.cfx, r.sx);
    boxray_axis(o.y, inv.y, bound, r.idy, r.cny, r.cfy, r.sy);
    boxray_axis(o.z, inv.z, bound, r.idz, r.cnz, r.cfz, r.sz);
    return r;
}
RT_DEV BoxRay boxray_make(V3 o, V3 d, float bound) { return boxray_make_inv(o, mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z), bound); }
/* round a double bound of the ray interval to float, outward */
RT_DEV float float_below(Real t) {
    const float f = (float)t;
    return __builtin_fmaf(-__builtin_fabsf(f), 0x1p-23f, f);
}
RT_DEV float float_above(Real t) {
    const float f = (float)t;
    return __builtin_fmaf(__builtin_fabsf(f), 0x1p-23f, f);
}
/* One FBvh record through four 16-byte loads.  (Reading the fields one by one lets the optimiser turn
 * the per-ray choice between a box's min and max plane into a choice between two ADDRESSES: twelve
 * dword loads in three dependent groups per node -- three memory round trips instead of one.) */
struct NodeRegs {
    float lmin[3], lmax[3], rmin[3], rmax[3];
    int left, right;
};
RT_DEV NodeRegs load_node(const FBvh* nodes, int node) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    /* global address space: a plain pointer read out of DScene would make these flat loads */
    const __attribute__((address_space(1))) f32x4* q =
        (const __attribute__((address_space(1))) f32x4*)(unsigned long long)(nodes + node);
    const f32x4 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
    NodeRegs n;
    n.lmin[0] = w0.x, n.lmin[1] = w0.y, n.lmin[2] = w0.z, n.lmax[0] = w0.w;
    n.lmax[1] = w1.x, n.lmax[2] = w1.y, n.rmin[0] = w1.z, n.rmin[1] = w1.w;
    n.rmin[2] = w2.x, n.rmax[0] = w2.y, n.rmax[1] = w2.z, n.rmax[2] = w2.w;
    n.left = __float_as_int(w3.x), n.right = __float_as_int(w3.y);
    return n;
}
RT_DEV bool boxray_hit(const BoxRay& r, const float* bmin, const float* bmax, float tmin, float tmax, float& tnear) {
    const float nx = r.sx ? bmax[0] : bmin[0], fx = r.sx ? bmin[0] : bmax[0];
    const float ny = r.sy ? bmax[1] : bmin[1], fy = r.sy ? bmin[1] : bmax[1];
    const float nz = r.sz ? bmax[2] : bmin[2], fz = r.sz ? bmin[2] : bmax[2];
    /* fmaxf / fminf drop NaNs (0 * inf): such an axis does not constrain the interval */
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(nx, r.idx, r.cnx), __builtin_fmaf(ny, r.idy, r.cny)),
                                     __builtin_fmaxf(__builtin_fmaf(nz, r.idz, r.cnz), tmin));
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaf(fx, r.idx, r.cfx), __builtin_fmaf(fy, r.idy, r.cfy)),
                                     __builtin_fminf(__builtin_fmaf(fz, r.idz, r.cfz), tmax));
    tnear = tn;
    return tn <= tf;
}

/* One reference of an instance against the ray in the instance frame.  (Exact ties in t, e.g. the
 * coplanar side faces of adjacent boxes in scene 9, are won by whichever primitive is tested
 * last; in the reference that is decided by 1-ulp noise of its BVH box tests, so neither order
 * can be called "the" reference behaviour.  Such faces share material and normal there.) */
template <bool WAVE_EXIT = false, bool SHARED = false>
RT_DEV bool fast_prim_hit(const rtr_node& n, V3 o, V3 d, const RayDiv& q, Real time, Real tmin, Real tmax, Real& t) {
    const int type = n.type;
    if (type >= RTR_NODE_XY_RECT) {
        Real a, b;
        return rect_hit_t<WAVE_EXIT, SHARED>(n, type, o, d, q, tmin, tmax, t, a, b);
    }
    V3 center;
    Real radius;
    sphere_geom(n, type, time, center, radius);
    return sphere_hit_t<SHARED>(center, radius, o, d, q, tmin, tmax, t);
}
/* Exact ties in t.  Every hit() of the reference accepts t == t_max, so of two surfaces at exactly the
 * same t the one its walk visits LATER wins (two coplanar rects with different materials).  Upload marks
 * the references that can tie with another one of their instance (same plane and overlapping extent,
 * or the same sphere twice) with RT_TIE_FLAG and stores their visiting position next to it; only those
 * pay for the comparison: a tie is accepted only from a primitive visited later than the current
 * holder.  Across instances the scan order is the visiting order of each instance's first primitive and
 * `t == t_max` is accepted, so the later instance wins like the reference's later visit; coplanar rects
 * whose instance order contradicts their visiting order are flagged as well (rtr_upload_scene), their
 * positions are comparable because `order` lives across the instance loop.  Not reproduced: coplanar
 * rotated faces of two different transform chains, and ties where the reference's own choice hangs on
 * 1-ulp noise of an UNPADDED box test (faces of `box` objects that touch: the later one is only
 * visited if aabb::hit of its box, entered at exactly that t, survives `t_max <= t_min`). */
#define RT_TIE_FLAG (1 << 30)
/* References behind box tests of the reference's bvh_nodes (RT_GUARD_FLAG): a sphere with a negative radius -- hollow
 * glass, scenes.cpp:903 -- has an inverted bounding box (sphere.h:62-66), the boxes of the bvh_nodes above it do not
 * enclose it, and bvh_node::hit (bvh.h:40-50) lets a ray through to it only if every one of them is hit within
 * [t_min, closest t so far] (aabb.h:31-48, with ray.h's 1 / d).  In a sub-scene scanned in the reference's visiting
 * order the running t_max IS that "closest so far" (rt_compile.h: guard_mode). */
#define RT_GUARD_FLAG (1 << 29) /* in rtr_node::reserved / FLeaf::tag of a reference, next to RT_TIE_FLAG */
RT_DEV bool guard_pass(const DScene& sc, int first, int count, V3 o, V3 d, Real tmin, Real tmax) {
    const V3 inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    bool enter = true;
    for (int g = 0; g < count; ++g) {
        const FGuard gb = ld_const(sc.fguard, first + g);
        enter = enter && aabb_hit(gb.b, o, inv, tmin, tmax);
    }
    return enter;
}
template <bool TIES = true, bool WAVE_EXIT = false, bool SHARED = false, bool GUARD = TIES>
RT_DEV bool fast_ref_hit(const DScene& sc, int ref, V3 o, V3 d, const RayDiv& q, Real time, Real tmin, Real tmax, Real& t,
                         int& order) {
    const rtr_node n = ld_const(sc.fprim, ref);
    const int tag = n.reserved;
    /* (guarded references exist only in scenes that take the kernels with TIES or GUARD: rtr_upload_scene) */
    if (GUARD && (tag & RT_GUARD_FLAG) &&
        !guard_pass(sc, (int)__double_as_longlong(n.f[4]), (int)__double_as_longlong(n.f[5]), o, d, tmin, tmax))
        return false;
    if (!fast_prim_hit<WAVE_EXIT, SHARED>(n, o, d, q, time, tmin, tmax, t)) return false;
    if (TIES && (tag & RT_TIE_FLAG)) {
        const int visit = tag & ~(RT_TIE_FLAG | RT_GUARD_FLAG);
        if (t == tmax && visit < order) return false;
        order = visit;
    }
    return true;
}

template <bool TIES = true, bool WAVE_EXIT = false>
RT_DEV bool fast_ref_hit(const DScene& sc, int ref, V3 o, V3 d, Real time, Real tmin, Real tmax, Real& t, int& order) {
    return fast_ref_hit<TIES, WAVE_EXIT, false>(sc, ref, o, d, raydiv_none(), time, tmin, tmax, t, order);
}

/* Closest hit (ANY = false) or first hit found (ANY = true: shadow rays only need existence).
 * Returns the reference and instance of the hit; `tmax` returns its t.  TREES = false: the caller
 * knows that no instance of the scene has a box tree and no reference is tie-capable (RT_TRAV_FLAT),
 * which keeps that code and its registers out of the kernels of small scenes (the Cornell box:
 * 118 VGPRs, no spills). */
/* ---- type runs of a linearly scanned instance (FInst::run) ------------------------------------------------------
 * One rectangle of a run against the ray: the arithmetic of rect_hit_axes on a packed record. */
template <int TYPE, bool WEXIT, bool SHARED>
RT_DEV void run_rect_test(Real a0, Real a1, Real b0, Real b1, Real k, int ref, V3 o, V3 d, const RayDiv& q, Real tmin,
                          Real& tmax, int& hit_ref) {
    const Real ok = TYPE == RTR_NODE_XY_RECT ? o.z : (TYPE == RTR_NODE_XZ_RECT ? o.y : o.x);
    const Real dk = TYPE == RTR_NODE_XY_RECT ? d.z : (TYPE == RTR_NODE_XZ_RECT ? d.y : d.x);
    const Real rk = TYPE == RTR_NODE_XY_RECT ? q.rz : (TYPE == RTR_NODE_XZ_RECT ? q.ry : q.rx);
    const Real oa = TYPE == RTR_NODE_YZ_RECT ? o.y : o.x, da = TYPE == RTR_NODE_YZ_RECT ? d.y : d.x;
    const Real ob = TYPE == RTR_NODE_XY_RECT ? o.y : o.z, db = TYPE == RTR_NODE_XY_RECT ? d.y : d.z;
    const Real t = div_shared<SHARED>(k - ok, dk, rk, q.guard);
    const bool off = (t < tmin) | (t > tmax);
    if (WEXIT && !__any(!off)) return;
    const Real a = oa + t * da;
    const Real b = ob + t * db;
    const bool out = off | (a < a0) | (a > a1) | (b < b0) | (b > b1);
    tmax = out ? tmax : t;
    hit_ref = out ? hit_ref : ref;
}
/* The same test for SHARED frames as one hand-scheduled block.  Every instruction of these loops costs a wave the same
 * issue slot, scalar or vector (tools/issue_rates.py: v_fma_f64 2.8, v_cmp_f64 2.8, s_and_b64 2.8 cycles per SIMD), and the
 * compiler's form spends eleven of its twenty-four on the acceptance mask and the conditional update: six compares into
 * scalar pairs, five scalar ORs, a move and three selects.  Here each compare narrows EXEC itself (v_cmpx), the update is
 * two moves executed by the lanes that are left, and EXEC is put back: eighteen instructions, the same arithmetic on the
 * same operands in the same order (a SHARED frame produces no NaN, so "not less" and "greater or equal" agree).
 * `tmin` is taken through a scalar pair when it is a literal of the caller (the integrators' 0.001), else per lane. */
template <int TYPE, bool WEXIT, bool LANE = false>
RT_DEV void run_rect_test_x(Real a0, Real a1, Real b0, Real b1, Real k, int ref, V3 o, V3 d, const RayDiv& q, Real tmin,
                            Real& tmax, int& hit_ref) {
    const Real ok = TYPE == RTR_NODE_XY_RECT ? o.z : (TYPE == RTR_NODE_XZ_RECT ? o.y : o.x);
    const Real dk = TYPE == RTR_NODE_XY_RECT ? d.z : (TYPE == RTR_NODE_XZ_RECT ? d.y : d.x);
    const Real rk = TYPE == RTR_NODE_XY_RECT ? q.rz : (TYPE == RTR_NODE_XZ_RECT ? q.ry : q.rx);
    const Real oa = TYPE == RTR_NODE_YZ_RECT ? o.y : o.x, da = TYPE == RTR_NODE_YZ_RECT ? d.y : d.x;
    const Real ob = TYPE == RTR_NODE_XY_RECT ? o.y : o.z, db = TYPE == RTR_NODE_XY_RECT ? d.y : d.z;
    Real n, t, e, u;
    unsigned long long save;
/* TMIN: constraint of the t_min operand ("s": scalar pair, "v": per lane); REC: that of the record's fields and the
 * reference index ("s" in the wave-uniform scans, "v" where every lane tests a record of its own: LANE); BRANCH: the
 * wave-level exit of shadow rays (no lane reaches the plane inside its interval) or nothing */
#define RT_RECT_X(TMIN, REC, BRANCH)                                                                                          \
    asm volatile("v_add_f64 %[n], %[k], -%[ok]\n\t"                                                                     \
                 "v_mul_f64 %[t], %[n], %[rk]\n\t"                                                                       \
                 "v_fma_f64 %[e], -%[dk], %[t], %[n]\n\t"                                                                \
                 "v_fma_f64 %[t], %[e], %[rk], %[t]\n\t"                                                                 \
                 "s_mov_b64 %[save], exec\n\t"                                                                           \
                 "v_cmpx_ngt_f64 vcc, %[tmin], %[t]\n\t"                                                                 \
                 "v_cmpx_ngt_f64 vcc, %[t], %[tmax]\n\t" BRANCH                                                          \
                 "v_mul_f64 %[u], %[t], %[da]\n\t"                                                                       \
                 "v_add_f64 %[u], %[oa], %[u]\n\t"                                                                       \
                 "v_cmpx_ngt_f64 vcc, %[a0], %[u]\n\t"                                                                   \
                 "v_cmpx_nlt_f64 vcc, %[a1], %[u]\n\t"                                                                   \
                 "v_mul_f64 %[u], %[t], %[db]\n\t"                                                                       \
                 "v_add_f64 %[u], %[ob], %[u]\n\t"                                                                       \
                 "v_cmpx_ngt_f64 vcc, %[b0], %[u]\n\t"                                                                   \
                 "v_cmpx_nlt_f64 vcc, %[b1], %[u]\n\t"                                                                   \
                 "v_mov_b64 %[tmax], %[t]\n\t"                                                                           \
                 "v_mov_b32 %[hit], %[ref]\n"                                                                             \
                 ".Lrx%=:\n\t"                                                                                           \
                 "s_mov_b64 exec, %[save]"                                                                                 \
                 : [n] "=&v"(n), [t] "=&v"(t), [e] "=&v"(e), [u] "=&v"(u), [save] "=&s"(save), [tmax] "+v"(tmax),             \
                   [hit] "+v"(hit_ref)                                                                                     \
                 : [k] REC(k), [ok] "v"(ok), [dk] "v"(dk), [rk] "v"(rk), [oa] "v"(oa), [da] "v"(da), [ob] "v"(ob),             \
                   [db] "v"(db), [a0] REC(a0), [a1] REC(a1), [b0] REC(b0), [b1] REC(b1), [ref] REC(ref), [tmin] TMIN(tmin)    \
                 : "vcc")
    if (LANE) {
        if (__builtin_constant_p(tmin))
            RT_RECT_X("s", "v", "");
        else
            RT_RECT_X("v", "v", "");
    } else if (__builtin_constant_p(tmin)) {
        if (WEXIT)
            RT_RECT_X("s", "s", "s_cbranch_execz .Lrx%=\n\t");
        else
            RT_RECT_X("s", "s", "");
    } else {
        if (WEXIT)
            RT_RECT_X("v", "s", "s_cbranch_execz .Lrx%=\n\t");
        else
            RT_RECT_X("v", "s", "");
    }
#undef RT_RECT_X
}
#ifndef RTR_RECT_ASM
#define RTR_RECT_ASM 1
#endif
template <int TYPE, bool WEXIT, bool SHARED, bool LANE = false>
RT_DEV void run_rect(Real a0, Real a1, Real b0, Real b1, Real k, int ref, V3 o, V3 d, const RayDiv& q, Real tmin, Real& tmax,
                     int& hit_ref) {
    if (RTR_RECT_ASM && SHARED && !q.guard)
        run_rect_test_x<TYPE, WEXIT, LANE>(a0, a1, b0, b1, k, ref, o, d, q, tmin, tmax, hit_ref);
    else
        run_rect_test<TYPE, WEXIT, SHARED>(a0, a1, b0, b1, k, ref, o, d, q, tmin, tmax, hit_ref);
}
template <int TYPE, bool WEXIT, bool SHARED, bool LANE = false>
RT_DEV void run_rects(const RT_CONST_AS double* p, int cnt, int ref, V3 o, V3 d, const RayDiv& q, Real tmin, Real& tmax,
                      int& hit_ref) {
    int k = 0;
    for (; k + 1 < cnt; k += 2, p += 10) { /* two records, one scalar-load round trip */
        const Real f0 = p[0], f1 = p[1], f2 = p[2], f3 = p[3], f4 = p[4];
        const Real g0 = p[5], g1 = p[6], g2 = p[7], g3 = p[8], g4 = p[9];
        run_rect<TYPE, WEXIT, SHARED, LANE>(f0, f1, f2, f3, f4, ref + k, o, d, q, tmin, tmax, hit_ref);
        run_rect<TYPE, WEXIT, SHARED, LANE>(g0, g1, g2, g3, g4, ref + k + 1, o, d, q, tmin, tmax, hit_ref);
    }
    if (k < cnt) run_rect<TYPE, WEXIT, SHARED, LANE>(p[0], p[1], p[2], p[3], p[4], ref + k, o, d, q, tmin, tmax, hit_ref);
}
/* A `box` (geometry/box.h:31-47): its six sides, in the order of its hittable_list, from ONE record x0 x1 y0 y1 z0 z1 --
 * one round trip and no per-run overhead for what were three runs of two; the tests are the rectangle tests above, one
 * after the other (hittable_list.h:33-47). */
#define RT_RUN_BOX 5 /* run type code (types are stored minus RTR_NODE_SPHERE) */
/* the six tests of one box record x0 x1 y0 y1 z0 z1 (any pointer kind: packed scan data or an instance record's copy) */
template <bool WEXIT, bool SHARED, bool LANE, class P>
RT_DEV void run_box(P p, int ref, V3 o, V3 d, const RayDiv& q, Real tmin, Real& tmax, int& hit_ref) {
    const Real x0 = p[0], x1 = p[1], y0 = p[2], y1 = p[3], z0 = p[4], z1 = p[5];
    run_rect<RTR_NODE_XY_RECT, WEXIT, SHARED, LANE>(x0, x1, y0, y1, z1, ref, o, d, q, tmin, tmax, hit_ref);
    run_rect<RTR_NODE_XY_RECT, WEXIT, SHARED, LANE>(x0, x1, y0, y1, z0, ref + 1, o, d, q, tmin, tmax, hit_ref);
    run_rect<RTR_NODE_XZ_RECT, WEXIT, SHARED, LANE>(x0, x1, z0, z1, y1, ref + 2, o, d, q, tmin, tmax, hit_ref);
    run_rect<RTR_NODE_XZ_RECT, WEXIT, SHARED, LANE>(x0, x1, z0, z1, y0, ref + 3, o, d, q, tmin, tmax, hit_ref);
    run_rect<RTR_NODE_YZ_RECT, WEXIT, SHARED, LANE>(y0, y1, z0, z1, x1, ref + 4, o, d, q, tmin, tmax, hit_ref);
    run_rect<RTR_NODE_YZ_RECT, WEXIT, SHARED, LANE>(y0, y1, z0, z1, x0, ref + 5, o, d, q, tmin, tmax, hit_ref);
}
template <bool WEXIT, bool SHARED, bool LANE = false>
RT_DEV void run_boxes(const RT_CONST_AS double* p, int cnt, int ref, V3 o, V3 d, const RayDiv& q, Real tmin, Real& tmax,
                      int& hit_ref) {
    for (int k = 0; k < cnt; ++k, p += 6, ref += 6) run_box<WEXIT, SHARED, LANE>(p, ref, o, d, q, tmin, tmax, hit_ref);
}
template <bool SHARED>
RT_DEV void run_spheres(const RT_CONST_AS double* p, int cnt, int ref, V3 o, V3 d, const RayDiv& q, Real tmin, Real& tmax,
                        int& hit_ref) {
    for (int k = 0; k < cnt; ++k, p += 4) {
        Real t;
        if (sphere_hit_t<SHARED>(mk(p[0], p[1], p[2]), p[3], o, d, q, tmin, tmax, t)) tmax = t, hit_ref = ref + k;
    }
}
#define RT_RUN_GUARDED 6 /* run type code (kernels with GUARD only): records c r first count */
template <bool SHARED>
RT_DEV void run_guarded_spheres(const DScene& sc, const RT_CONST_AS double* p, int cnt, int ref, V3 o, V3 d, const RayDiv& q,
                                Real tmin, Real& tmax, int& hit_ref) {
    for (int k = 0; k < cnt; ++k, p += 6) {
        const int first = (int)__double_as_longlong(p[4]), count = (int)__double_as_longlong(p[5]);
        Real t;
        if (guard_pass(sc, first, count, o, d, tmin, tmax) && sphere_hit_t<SHARED>(mk(p[0], p[1], p[2]), p[3], o, d, q, tmin, tmax, t))
            tmax = t, hit_ref = ref + k;
    }
}
template <bool SHARED>
RT_DEV void run_moving_spheres(const RT_CONST_AS double* p, int cnt, int ref, V3 o, V3 d, const RayDiv& q, Real time, Real tmin,
                               Real& tmax, int& hit_ref) {
    for (int k = 0; k < cnt; ++k, p += 9) {
        const V3 c0 = mk(p[0], p[1], p[2]), c1 = mk(p[3], p[4], p[5]);
        const V3 center = add(c0, scl((time - p[6]) / (p[7] - p[6]), sub(c1, c0))); /* moving_sphere.h:32-34 */
        Real t;
        if (sphere_hit_t<SHARED>(center, p[8], o, d, q, tmin, tmax, t)) tmax = t, hit_ref = ref + k;
    }
}
/* every run of the instance, in visiting order (`t == t_max` is accepted: the later reference wins an exact tie) */
template <bool WEXIT, bool SHARED, bool LANE = false, bool GUARD = false>
RT_DEV void scan_runs(const DScene& sc, const FInst& I, V3 o, V3 d, const RayDiv& q, Real time, Real tmin, Real& tmax,
                      int& hit_ref) {
    const RT_CONST_AS double* p = as_const(sc.fscan) + I.scan_first;
    int ref = I.ref_first;
#pragma nounroll
    for (uint64_t runs = I.runs; runs != 0; runs >>= RT_RUN_BITS) {
        const int type = RTR_NODE_SPHERE + (int)((runs >> RT_RUN_COUNT_BITS) & 7), cnt = (int)(runs & RT_RUN_COUNT_MAX);
        RT_REGION(type >= RTR_NODE_XY_RECT ? (WEXIT ? RG_SH_RECTS : RG_RECTS) : (WEXIT ? RG_SH_SPHERES : RG_SPHERES)); /* boxes count as rects */
        if (type == RTR_NODE_XY_RECT) {
            run_rects<RTR_NODE_XY_RECT, WEXIT, SHARED, LANE>(p, cnt, ref, o, d, q, tmin, tmax, hit_ref);
            p += 5 * cnt;
        } else if (type == RTR_NODE_XZ_RECT) {
            run_rects<RTR_NODE_XZ_RECT, WEXIT, SHARED, LANE>(p, cnt, ref, o, d, q, tmin, tmax, hit_ref);
            p += 5 * cnt;
        } else if (type == RTR_NODE_YZ_RECT) {
            run_rects<RTR_NODE_YZ_RECT, WEXIT, SHARED, LANE>(p, cnt, ref, o, d, q, tmin, tmax, hit_ref);
            p += 5 * cnt;
        } else if (type == RTR_NODE_SPHERE + RT_RUN_BOX) {
            run_boxes<WEXIT, SHARED, LANE>(p, cnt, ref, o, d, q, tmin, tmax, hit_ref);
            p += 6 * cnt;
            ref += 5 * cnt; /* six references per box */
        } else if (type == RTR_NODE_SPHERE) {
            run_spheres<SHARED>(p, cnt, ref, o, d, q, tmin, tmax, hit_ref);
            p += 4 * cnt;
        } else if (GUARD && type == RTR_NODE_SPHERE + RT_RUN_GUARDED) {
            run_guarded_spheres<SHARED>(sc, p, cnt, ref, o, d, q, tmin, tmax, hit_ref);
            p += 6 * cnt;
        } else {
            run_moving_spheres<SHARED>(p, cnt, ref, o, d, q, time, tmin, tmax, hit_ref);
            p += 9 * cnt;
        }
        ref += cnt;
    }
}

/* ---- leaf references of a box tree (per-lane records) ---------------------------------------------------------- */
RT_DEV FLeaf load_leaf(const FLeaf* base, int ref) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const __attribute__((address_space(1))) u64x2* q = (const __attribute__((address_space(1))) u64x2*)(unsigned long long)(base + ref);
    const u64x2 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
    FLeaf L;
    L.f[0] = __longlong_as_double(w0.x), L.f[1] = __longlong_as_double(w0.y);
    L.f[2] = __longlong_as_double(w1.x), L.f[3] = __longlong_as_double(w1.y);
    L.f[4] = __longlong_as_double(w2.x), L.f[5] = __longlong_as_double(w2.y);
    L.type = (int)(w3.x & 0xffffffffu), L.tag = (int)(w3.x >> 32);
    L.pad[0] = L.pad[1] = 0;
    return L;
}
/* the references [r0, r1) of one leaf */
template <bool TIES, bool WEXIT, bool ANY>
RT_DEV bool leaf_refs(const DScene& sc, int r0, int r1, V3 o, V3 d, Real time, Real tmin, Real& tmax, int& hit_ref, int& order) {
    for (int r = r0; r < r1; ++r) {
        const FLeaf L = load_leaf(sc.fleaf, r);
        const int type = L.type;
        Real t;
        bool hit;
        if (type >= RTR_NODE_XY_RECT) {
            rtr_node n; /* the fields rect_hit_t reads */
            n.f[0] = L.f[0], n.f[1] = L.f[1], n.f[2] = L.f[2], n.f[3] = L.f[3];
            n.f[4] = L.f[4];
            Real a, b;
            hit = rect_hit_t<WEXIT, false>(n, type, o, d, raydiv_none(), tmin, tmax, t, a, b);
        } else if (type == RTR_NODE_SPHERE) {
            hit = sphere_hit_t(mk(L.f[0], L.f[1], L.f[2]), L.f[3], o, d, tmin, tmax, t);
        } else {
            const rtr_node n = ld_const(sc.fprim, r);
            V3 center;
            Real radius;
            sphere_geom(n, type, time, center, radius);
            hit = sphere_hit_t(center, radius, o, d, tmin, tmax, t);
        }
        if (!hit) continue;
        if (TIES && (L.tag & RT_TIE_FLAG)) { /* see fast_ref_hit */
            const int visit = L.tag & ~(RT_TIE_FLAG | RT_GUARD_FLAG);
            if (t == tmax && visit < order) continue;
            order = visit;
        }
        tmax = t;
        hit_ref = r;
        if (ANY) return true;
    }
    return false;
}

#ifndef RTR_MEDIUM_SHARED
#define RTR_MEDIUM_SHARED 0 /* 1: the sphere-bounded media of a step program divide by |d|^2 through one reciprocal (measured: 1 530 -> 1 499, 792 -> 766 Msamples/s on scenes 9 / 22: registers) */
#endif
#ifndef RTR_TREE_SHARED
#define RTR_TREE_SHARED 0 /* 1: kernels with box trees share reciprocals too (linear scans and leaf primitives) */
#endif
/* The references of one instance against the ray in the instance's frame: linear scan or box tree.  Returns true when
 * an ANY cast has found its hit.  SHARED: see div_shared. */
template <bool ANY, bool TREES, bool WEXIT, bool SHARED, bool GUARD = false>
__device__ __forceinline__ bool scan_instance(const DScene& sc, const FInst& I, V3 lo, V3 ld, const RayDiv& q, Real time,
                                              Real tmin, Real& tmax, int& hit_ref, int& order, const Stack st, const int sp0) {
    if (!TREES || I.bvh_root < 0) {
        if (SHARED && (I.flags & RT_INST_RUNS)) {
            /* (an ANY cast goes on after its first hit: a lane-level exit inside these loops costs every trip,
             * and most shadow rays reach their light) */
            scan_runs<WEXIT, SHARED, false, GUARD>(sc, I, lo, ld, q, time, tmin, tmax, hit_ref);
            return ANY && hit_ref >= 0;
        }
        RT_REGION(WEXIT ? RG_SH_GENERIC : RG_GENERIC);
        const int r0 = I.ref_first, r1 = r0 + I.n_ref;
        for (int r = r0; r < r1; ++r) {
            Real t;
            if (fast_ref_hit<TREES, WEXIT, false, TREES || GUARD>(sc, r, lo, ld, q, time, tmin, tmax, t, order)) {
                tmax = t;
                hit_ref = r;
                if (ANY) return true;
            }
        }
        return false;
    }
    /* while-while traversal: every lane first walks down to its next leaf, then the wave
     * tests leaf primitives together (the expensive, exact part) */
    const BoxRay br = boxray_make_inv(lo, SHARED ? mk(q.rx, q.ry, q.rz) : mk(1.0 / ld.x, 1.0 / ld.y, 1.0 / ld.z), I.bound);
    const float tmin_f = float_below(tmin);
    float tmax_f = float_above(tmax);
    int sp = sp0;
    int node = I.bvh_root;
    while (true) {
        RT_REGION(WEXIT ? RG_SH_TREE : RG_TREE);
        while (node >= 0) {
            const NodeRegs b = load_node(sc.fbvh, node);
            float tl, tr;
            const bool hl = boxray_hit(br, b.lmin, b.lmax, tmin_f, tmax_f, tl);
            const bool hr = boxray_hit(br, b.rmin, b.rmax, tmin_f, tmax_f, tr);
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
                node = sp > sp0 ? st.get(--sp) : RT_BVH_DONE;
            }
        }
        if (node == RT_BVH_DONE) break;
        RT_REGION(WEXIT ? RG_SH_LEAVES : RG_LEAVES);
        const int code = -1 - node;
        const int r0 = code >> 3, r1 = r0 + (code & 7) + 1;
        const int before = hit_ref;
        if (leaf_refs<TREES, WEXIT, ANY>(sc, r0, r1, lo, ld, time, tmin, tmax, hit_ref, order)) return true;
        if (hit_ref != before) tmax_f = float_above(tmax);
        node = sp > sp0 ? st.get(--sp) : RT_BVH_DONE;
    }
    return false;
}

/* The per-lane walk of a sub-scene's top tree (struct FSub): inner nodes like scan_instance's, then every lane that
 * has reached a leaf handles ITS leaf -- references of the untransformed instance in the world frame, or one transformed
 * instance: its record through vector loads, the ray into its frame, its references (or its own box tree, nested on
 * the same stack).  Visiting order is whatever the tree gives; exact ties in t are decided by the visiting positions
 * of the tagged references like everywhere else (RT_TIE_FLAG; rtr_upload_scene tags every pair of coplanar rects of
 * two instances of such a sub-scene, whatever their order). */
template <bool ANY>
__device__ __forceinline__ bool trace_top(const DScene& sc, const FSub sub, V3 o, V3 d, Real time, Real tmin, Real& tmax,
                                          int& hit_ref, int& hit_inst, int& order, const Stack st, const int sp0) {
    const BoxRay br = boxray_make(o, d, sub.top_bound);
    const float tmin_f = float_below(tmin);
    float tmax_f = float_above(tmax);
    const bool origin_ok = sc.shared_div != 0 && raydiv_origin_ok(o);
    int sp = sp0;
    int node = sub.top_root;
    while (true) {
        RT_REGION(ANY ? RG_SH_TREE : RG_TREE);
        while (node >= 0) {
            const NodeRegs b = load_node(sc.fbvh, node);
            float tl, tr;
            const bool hl = boxray_hit(br, b.lmin, b.lmax, tmin_f, tmax_f, tl);
            const bool hr = boxray_hit(br, b.rmin, b.rmax, tmin_f, tmax_f, tr);
            const int cl = b.left, cr = b.right;
            if (hl && hr) {
                const bool left_first = tl <= tr;
                st.put(sp++, left_first ? cr : cl);
                node = left_first ? cl : cr;
            } else if (hl) {
                node = cl;
            } else if (hr) {
                node = cr;
            } else {
                node = sp > sp0 ? st.get(--sp) : RT_BVH_DONE;
            }
        }
        if (node == RT_BVH_DONE) break;
        RT_REGION(ANY ? RG_SH_LEAVES : RG_LEAVES);
        const int code = -1 - node;
        const int before = hit_ref;
        V3 lo = o, ld = d;
        int owner = sub.world_inst;
        FInst I{}; /* what the leaf code below reads of it */
        I.ref_first = code >> 3, I.n_ref = (code & 7) + 1, I.bvh_root = -1;
        if (code & RT_TOP_INST) {
            owner = code & (RT_TOP_INST - 1);
            I = ld_const(sc.finst, owner);
            wrapper_enter(I.xf_type[0], I.xf_f[0], lo, ld);
            if (I.n_xf > 1) wrapper_enter(I.xf_type[1], I.xf_f[1], lo, ld);
            for (int k = RT_INST_XF_INLINE; k < I.n_xf; ++k) {
                const FXf x = ld_const(sc.fxf, I.xf_first + k);
                wrapper_enter(x.type, x.f, lo, ld);
            }
        }
        bool found;
        if (I.bvh_root >= 0) { /* an instance with a box tree of its own: the same walk one level down, above this one's stack */
            found = scan_instance<ANY, true, false, false>(sc, I, lo, ld, raydiv_none(), time, tmin, tmax, hit_ref, order, st, sp);
        } else if (I.flags & RT_INST_RUNS) {
            /* packed runs (FInst::runs): a `box` is one record and one round trip instead of six leaf records one after
             * the other; the arithmetic is the flat kernels', divisions through the frame's shared reciprocals where
             * every lane of the wave that is here may (div_shared) */
            RayDiv q = raydiv_make(ld, tmin, origin_ok);
            if (I.flags & RT_INST_SPHERES) raydiv_spheres(q, ld);
            const bool one_box = I.runs == (uint64_t)(RT_RUN_BOX << RT_RUN_COUNT_BITS | 1); /* its record is I.head */
            if (q.fast) {
                if (one_box)
                    run_box<false, true, true>(I.head, I.ref_first, lo, ld, q, tmin, tmax, hit_ref);
                else
                    scan_runs<false, true, true>(sc, I, lo, ld, q, time, tmin, tmax, hit_ref);
            } else {
                if (one_box)
                    run_box<false, false, true>(I.head, I.ref_first, lo, ld, q, tmin, tmax, hit_ref);
                else
                    scan_runs<false, false, true>(sc, I, lo, ld, q, time, tmin, tmax, hit_ref);
            }
            found = ANY && hit_ref != before;
        } else {
            found = leaf_refs<true, false, ANY>(sc, I.ref_first, I.ref_first + I.n_ref, lo, ld, time, tmin, tmax, hit_ref, order);
        }
        if (hit_ref != before) hit_inst = owner, tmax_f = float_above(tmax);
        if (ANY && found) return true;
        node = sp > sp0 ? st.get(--sp) : RT_BVH_DONE;
    }
    return false;
}

template <bool ANY, bool TREES, bool WEXIT, bool TOP, bool GUARD>
__device__ __forceinline__ bool trace_fast(const DScene& sc, const FSub sub, V3 o, V3 d, Real time, Real tmin, Real& tmax,
                                           int& hit_ref, int& hit_inst, const Stack st, const int sp0) {
    /* a sub-scene with a top tree, in a kernel that carries the walk (RT_TRAV_TOP; the others scan its instances in
     * order like any sub-scene's): only the untransformed instance is scanned here, and only when that has no tree */
    const bool top = TOP && sub.top_root >= 0;
    const int inst_first = top ? sub.world_inst : sub.inst_first;
    const int n_inst = top ? (sub.world_linear ? 1 : 0) : sub.n_inst;
    /* with a handful of instances some lane of the wave enters every one of them, so the boxes
     * would prune nothing at wave level: skip them (and the three divisions of 1/d) */
    const bool use_boxes = n_inst > RT_FAST_NO_BOX_MAX;
    V3 inv = mk(0, 0, 0);
    if (use_boxes) inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    hit_ref = -1;
    hit_inst = -1;
    int order = -1;
#ifndef RTR_NO_SHARED_DIV
    const bool origin_ok = (RTR_TREE_SHARED || !TREES) && sc.shared_div != 0 && raydiv_origin_ok(o);
#endif
    for (int ii = inst_first; ii < inst_first + n_inst; ++ii) {
        RT_REGION(WEXIT ? RG_SH_SETUP : RG_SETUP);
        const FInst I = ld_const(sc.finst, ii);
        V3 lo = o, ld = d;
        const int n_xf = I.n_xf;
        if (n_xf) {
            if (use_boxes) {
                Real tn;
                if (!box_enter(I.bmin, I.bmax, o, inv, tmin, tmax, tn)) continue;
            }
            wrapper_enter(I.xf_type[0], I.xf_f[0], lo, ld);
            if (n_xf > 1) wrapper_enter(I.xf_type[1], I.xf_f[1], lo, ld);
            for (int k = RT_INST_XF_INLINE; k < n_xf; ++k) {
                const FXf x = ld_const(sc.fxf, I.xf_first + k);
                wrapper_enter(x.type, x.f, lo, ld);
            }
        }
        /* what the divisions of this frame share (see div_shared); made per frame rather than carried from the world
         * frame: fifteen instructions against six registers held across the instance loop */
        /* (kernels with box trees are bound by registers and the latency of their node fetches, not by instruction
         * count: the shared reciprocals cost them more in spills than the shorter divisions return -- measured on
         * scenes 9 / 22: 1 500 -> 1 350 and 770 -> 630 Msamples/s -- so they keep the plain divisions) */
        RayDiv q = raydiv_none();
#ifndef RTR_NO_SHARED_DIV /* experiments: the plain divisions, none of the shared-reciprocal code */
        if (RTR_TREE_SHARED || !TREES) {
            q = raydiv_make(ld, tmin, origin_ok);
            if (I.flags & RT_INST_SPHERES) raydiv_spheres(q, ld);
        }
#endif
        /* references are numbered instance by instance: a hit of this instance is one with ref >= ref_first */
        bool found;
        if (q.fast)
            found = scan_instance<ANY, TREES, WEXIT, true, GUARD>(sc, I, lo, ld, q, time, tmin, tmax, hit_ref, order, st, sp0);
        else
            found = scan_instance<ANY, TREES, WEXIT, false, GUARD>(sc, I, lo, ld, q, time, tmin, tmax, hit_ref, order, st, sp0);
        RT_REGION(WEXIT ? RG_SH_SETUP : RG_SETUP);
        if (hit_ref >= I.ref_first) hit_inst = ii;
        if (ANY && found) return true;
    }
    if (top) {
        RT_REGION(WEXIT ? RG_SH_TREE : RG_TREE);
        if (trace_top<ANY>(sc, sub, o, d, time, tmin, tmax, hit_ref, hit_inst, order, st, sp0) && ANY) return true;
    }
    return hit_ref >= 0;
}

/* Build the reference's hit_record for (reference, instance, t): the primitive's own hit()
 * tail in the instance frame, then the epilogues of the wrappers above it, innermost first.  Generic form:
 * any chain length, wrapper list through FRef / fexit / nodes (one dependent load per hop). */
template <bool UV>
__device__ __forceinline__ void fast_finish_long(const DScene& sc, V3 o, V3 d, Real time, Real t, int ref, int inst, Hit& rec) {
    const FInst I = ld_const(sc.finst, inst);
    const FRef R = ld_const(sc.fref, ref);
    V3 lo = o, ld = d;
    for (int k = 0; k < I.n_xf; ++k) {
        const FXf x = ld_const(sc.fxf, I.xf_first + k);
        wrapper_enter(x.type, x.f, lo, ld);
    }
    const rtr_node n = ld_const(sc.fprim, ref);
    const int type = n.type;
    if (type >= RTR_NODE_XY_RECT) {
        Real oa, da, ob, db;
        if (type == RTR_NODE_XY_RECT) {
            oa = lo.x, da = ld.x, ob = lo.y, db = ld.y;
        } else if (type == RTR_NODE_XZ_RECT) {
            oa = lo.x, da = ld.x, ob = lo.z, db = ld.z;
        } else {
            oa = lo.y, da = ld.y, ob = lo.z, db = ld.z;
        }
        rect_fill(n, type, lo, ld, t, oa + t * da, ob + t * db, UV, rec);
    } else {
        V3 center;
        Real radius;
        sphere_geom(n, type, time, center, radius);
        sphere_fill(n, type, center, radius, lo, ld, t, UV, rec);
    }
    /* wrapper epilogues: the k-th translate/rotate_y from the inside saw the ray after the
     * instance's first (n_xf - k) transform ops */
    int level = I.n_xf;
    for (int e = 0; e < R.n_exit; ++e) {
        const rtr_node w = ld_const(sc.nodes, as_const(sc.fexit)[R.exit_first + e]);
        V3 wd = d;
        if (w.type != RTR_NODE_FLIP_FACE) {
            V3 wo = o;
            for (int k = 0; k < level; ++k) {
                const FXf x = ld_const(sc.fxf, I.xf_first + k);
                wrapper_enter(x.type, x.f, wo, wd);
            }
            --level;
        }
        wrapper_epilogue(w, wd, rec);
    }
}
/* translate::hit / rotate_y::hit / flip_face::hit after the child hit, from a transform op instead of a node record
 * (hittable.h:58-61,142-155,168: the arithmetic of wrapper_epilogue) */
RT_DEV void xf_epilogue(int type, const double* f, V3 wd, Hit& rec) {
    if (type == RTR_NODE_TRANSLATE) {
        rec.p = add(rec.p, ld3(f));
        set_face_normal(rec, wd, rec.n);
    } else {
        const Real s = f[0], c = f[1];
        V3 p = rec.p, nn = rec.n;
        p.x = c * rec.p.x + s * rec.p.z;
        p.z = -s * rec.p.x + c * rec.p.z;
        nn.x = c * rec.n.x + s * rec.n.z;
        nn.z = -s * rec.n.x + c * rec.n.z;
        rec.p = p;
        set_face_normal(rec, wd, nn);
    }
}
/* The usual case -- a chain of at most two transforms -- needs the instance record (ops inline) and the primitive
 * record (geometry + wrapper code in f[9]): two independent loads, one round trip, where the generic form chases
 * FInst -> FXf, FRef -> fexit -> nodes per wrapper. */
template <bool UV>
RT_DEV void fast_finish(const DScene& sc, V3 o, V3 d, Real time, Real t, int ref, int inst, Hit& rec) {
    RT_REGION(RG_FINISH);
    const FInst I = ld_const(sc.finst, inst);
    const rtr_node n = ld_const(sc.fprim, ref);
    unsigned long long code = (unsigned long long)__double_as_longlong(n.f[9]);
    const int type = n.type;
    if (I.n_xf > RT_INST_XF_INLINE || type == RTR_NODE_MOVING_SPHERE || code == RT_EXIT_LONG) {
        fast_finish_long<UV>(sc, o, d, time, t, ref, inst, rec);
        return;
    }
    /* the ray as each wrapper passed it down: after op 0, after op 1 (only directions matter to the epilogues) */
    V3 lo = o, ld = d;
    V3 d1 = d;
    if (I.n_xf > 0) {
        wrapper_enter(I.xf_type[0], I.xf_f[0], lo, ld);
        d1 = ld;
        if (I.n_xf > 1) wrapper_enter(I.xf_type[1], I.xf_f[1], lo, ld);
    }
    if (type >= RTR_NODE_XY_RECT) {
        Real oa, da, ob, db;
        if (type == RTR_NODE_XY_RECT) {
            oa = lo.x, da = ld.x, ob = lo.y, db = ld.y;
        } else if (type == RTR_NODE_XZ_RECT) {
            oa = lo.x, da = ld.x, ob = lo.z, db = ld.z;
        } else {
            oa = lo.y, da = ld.y, ob = lo.z, db = ld.z;
        }
        rect_fill(n, type, lo, ld, t, oa + t * da, ob + t * db, UV, rec);
    } else {
        sphere_fill(n, type, ld3(n.f), n.f[3], lo, ld, t, UV, rec);
    }
    int level = I.n_xf;
    for (; code != 0; code >>= 2) {
        if ((code & 3) == 2) {
            rec.front = !rec.front;
        } else {
            --level; /* 1 or 0: selected by hand, an indexed access would put the record into scratch memory */
            const bool outer = level == 0;
            const double f[3] = {outer ? I.xf_f[0][0] : I.xf_f[1][0], outer ? I.xf_f[0][1] : I.xf_f[1][1],
                                 outer ? I.xf_f[0][2] : I.xf_f[1][2]};
            xf_epilogue(outer ? I.xf_type[0] : I.xf_type[1], f, outer ? d1 : ld, rec);
        }
    }
}

/* ---- the two ray casts of the integrators, for either traversal ------------------------------ */
/* TRAV: 0 = reference-order traversal, 1 = the same with constant_medium support, 2 = compiled scene */
#define RT_TRAV_EXACT 0
#define RT_TRAV_MEDIA 1
#define RT_TRAV_FAST 2
#define RT_TRAV_PROGRAM 3 /* scenes with media: the step program (struct FStep) */
#define RT_TRAV_FLAT 4    /* compiled scene without box trees and without tie-capable references */
#define RT_TRAV_TOP 5     /* RT_TRAV_FAST on a scene whose sub-scene 0 has a top tree (FSub::top_root): per-lane instance walk */
#define RT_TRAV_PROGRAM_EXT 6 /* template value only: RT_TRAV_PROGRAM whose program may hold guarded steps (FStep kind 3) and media
                                under wrappers (FStep::n_xf); compiled into the plain program kernels that code cost scenes 9 / 22
                                7 % / 4 %, so the megakernel has both and picks per scene; the other kernels take this one */
constexpr bool rt_is_program(int trav) { return trav == RT_TRAV_PROGRAM || trav == RT_TRAV_PROGRAM_EXT; }
#define RT_TRAV_FLAT_GUARD 7 /* template value only: RT_TRAV_FLAT of a scene with guarded references (hollow spheres): type runs with the
                                guarded run and the guard test in the generic loop; its own variant for the same reason */
constexpr bool rt_is_flat(int trav) { return trav == RT_TRAV_FLAT || trav == RT_TRAV_FLAT_GUARD; }

/* Run the ray-cast program over [tmin, tmax].  Returns whether anything was hit; then `tmax` is
 * the hit's t and either `med` >= 0 (the step of the medium that scattered the ray) or
 * (`ref`, `inst`) name the surface.  One trace_fast call site serves the geometry steps and both
 * boundary casts of a medium (constant_medium.h:62-66). */
template <bool ANY, bool EXT = true>
__device__ __forceinline__ bool run_program(const DScene& sc, V3 o, V3 d, Real time, Real tmin, Real& tmax, int& ref,
                                            int& inst, int& med, uint32_t& rng, const Stack st) {
    ref = -1, inst = -1, med = -1;
    bool any = false;
    const int n_steps = sc.n_fstep;

    for (int k = 0; k < n_steps; ++k) {
        /* a shadow ray that is blocked may stop once no medium is left to draw */
        if (ANY && any && k >= sc.fstep_tail) break;
        RT_REGION(RG_MEDIA);
        const FStep step = ld_const(sc.fstep, k);
        /* constant_medium.h:68-103 once both boundary hits exist: clip to the ray's interval, draw, scatter or not */
        V3 md = d; /* the ray's direction in the medium's own frame */
        if (EXT && step.n_xf > 0) {
            V3 mo = o;
            for (int k = 0; k < step.n_xf; ++k) {
                const FXf x = ld_const(sc.fxf, step.xf_first + k);
                wrapper_enter(x.type, x.f, mo, md);
            }
        }
        auto medium_between = [&](Real t1, Real t2) {
            if (t1 < tmin) t1 = tmin;
            if (t2 > tmax) t2 = tmax;
            if (!(t1 >= t2)) {
                if (t1 < 0) t1 = 0;
                const Real ray_length = len(md);
                const Real distance_inside_boundary = (t2 - t1) * ray_length;
                const Real hit_distance = step.neg_inv_density * log(rng_next(rng));
                if (!(hit_distance > distance_inside_boundary)) {
                    tmax = t1 + hit_distance / ray_length;
                    med = k, any = true;
                }
            }
        };
        if (step.kind == 2) {
            /* the boundary is ONE plain sphere in the world frame (the fog ball and the mist around everything
             * in final_scene): both boundary->hit calls (constant_medium.h:62-66) side by side, so the
             * discriminant, its root and the near root of sphere::hit are computed once -- the same operations on
             * the same operands, shared by the optimiser, instead of two rounds through the generic cast */
            const rtr_node n = ld_const(sc.fprim, step.pad);
            const V3 center = ld3(n.f);
            const Real radius = n.f[3];
            Real t1, t2;
            /* both boundary casts divide by |d|^2 of the world frame, up to four times (t_min = -inf here: tiny
             * numerators take the plain division) */
            RayDiv mq = raydiv_none();
            if (RTR_MEDIUM_SHARED && sc.shared_div) {
                mq.guard = true;
                mq.a = len2(d);
                mq.ra = rcp_refined(mq.a);
                mq.fast = raydiv_origin_ok(o) && __all(rcp_safe(mq.a));
            }
            if (mq.fast) {
                if (sphere_hit_t<true>(center, radius, o, d, mq, -RT_INF, RT_INF, t1) &&
                    sphere_hit_t<true>(center, radius, o, d, mq, t1 + 0.0001, RT_INF, t2))
                    medium_between(t1, t2);
            } else if (sphere_hit_t(center, radius, o, d, -RT_INF, RT_INF, t1) &&
                       sphere_hit_t(center, radius, o, d, t1 + 0.0001, RT_INF, t2)) {
                medium_between(t1, t2);
            }
            continue;
        }
        const FSub sub = ld_const(sc.fsub, step.sub);
        const bool medium = step.kind == 1;
        bool enter = true;
        if (EXT && step.kind == 3) { /* the bvh_node::hit calls above the primitive, with the closest t so far */
            enter = guard_pass(sc, step.pad, step.mat, o, d, tmin, tmax);
            if (!__any(enter)) continue;
        }
        Real lo = medium ? -RT_INF : tmin;
        Real t1 = 0;
        const int passes = medium ? 2 : 1;
#pragma nounroll
        for (int pass = 0; pass < passes; ++pass) {
            Real t = medium ? RT_INF : (!EXT || enter ? tmax : -RT_INF); /* (an empty interval: the lane's ray is not let in) */
            int r, i;
            /* ANY = a shadow ray: closest hit all the same (a medium behind needs t_max), but its finite interval
             * lets whole waves leave the rectangle tests early (rect_hit_axes) */
            const bool h = trace_fast<false, true, ANY>(sc, sub, o, d, time, lo, t, r, i, st, 0);
            if (!medium) {
                if (h) tmax = t, ref = r, inst = i, med = -1, any = true;
            } else if (!h) {
                break;
            } else if (pass == 0) {
                t1 = t;
                lo = t + 0.0001;
            } else {
                medium_between(t1, t);
            }
        }
    }
    return any;
}

/* The hit record of a medium step that scattered the ray at t (constant_medium.h:95-101), then the epilogues of the
 * wrappers the medium sits under, innermost first (fast_finish_long's loop: the k-th transform from the inside saw the ray
 * after the chain's first n_xf - k ops). */
template <bool EXT = true>
__device__ __forceinline__ void medium_finish(const DScene& sc, int med, V3 o, V3 d, Real t, Hit& rec) {
    if (!EXT) {
        rec.t = t;
        rec.p = add(o, scl(t, d));
        rec.n = mk(1, 0, 0);
        rec.front = true;
        rec.mat = as_const(sc.fstep)[med].mat;
        return;
    }
    const FStep step = ld_const(sc.fstep, med);
    V3 lo = o, ld = d;
    for (int k = 0; k < step.n_xf; ++k) {
        const FXf x = ld_const(sc.fxf, step.xf_first + k);
        wrapper_enter(x.type, x.f, lo, ld);
    }
    rec.t = t;
    rec.p = add(lo, scl(t, ld));
    rec.n = mk(1, 0, 0);
    rec.front = true;
    rec.mat = step.mat;
    int level = step.n_xf;
    for (int e = 0; e < step.n_exit; ++e) {
        const rtr_node w = ld_const(sc.nodes, as_const(sc.fexit)[step.exit_first + e]);
        V3 wd = d;
        if (w.type != RTR_NODE_FLIP_FACE) {
            V3 wo = o;
            for (int k = 0; k < level; ++k) {
                const FXf x = ld_const(sc.fxf, step.xf_first + k);
                wrapper_enter(x.type, x.f, wo, wd);
            }
            --level;
        }
        wrapper_epilogue(w, wd, rec);
    }
}

/* UV_POSSIBLE = false: the caller's material set has no texture that reads (u,v) (lean / quadlit kernels) */
template <int TRAV, bool UV_POSSIBLE = true>
__device__ __forceinline__ bool cast_closest(const DScene& sc, V3 o, V3 d, Real time, Hit& rec, uint32_t& rng,
                                             const Stack st, Real tmin = 0.001, Real tmax = RT_INF) {
    if (TRAV == RT_TRAV_FAST || rt_is_flat(TRAV) || TRAV == RT_TRAV_TOP) {
        int ref, inst;
        if (!trace_fast<false, !rt_is_flat(TRAV), false, TRAV == RT_TRAV_TOP, TRAV == RT_TRAV_FLAT_GUARD>(sc, sub_scene0(sc), o, d, time, tmin, tmax, ref, inst, st, 0))
            return false;
        if (UV_POSSIBLE && sc.needs_uv)
            fast_finish<true>(sc, o, d, time, tmax, ref, inst, rec);
        else
            fast_finish<false>(sc, o, d, time, tmax, ref, inst, rec);
        return true;
    }
    if (rt_is_program(TRAV)) {
        int ref, inst, med;
        if (!run_program<false, TRAV == RT_TRAV_PROGRAM_EXT>(sc, o, d, time, tmin, tmax, ref, inst, med, rng, st)) return false;
        if (med >= 0) {
            medium_finish<TRAV == RT_TRAV_PROGRAM_EXT>(sc, med, o, d, tmax, rec);
        } else if (UV_POSSIBLE && sc.needs_uv) {
            fast_finish<true>(sc, o, d, time, tmax, ref, inst, rec);
        } else {
            fast_finish<false>(sc, o, d, time, tmax, ref, inst, rec);
        }
        return true;
    }
    return traverse<true, TRAV == RT_TRAV_MEDIA>(sc, sc.root, o, d, time, tmin, tmax, rec, rng, st, 0);
}
template <int TRAV>
__device__ __forceinline__ bool cast_shadow(const DScene& sc, V3 o, V3 d, Real tmax, uint32_t& rng, const Stack st) {
    if (TRAV == RT_TRAV_FAST || rt_is_flat(TRAV) || TRAV == RT_TRAV_TOP) {
        int ref, inst;
        return trace_fast<true, !rt_is_flat(TRAV), true, TRAV == RT_TRAV_TOP, TRAV == RT_TRAV_FLAT_GUARD>(sc, sub_scene0(sc), o, d, 0.0, 0.001, tmax, ref, inst, st, 0);
    }
    if (rt_is_program(TRAV)) { /* the media behind a blocker still draw */
        int ref, inst, med;
        return run_program<true, TRAV == RT_TRAV_PROGRAM_EXT>(sc, o, d, 0.0, 0.001, tmax, ref, inst, med, rng, st);
    }
    Hit dummy;
    return traverse<false, TRAV == RT_TRAV_MEDIA>(sc, sc.root, o, d, 0.0, 0.001, tmax, dummy, rng, st, 0);
}
/* ---- materials/perlin.h:21-111 -------------------------------------------------------------- */
RT_DEV Real perlin_noise(const rtr_perlin* pnp, V3 p) {
    const RT_CONST_AS int32_t* perm_x = as_const(&pnp->perm_x[0]);
    const RT_CONST_AS int32_t* perm_y = as_const(&pnp->perm_y[0]);
    const RT_CONST_AS int32_t* perm_z = as_const(&pnp->perm_z[0]);
    const RT_CONST_AS double* ranvec = as_const(&pnp->ranvec[0][0]);
    const Real fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    const Real u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const int i = (int)fx, j = (int)fy, k = (int)fz;
    const Real uu = u * u * (3 - 2 * u);
    const Real vv = v * v * (3 - 2 * v);
    const Real ww = w * w * (3 - 2 * w);
    Real accum = 0.0;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int ix = perm_x[(i + a) & 255] ^ perm_y[(j + b) & 255] ^ perm_z[(k + c) & 255];
                V3 cv = mk(ranvec[3 * ix], ranvec[3 * ix + 1], ranvec[3 * ix + 2]);
                V3 weight_v = mk(u - a, v - b, w - c);
                accum += (a * uu + (1 - a) * (1 - uu)) * (b * vv + (1 - b) * (1 - vv)) *
                         (c * ww + (1 - c) * (1 - ww)) * dot(cv, weight_v);
            }
    return accum;
}
RT_DEV Real perlin_turb(const rtr_perlin* pn, V3 p) { /* perlin.h:41-54 */
    Real accum = 0.0;
    V3 temp_p = p;
    Real weight = 1.0;
    for (int i = 0; i < 7; i++) {
        accum += weight * perlin_noise(pn, temp_p);
        weight *= 0.5;
        temp_p = mk(temp_p.x * 2, temp_p.y * 2, temp_p.z * 2);
    }
    return __builtin_fabs(accum);
}

/* ---- materials/texture.h:11-162 ----------------------------------------------------------------- */
/* checker / noise / image textures */
__device__ __forceinline__ V3 tex_value_slow(const DScene& sc, int ix, Real u, Real v, V3 p) {
    /* checker textures nest (texture.h:60-66); unrolled to a bounded loop instead of recursion */
    for (int guard = 0; guard < 8; ++guard) {
        const rtr_texture t = ld_const(sc.textures, ix);
        const int type = t.type;
        if (type == RTR_TEX_SOLID) return ld3(t.f);
        if (type == RTR_TEX_CHECKER) { /* texture.h:68-75 */
            Real sines = sin(10 * p.x) * sin(10 * p.y) * sin(10 * p.z);
            ix = sines < 0 ? t.b : t.a;
            continue;
        }
        if (type == RTR_TEX_NOISE) { /* texture.h:155-158 */
            Real x = 1 + sin(t.f[0] * p.z + 10 * perlin_turb(sc.perlin + t.a, p));
            return scl(x, mk(0.5, 0.5, 0.5));
        }
        /* image_texture, texture.h:115-139 */
        if (t.a < 0) return mk(0, 1, 1);
        const rtr_image im = ld_const(sc.images, t.a);
        u = clampd(u, 0.0, 1.0);
        v = 1.0 - clampd(v, 0.0, 1.0);
        int i = (int)(u * im.width);
        int j = (int)(v * im.height);
        if (i >= im.width) i = im.width - 1;
        if (j >= im.height) j = im.height - 1;
        const Real color_scale = 1.0 / 255.0;
        const RT_CONST_AS uint8_t* px = as_const(sc.image_bytes + im.offset + (size_t)j * 3 * im.width + (size_t)i * 3);
        return mk(color_scale * px[0], color_scale * px[1], color_scale * px[2]);
    }
    return mk(0, 0, 0);
}
/* Material-set specialisation.  Upload knows which material and texture classes a scene uses;
 * kernels are instantiated for the common diffuse-only set (lambertian + diffuse_light with
 * solid_color textures: the Cornell boxes) without the code of the other classes, which keeps
 * Cook-Torrance / pow() / perlin out of their register budget.  RT_MS_FULL handles everything. */
#define RT_MS_LEAN 0
#define RT_MS_FULL 1
#define RT_MS_QUADLIT 2 /* every material, but only QuadLights and no texture that reads (u,v): keeps the delta /
                           environment light code (binary searches, sin / cos / acos / atan2) and the (u,v)
                           reconstruction out of e.g. scene 23's kernel */

template <int MS = RT_MS_FULL>
RT_DEV V3 tex_value(const DScene& sc, int ix, Real u, Real v, V3 p) {
    const rtr_texture t = ld_const(sc.textures, ix);
    if (MS == RT_MS_LEAN || t.type == RTR_TEX_SOLID) return ld3(t.f); /* texture.h:46-48 */
    return tex_value_slow(sc, ix, u, v, p);
}
RT_DEV Real tex_scalar(const DScene& sc, int ix, Real u, Real v, V3 p) { return tex_value(sc, ix, u, v, p).x; }
RT_DEV V3 tex_normal(const DScene& sc, int ix, Real u, Real v, V3 p) { /* texture.h:19-22 */
    V3 c = tex_value(sc, ix, u, v, p);
    return unit(sub(scl(2.0, c), mk(1, 1, 1)));
}

/* ---- core/onb.h:24-37 ------------------------------------------------------------------------------- */
struct Onb {
    V3 u, v, w;
};
RT_DEV Onb onb_from_w(V3 n) {
    Onb b;
    b.w = unit(n);
    V3 a = (__builtin_fabs(b.w.x) > 0.9) ? mk(0, 1, 0) : mk(1, 0, 0);
    b.v = unit(cross(b.w, a));
    b.u = cross(b.w, b.v);
    return b;
}
RT_DEV V3 onb_local(const Onb& b, V3 a) { return add(add(scl(a.x, b.u), scl(a.y, b.v)), scl(a.z, b.w)); }

/* ---- materials/material.h ------------------------------------------------------------------------------ */
struct BSDFSample { /* material.h:13-20 */
    V3 wi, f;
    Real pdf;
    bool is_specular;
    bool is_transmission; /* write-only in the reference; kept for the unit vectors */
};

RT_DEV Real distribution_ggx(V3 N, V3 H, Real roughness) { /* material.h:398-409 */
    Real a = roughness * roughness;
    Real a2 = a * a;
    Real NdotH = maxd(dot(N, H), 0.0);
    Real NdotH2 = NdotH * NdotH;
    Real denom = (NdotH2 * (a2 - 1.0) + 1.0);
    denom = RT_PI * denom * denom;
    return a2 / denom;
}
RT_DEV Real geometry_schlick_ggx(Real NdotV, Real roughness) { /* material.h:411-419 */
    Real k = (roughness * roughness) / 2.0;
    Real denom = NdotV * (1.0 - k) + k;
    return NdotV / denom;
}
RT_DEV Real geometry_smith(V3 N, V3 V, V3 L, Real roughness) { /* material.h:421-428 */
    Real NdotV = maxd(dot(N, V), 0.0);
    Real NdotL = maxd(dot(N, L), 0.0);
    Real ggx2 = geometry_schlick_ggx(NdotV, roughness);
    Real ggx1 = geometry_schlick_ggx(NdotL, roughness);
    return ggx1 * ggx2;
}
/* pow(x, 5.0) of the reference (material.h:202,431; glibc: correctly rounded in nearly all cases) as
 * x^5 in double-double arithmetic: the rounded sum is the correctly rounded x^5 up to rare half-way
 * cases -- closer to glibc than OCML's pow, at a tenth of its instructions. */
RT_DEV Real pow5(Real x) {
    const Real h2 = x * x, l2 = __builtin_fma(x, x, -h2);
    const Real h4 = h2 * h2, l4 = __builtin_fma(h2, h2, -h4) + 2.0 * h2 * l2;
    const Real h5 = h4 * x, l5 = __builtin_fma(h4, x, -h5) + l4 * x;
    return h5 + l5;
}
RT_DEV V3 fresnel_schlick(Real cosTheta, V3 F0) { /* material.h:430-432 */
    return add(F0, scl(pow5(1.0 - cosTheta), sub(mk(1, 1, 1), F0)));
}
template <class M>
RT_DEV V3 pbr_normal(const DScene& sc, const M& m, const Hit& rec) { /* material.h:247-261 */
    V3 N = rec.n;
    if (m.tex[3] >= 0) {
        V3 ax0;
        if (__builtin_fabs(N.y) > 0.999)
            ax0 = mk(1, 0, 0);
        else
            ax0 = unit(cross(N, mk(0, 1, 0)));
        V3 ax1 = cross(N, ax0);
        V3 ln = tex_normal(sc, m.tex[3], rec.u, rec.v, rec.p);
        N = unit(add(add(scl(ln.x, ax0), scl(ln.y, ax1)), scl(ln.z, N)));
    }
    return N;
}
/* What a hit's material needs from its textures, evaluated ONCE per hit.  The reference re-evaluates
 * the same pure texture functions inside every eval() / pdf() / sample() call of a bounce (up to eleven
 * lookups for a PBRMaterial); the values are the same, so are the results. */
struct MatCtx {
    int type;
    Real f0, f1, f2, f3; /* the record's own parameters (metal: albedo + fuzz, dielectric: ir) */
    V3 albedo;           /* lambertian / diffuse_light / isotropic: value of tex[0]; PBR: base colour */
    V3 N;                /* PBR: shading normal (material.h:247-261) */
    Real rough, metal;   /* PBR: roughness clamped to [0.01, 1] (:264,327,368), metallic */
};
__device__ __forceinline__ Real pbr_pdf(const MatCtx& c, V3 wo, V3 wi) {
    /* material.h:305-340 */
    const V3 N = c.N;
    if (dot(N, wi) <= 0) return 0;
    const Real rough = c.rough;
    Real pdf_diff = dot(N, wi) / RT_PI;
    V3 H = unit(add(wo, wi));
    Real D = distribution_ggx(N, H, rough);
    Real NdotH = maxd(dot(N, H), 0.0);
    Real HdotV = maxd(dot(H, wo), 0.0);
    Real pdf_spec = (D * NdotH) / (4.0 * HdotV + 0.0001);
    return 0.5 * pdf_diff + 0.5 * pdf_spec;
}
__device__ __forceinline__ V3 pbr_eval(const MatCtx& c, V3 wo, V3 wi) {
    /* material.h:342-396 */
    const V3 N = c.N;
    Real NdotL = dot(N, wi);
    Real NdotV = dot(N, wo);
    if (NdotL <= 0 || NdotV <= 0) return mk(0, 0, 0);
    const Real rough = c.rough, metal = c.metal;
    const V3 base_color = c.albedo;
    V3 H = unit(add(wo, wi));
    V3 F0 = mk(0.04, 0.04, 0.04);
    V3 metal_vec = mk(metal, metal, metal);
    F0 = add(mul(sub(mk(1.0, 1.0, 1.0), metal_vec), F0), mul(metal_vec, base_color));
    V3 F = fresnel_schlick(maxd(dot(H, wo), 0.0), F0);
    Real D = distribution_ggx(N, H, rough);
    Real G = geometry_smith(N, wo, wi, rough);
    V3 numerator = scl(D * G, F);
    Real denominator = 4.0 * NdotV * NdotL + 0.0001;
    V3 specular = divs(numerator, denominator);
    V3 kD = sub(mk(1.0, 1.0, 1.0), F);
    kD = scl(1.0 - metal, kD);
    V3 diffuse = divs(mul(kD, base_color), RT_PI);
    return add(diffuse, specular);
}
/* PBRMaterial::eval and ::pdf of the same pair of directions (the light sample of sample_lights_mis, the BSDF sample of
 * PBRMaterial::sample: material.h:296-298, mis_path_integrator.h:215-224): the half vector, D, N.H and H.V are the same
 * expressions in both (material.h:317-331, :356-373) -- computed once here, each result as above */
__device__ __forceinline__ void pbr_eval_pdf(const MatCtx& c, V3 wo, V3 wi, V3& f, Real& pdf) {
    const V3 N = c.N;
    const Real NdotL = dot(N, wi);
    f = mk(0, 0, 0), pdf = 0;
    if (NdotL <= 0) return; /* material.h:312 and :349 */
    const Real rough = c.rough, metal = c.metal;
    const V3 H = unit(add(wo, wi));
    const Real D = distribution_ggx(N, H, rough);
    const Real NdotH = maxd(dot(N, H), 0.0);
    const Real HdotV = maxd(dot(H, wo), 0.0);
    const Real pdf_diff = NdotL / RT_PI;
    const Real pdf_spec = (D * NdotH) / (4.0 * HdotV + 0.0001);
    pdf = 0.5 * pdf_diff + 0.5 * pdf_spec;
    const Real NdotV = dot(N, wo);
    if (NdotV <= 0) return; /* material.h:349 */
    const V3 base_color = c.albedo;
    V3 F0 = mk(0.04, 0.04, 0.04);
    const V3 metal_vec = mk(metal, metal, metal);
    F0 = add(mul(sub(mk(1.0, 1.0, 1.0), metal_vec), F0), mul(metal_vec, base_color));
    const V3 F = fresnel_schlick(HdotV, F0);
    const Real G = geometry_smith(N, wo, wi, rough);
    const V3 numerator = scl(D * G, F);
    const Real denominator = 4.0 * NdotV * NdotL + 0.0001;
    const V3 specular = divs(numerator, denominator);
    V3 kD = sub(mk(1.0, 1.0, 1.0), F);
    kD = scl(1.0 - metal, kD);
    const V3 diffuse = divs(mul(kD, base_color), RT_PI);
    f = add(diffuse, specular);
}
RT_DEV Real reflectance(Real cosine, Real ref_idx) { /* material.h:199-203 */
    Real r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    return r0 + (1 - r0) * pow5(1 - cosine);
}

template <int MS = RT_MS_FULL>
RT_DEV MatCtx mat_prepare(const DScene& sc, const Hit& rec) {
    RT_REGION(RG_MATPREP);
    const FMat m = ld_const(sc.fmat, rec.mat); /* constant address space: never a flat load */
    MatCtx c;
    c.type = m.type;
    c.f0 = m.f[0], c.f1 = m.f[1], c.f2 = m.f[2], c.f3 = m.f[3];
    c.albedo = mk(0, 0, 0), c.N = rec.n, c.rough = 0, c.metal = 0;
    if (MS == RT_MS_LEAN || m.solid) { /* every texture the material reads is a constant: its values came with the record */
        c.albedo = ld3(m.albedo);
        c.rough = m.rough, c.metal = m.metal;
        return c;
    }
    if (c.type == RTR_MAT_LAMBERTIAN || c.type == RTR_MAT_DIFFUSE_LIGHT) {
        c.albedo = tex_value<MS>(sc, m.tex[0], rec.u, rec.v, rec.p);
    } else if (MS != RT_MS_LEAN && c.type == RTR_MAT_ISOTROPIC) {
        c.albedo = tex_value(sc, m.tex[0], rec.u, rec.v, rec.p);
    } else if (MS != RT_MS_LEAN && c.type == RTR_MAT_PBR) {
        c.N = pbr_normal(sc, m, rec);
        c.rough = clampd(tex_scalar(sc, m.tex[1], rec.u, rec.v, rec.p), 0.01, 1.0);
        c.metal = tex_scalar(sc, m.tex[2], rec.u, rec.v, rec.p);
        c.albedo = tex_value(sc, m.tex[0], rec.u, rec.v, rec.p);
    }
    return c;
}

/* material::emitted(rec, wo): material.h:32-34, :222-227 (front face only) */
RT_DEV V3 mat_emitted(const MatCtx& c, const Hit& rec) {
    if (c.type == RTR_MAT_DIFFUSE_LIGHT && rec.front) return c.albedo;
    return mk(0, 0, 0);
}
/* material::emitted(u, v, p): material.h:27-29, :218-220 (two-sided) */
RT_DEV V3 mat_emitted_legacy(const MatCtx& c) {
    if (c.type == RTR_MAT_DIFFUSE_LIGHT) return c.albedo;
    return mk(0, 0, 0);
}

/* PBRMaterial::sample (material.h:245-303) */
__device__ __forceinline__ bool pbr_sample(const MatCtx& c, V3 wo, BSDFSample& s, uint32_t& rng) {
    const V3 N = c.N;
    const Real rough = c.rough;
    /* Both branches of material.h:263-302 build the same basis around N, draw r1 then r2 and take cos / sin of
     * 2 pi r1 (the cosine branch inside random_cosine_direction, vec3.h:261-269): done once here for the whole wave,
     * which nearly always holds lanes of both branches; each lane then finishes its own branch with its own operands */
    const bool ggx = rng_next(rng) < 0.5;
    const Onb uvw = onb_from_w(N);
    const Real r1 = rng_next(rng);
    const Real r2 = rng_next(rng);
    const Real phi = 2.0 * RT_PI * r1;
    Real sphi, cphi; /* one argument reduction for both (see random_cosine_direction) */
    sincos(phi, &sphi, &cphi);
    if (ggx) {
        Real a = rough * rough;
        Real cos_theta = __builtin_sqrt((1.0 - r2) / (1.0 + (a * a - 1.0) * r2));
        Real sin_theta = __builtin_sqrt(1.0 - cos_theta * cos_theta);
        V3 H_local = mk(sin_theta * cphi, sin_theta * sphi, cos_theta);
        V3 H = onb_local(uvw, H_local);
        V3 L = reflect(neg(wo), H);
        if (dot(N, L) <= 0) return false;
        s.wi = L;
    } else {
        const Real z = __builtin_sqrt(1 - r2); /* vec3.h:261-269 */
        const Real x = cphi * __builtin_sqrt(r2);
        const Real y = sphi * __builtin_sqrt(r2);
        V3 L = onb_local(uvw, mk(x, y, z));
        if (dot(N, L) <= 0) L = N;
        s.wi = unit(L);
    }
    s.is_specular = false;
    pbr_eval_pdf(c, wo, s.wi, s.f, s.pdf);
    if (s.pdf < 1e-6) return false;
    return true;
}

template <int MS = RT_MS_FULL>
__device__ __forceinline__ bool mat_sample(const MatCtx& c, const Hit& rec, V3 wo, BSDFSample& s, uint32_t& rng) {
    const int type = c.type;
    if (type == RTR_MAT_LAMBERTIAN) { /* material.h:79-90 */
        V3 scatter_direction = add(rec.n, random_unit_vector(rng));
        if (near_zero(scatter_direction)) scatter_direction = rec.n;
        s.wi = unit(scatter_direction);
        s.pdf = dot(rec.n, s.wi) / RT_PI;
        s.f = divs(c.albedo, RT_PI);
        s.is_specular = false;
        return true;
    }
    if (MS == RT_MS_LEAN) return false; /* diffuse_light::sample (material.h:213-216) */
    if (type == RTR_MAT_METAL) { /* material.h:123-131 */
        V3 reflected = reflect(unit(neg(wo)), rec.n);
        s.wi = unit(add(reflected, scl(c.f3, random_in_unit_sphere(rng))));
        s.f = mk(c.f0, c.f1, c.f2);
        s.pdf = 1.0;
        s.is_specular = true;
        return dot(s.wi, rec.n) > 0;
    }
    if (type == RTR_MAT_DIELECTRIC) { /* material.h:152-174 */
        s.f = mk(1.0, 1.0, 1.0);
        s.is_specular = true;
        s.pdf = 1.0;
        Real ir = c.f0;
        Real refraction_ratio = rec.front ? (1.0 / ir) : ir;
        V3 unit_direction = neg(wo);
        Real cos_theta = __builtin_fmin(dot(neg(unit_direction), rec.n), 1.0);
        Real sin_theta = __builtin_sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = refraction_ratio * sin_theta > 1.0;
        if (cannot_refract || reflectance(cos_theta, refraction_ratio) > rng_next(rng)) {
            s.wi = reflect(unit_direction, rec.n);
            s.is_transmission = false;
        } else {
            s.wi = refract(unit_direction, rec.n, refraction_ratio);
            s.is_transmission = true;
        }
        return true;
    }
    if (type == RTR_MAT_PBR) return pbr_sample(c, wo, s, rng);
    return false; /* diffuse_light (material.h:213-216), isotropic (base class, :42-45) */
}

/* material::eval: base 0 (material.h:48-51), lambertian without hemisphere test (:98-101), PBR (:342) */
template <int MS = RT_MS_FULL>
RT_DEV V3 mat_eval(const MatCtx& c, V3 wo, V3 wi) {
    if (c.type == RTR_MAT_LAMBERTIAN) return divs(c.albedo, RT_PI);
    if (MS != RT_MS_LEAN && c.type == RTR_MAT_PBR) return pbr_eval(c, wo, wi);
    return mk(0, 0, 0);
}
/* material::pdf: base 0 (material.h:54-57), lambertian (:92-96), PBR (:305) */
template <int MS = RT_MS_FULL>
RT_DEV Real mat_pdf(const MatCtx& c, const Hit& rec, V3 wo, V3 wi) {
    if (c.type == RTR_MAT_LAMBERTIAN) {
        Real cosine = dot(rec.n, unit(wi));
        return cosine < 0 ? 0 : cosine / RT_PI;
    }
    if (MS != RT_MS_LEAN && c.type == RTR_MAT_PBR) return pbr_pdf(c, wo, wi);
    return 0.0;
}
/* material::eval and material::pdf of one pair of directions */
template <int MS = RT_MS_FULL>
RT_DEV void mat_eval_pdf(const MatCtx& c, const Hit& rec, V3 wo, V3 wi, V3& f, Real& pdf) {
    if (MS != RT_MS_LEAN && c.type == RTR_MAT_PBR) {
        pbr_eval_pdf(c, wo, wi, f, pdf);
        return;
    }
    f = mat_eval<MS>(c, wo, wi);
    pdf = mat_pdf<MS>(c, rec, wo, wi);
}
/* legacy material::scatter(r_in, rec, attenuation, scattered): new ray = (rec.p, dir, r_in.time) */
template <int MS = RT_MS_FULL>
__device__ __forceinline__ bool mat_scatter(const MatCtx& c, V3 rd, const Hit& rec, V3& attenuation, V3& out_dir,
                                            uint32_t& rng) {
    const int type = c.type;
    if (type == RTR_MAT_LAMBERTIAN) { /* material.h:103-112 */
        V3 scatter_direction = add(rec.n, random_unit_vector(rng));
        if (near_zero(scatter_direction)) scatter_direction = rec.n;
        out_dir = scatter_direction;
        attenuation = c.albedo;
        return true;
    }
    if (MS == RT_MS_LEAN) return false; /* diffuse_light::scatter (material.h:229-232) */
    if (type == RTR_MAT_METAL) { /* material.h:133-140 */
        V3 reflected = reflect(unit(rd), rec.n);
        out_dir = add(reflected, scl(c.f3, random_in_unit_sphere(rng)));
        attenuation = mk(c.f0, c.f1, c.f2);
        return dot(out_dir, rec.n) > 0;
    }
    if (type == RTR_MAT_DIELECTRIC) { /* material.h:176-193 */
        attenuation = mk(1.0, 1.0, 1.0);
        Real ir = c.f0;
        Real refraction_ratio = rec.front ? (1.0 / ir) : ir;
        V3 unit_direction = unit(rd);
        Real cos_theta = __builtin_fmin(dot(neg(unit_direction), rec.n), 1.0);
        Real sin_theta = __builtin_sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = refraction_ratio * sin_theta > 1.0;
        if (cannot_refract || reflectance(cos_theta, refraction_ratio) > rng_next(rng))
            out_dir = reflect(unit_direction, rec.n);
        else
            out_dir = refract(unit_direction, rec.n, refraction_ratio);
        return true;
    }
    if (type == RTR_MAT_ISOTROPIC) { /* geometry/constant_medium.h:19-24 */
        out_dir = random_in_unit_sphere(rng);
        attenuation = c.albedo;
        return true;
    }
    return false; /* diffuse_light (material.h:229-232), PBRMaterial (base class, :66-69) */
}

/* ---- lighting/quad_light.h:18-77 --------------------------------------------------------------------------- */
struct LightSample {
    V3 Li, wi;
    Real pdf, dist;
    bool is_delta;
};
/* ---- lighting/environmental_light.h: HDR map + Distribution2D (layout: rtr_hip.h, RTR_LIGHT_ENV_MAP) ---- */
struct EnvMap {
    int w, h;
    bool probe;
    const RT_CONST_AS float* texels;
    const RT_CONST_AS double* tables;
    /* Distribution1D of map row v (v == h: the marginal): func[n], cdf[n + 1], func_int */
    RT_DEV const RT_CONST_AS double* dist(int v, int& n) const {
        n = v < h ? w : h;
        return tables + (v < h ? (size_t)v * (2 * w + 2) : (size_t)h * (2 * w + 2));
    }
};
RT_DEV EnvMap env_map(const rtr_light& l, const uint8_t* blob) {
    EnvMap m;
    m.w = (int)l.f[0], m.h = (int)l.f[1], m.probe = l.f[2] != 0;
    m.texels = as_const(reinterpret_cast<const float*>(blob + (size_t)l.f[3]));
    m.tables = as_const(reinterpret_cast<const double*>(blob + (size_t)l.f[4]));
    return m;
}
RT_DEV Real dist1d_sample(const RT_CONST_AS double* d, int n, Real u, Real& pdf_out, int& offset) { /* :30-45 */
    const RT_CONST_AS double *func = d, *cdf = d + n;
    const Real func_int = d[2 * n + 1];
    int lo = 0, hi = n + 1; /* std::lower_bound over cdf[0 .. n] */
    while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        if (cdf[mid] < u)
            lo = mid + 1;
        else
            hi = mid;
    }
    offset = lo - 1 > 0 ? lo - 1 : 0;
    offset = offset < n - 1 ? offset : n - 1;
    Real du = u - cdf[offset];
    const Real span = cdf[offset + 1] - cdf[offset];
    if (span > 0) du /= span;
    pdf_out = (func_int > 0) ? func[offset] / func_int : 0;
    return (offset + du) / n;
}
RT_DEV Real dist1d_pdf(const RT_CONST_AS double* d, int n, int index) { /* :47-49 */
    const Real func_int = d[2 * n + 1];
    return (func_int > 0) ? d[index] / (func_int * n) : 0;
}
RT_DEV V3 env_pixel(const EnvMap& m, int i, int j) { /* :276-289 */
    if (i < 0) i += m.w;
    if (i >= m.w) i -= m.w;
    if (j < 0) j = 0;
    if (j >= m.h) j = m.h - 1;
    const RT_CONST_AS float* t = m.texels + 3 * ((size_t)j * m.w + i);
    return mk(t[0], t[1], t[2]);
}
/* direction -> map coordinates (:233-250, :299-315); returns theta of the polar axis of the mapping */
RT_DEV Real env_uv(const EnvMap& m, V3 unit_dir, Real& u, Real& v) {
    if (m.probe) {
        const Real d = __builtin_sqrt(unit_dir.x * unit_dir.x + unit_dir.y * unit_dir.y);
        const Real theta = acos(unit_dir.z);
        const Real r_coord = (d > 0) ? (1.0 / RT_PI) * theta / d : 0.0;
        u = (unit_dir.x * r_coord + 1.0) * 0.5;
        v = (unit_dir.y * r_coord + 1.0) * 0.5;
        v = 1.0 - v;
        return theta;
    }
    const Real theta = acos(unit_dir.y);
    const Real phi = atan2(-unit_dir.z, unit_dir.x) + RT_PI;
    u = phi / (2 * RT_PI);
    v = theta / RT_PI;
    return theta;
}
RT_DEV V3 env_Le(const EnvMap& m, V3 direction) { /* :226-274 */
    Real u, v;
    env_uv(m, unit(direction), u, v);
    const Real u_img = u * m.w - 0.5;
    const Real v_img = v * m.h - 0.5;
    const int i0 = (int)floor(u_img);
    const int j0 = (int)floor(v_img);
    const Real du = u_img - i0;
    const Real dv = v_img - j0;
    const V3 c00 = env_pixel(m, i0, j0), c10 = env_pixel(m, i0 + 1, j0);
    const V3 c01 = env_pixel(m, i0, j0 + 1), c11 = env_pixel(m, i0 + 1, j0 + 1);
    const V3 c0 = add(scl(1 - du, c00), scl(du, c10));
    const V3 c1 = add(scl(1 - du, c01), scl(du, c11));
    return add(scl(1 - dv, c0), scl(dv, c1));
}
RT_DEV LightSample env_sample(const EnvMap& m, Real ux, Real uy) { /* :182-224 */
    LightSample s;
    s.dist = RT_INF;
    s.is_delta = false;
    s.wi = mk(0, 0, 0);
    s.Li = mk(0, 0, 0);
    s.pdf = 0;
    Real pdfs[2];
    int v_idx, u_idx, n;
    const RT_CONST_AS double* marg = m.dist(m.h, n);
    const Real v = dist1d_sample(marg, n, uy, pdfs[1], v_idx);
    const RT_CONST_AS double* cond = m.dist(v_idx, n);
    const Real u = dist1d_sample(cond, n, ux, pdfs[0], u_idx);
    const Real map_pdf = pdfs[0] * pdfs[1];
    if (map_pdf == 0) return s;
    Real phi, theta;
    if (m.probe) {
        const Real uc = u * 2.0 - 1.0;
        const Real vc = (1.0 - v) * 2.0 - 1.0;
        const Real r = __builtin_sqrt(uc * uc + vc * vc);
        if (r > 1.0) return s;
        theta = RT_PI * r;
        phi = atan2(vc, uc);
        const Real sin_theta = sin(theta);
        s.wi = mk(sin_theta * cos(phi), sin_theta * sin(phi), cos(theta));
    } else {
        phi = u * 2 * RT_PI - RT_PI;
        theta = v * RT_PI;
        const Real sin_theta = sin(theta);
        const Real cos_theta = cos(theta);
        s.wi = mk(sin_theta * cos(phi), cos_theta, -sin_theta * sin(phi));
    }
    const Real sin_theta = sin(theta);
    if (sin_theta < 1e-6) return s;
    s.pdf = map_pdf * m.w * m.h / (2.0 * RT_PI * RT_PI * sin_theta);
    s.Li = env_Le(m, s.wi);
    return s;
}
RT_DEV Real env_pdf(const EnvMap& m, V3 direction) { /* :291-331 */
    Real u, v;
    const Real theta = env_uv(m, unit(direction), u, v);
    const Real sin_theta = sin(theta);
    if (sin_theta < 1e-6) return 0;
    const int u_idx = (int)clampd((int)(u * m.w), 0, m.w - 1);
    const int v_idx = (int)clampd((int)(v * m.h), 0, m.h - 1);
    int n;
    const RT_CONST_AS double* cond = m.dist(v_idx, n);
    const Real pu = dist1d_pdf(cond, n, u_idx);
    const RT_CONST_AS double* marg = m.dist(m.h, n);
    const Real map_pdf = pu * dist1d_pdf(marg, n, v_idx);
    return map_pdf * m.w * m.h / (2.0 * RT_PI * RT_PI * sin_theta);
}

/* only MS == RT_MS_FULL kernels know lights other than QuadLights (rtr_upload_scene decides) */
template <int MS = RT_MS_FULL>
RT_DEV LightSample light_sample(const rtr_light& l, V3 p, Real ux, Real uy, uint32_t& rng, const uint8_t* blob) {
    LightSample s;
    const int type = MS != RT_MS_FULL ? (int)RTR_LIGHT_QUAD : l.type;
    if (type == RTR_LIGHT_ENV_MAP) return env_sample(env_map(l, blob), ux, uy);
    if (type == RTR_LIGHT_ENV_UNIFORM) { /* lighting/environmental_light.h:182-192 (no map loaded) */
        s.dist = RT_INF;
        s.is_delta = false;
        s.wi = random_unit_vector(rng); /* drawn from the same generator, after u */
        s.pdf = 1.0 / (4.0 * RT_PI);
        s.Li = mk(1, 1, 1);
        return s;
    }
    if (type != RTR_LIGHT_QUAD) {
        s.is_delta = true;
        s.pdf = 1.0;
        if (type == RTR_LIGHT_DIRECTIONAL) { /* lighting/directional_light.h:13-21 */
            s.wi = neg(ld3(l.f));
            s.dist = RT_INF;
            s.Li = ld3(l.f + 3);
            return s;
        }
        /* lighting/point_light.h:12-22, spot_light.h:14-32 */
        V3 d = sub(ld3(l.f), p);
        Real dist2 = len2(d);
        s.dist = __builtin_sqrt(dist2);
        s.wi = divs(d, s.dist);
        if (type == RTR_LIGHT_POINT) {
            s.Li = divs(ld3(l.f + 3), dist2);
        } else {
            Real cos_theta = dot(neg(s.wi), ld3(l.f + 3));
            s.Li = cos_theta < l.f[9] ? mk(0, 0, 0) : divs(ld3(l.f + 6), dist2);
        }
        return s;
    }
    /* lighting/quad_light.h:18-48 */
    s.is_delta = false;
    V3 light_point = add(add(ld3(l.f), scl(ux, ld3(l.f + 3))), scl(uy, ld3(l.f + 6)));
    V3 d = sub(light_point, p);
    Real dist_sq = len2(d);
    s.dist = __builtin_sqrt(dist_sq);
    s.wi = divs(d, s.dist);
    Real cos_theta = dot(neg(s.wi), ld3(l.f + 12));
    if (cos_theta <= 0) {
        s.Li = mk(0, 0, 0);
        s.pdf = 0;
        return s;
    }
    s.Li = ld3(l.f + 9);
    s.pdf = dist_sq / (l.f[15] * cos_theta);
    return s;
}
template <int MS = RT_MS_FULL>
RT_DEV Real light_pdf(const rtr_light& l, V3 origin, V3 direction, const uint8_t* blob) {
    if (MS == RT_MS_FULL) {
        if (l.type == RTR_LIGHT_ENV_UNIFORM) return 1.0 / (4.0 * RT_PI); /* environmental_light.h:293-294 */
        if (l.type == RTR_LIGHT_ENV_MAP) return env_pdf(env_map(l, blob), direction);
        if (l.type != RTR_LIGHT_QUAD) return 0.0; /* Light::pdf (light.h:26-28): delta lights */
    }
    V3 Q = ld3(l.f), U = ld3(l.f + 3), Vv = ld3(l.f + 6), normal = ld3(l.f + 12);
    Real denom = dot(direction, normal);
    if (denom >= -1e-6) return 0;
    Real t = dot(sub(Q, origin), normal) / denom;
    if (t < 0.001 || t > RT_INF) return 0;
    V3 planar = sub(add(origin, scl(t, direction)), Q);
    Real alpha = dot(planar, U) / len2(U);
    Real beta = dot(planar, Vv) / len2(Vv);
    if (alpha < 0 || alpha > 1 || beta < 0 || beta > 1) return 0;
    Real dist_sq = t * t * len2(direction);
    Real cos_theta = -denom / len(direction);
    return dist_sq / (l.f[15] * cos_theta);
}

/* ---- renderer/camera.h:32-40 --------------------------------------------------------------------------------- */
RT_DEV void camera_get_ray(const rtr_camera& c, Real s, Real t, uint32_t& rng, V3& o, V3& d, Real& tm) {
    V3 rd = scl(c.lens_radius, random_in_unit_disk(rng));
    V3 offset = add(scl(rd.x, ld3(c.u)), scl(rd.y, ld3(c.v)));
    V3 origin = ld3(c.origin);
    d = sub(sub(add(add(ld3(c.lower_left_corner), scl(s, ld3(c.horizontal))), scl(t, ld3(c.vertical))), origin),
            offset);
    o = add(origin, offset);
    tm = rng_range(rng, c.time0, c.time1);
}

/* ---- renderer/mis_path_integrator.h helpers -------------------------------------------------------------------- */
RT_DEV V3 clamp_radiance(V3 L) { /* :154-162, max_value = 100 */
    const Real max_value = 100.0;
    if (L.x > max_value || L.y > max_value || L.z > max_value) {
        Real max_c = max3(L);
        if (max_c > max_value) return scl(max_value / max_c, L);
    }
    return L;
}
RT_DEV Real power_heuristic(Real pdf_a, Real pdf_b) { /* :165-170 */
    Real a2 = pdf_a * pdf_a;
    Real b2 = pdf_b * pdf_b;
    Real denom = a2 + b2;
    return denom > 0 ? a2 / denom : 0.0;
}
template <int MS = RT_MS_FULL>
RT_DEV Real compute_light_pdf(const DScene& sc, V3 o, V3 d) { /* :173-188 */
    Real total_pdf = 0.0;
    Real light_select_pdf = 1.0 / sc.n_lights;
    for (int k = 0; k < sc.n_lights; ++k) {
        const rtr_light l = ld_const(sc.lights, k); /* wave-uniform index: scalar loads */
        total_pdf += light_pdf<MS>(l, o, d, sc.image_bytes) * light_select_pdf;
    }
    return total_pdf;
}

/* What a ray that leaves the scene adds to L (before the throughput-weighted sum):
 * mis_path_integrator.h:37-67, direct_light_integrator.h:41-54; the other integrators only know
 * the background colour.  The only infinite light flattened is the map-less EnvironmentLight,
 * whose Le is (1,1,1) (environmental_light.h:226-229). */
template <int INTEG, int MS = RT_MS_FULL>
RT_DEV V3 miss_radiance(const DScene& sc, V3 thr, V3 ro, V3 rd, int depth, bool specular_bounce, Real prev_bsdf_pdf) {
    if (MS == RT_MS_FULL && (INTEG == RTR_INTEGRATOR_MIS || INTEG == RTR_INTEGRATOR_NEE)) {
        V3 env = mk(0, 0, 0);
        bool found = false;
        for (int k = 0; k < sc.n_lights; ++k) {
            const rtr_light lk = ld_const(sc.lights, k);
            const int type = lk.type;
            if (type != RTR_LIGHT_ENV_UNIFORM && type != RTR_LIGHT_ENV_MAP) continue;
            V3 le = mk(1, 1, 1); /* environmental_light.h:226-229 */
            if (type == RTR_LIGHT_ENV_MAP) le = env_Le(env_map(lk, sc.image_bytes), rd);
            if (INTEG == RTR_INTEGRATOR_NEE)
                env = add(env, mul(thr, le)); /* L += throughput * Le, light by light */
            else
                env = add(env, le);
            found = true;
        }
        if (found) {
            if (INTEG == RTR_INTEGRATOR_NEE) return env;
            if (depth == 0 || specular_bounce) return mul(thr, env);
            const Real mis_weight = power_heuristic(prev_bsdf_pdf, compute_light_pdf<MS>(sc, ro, rd));
            return scl(mis_weight, mul(thr, env));
        }
    }
    return mul(thr, ld3(sc.background));
}

struct PathCounters {
    uint32_t closest, shadow;
};

/* The per-path state both integrators carry from bounce to bounce
 * (mis_path_integrator.h:28-32, rr_path_integrator.h:23-25). */
struct PathState {
    V3 ro, rd; /* current_ray origin / direction (direction is not normalised) */
    Real tm;   /* current_ray.time() */
    V3 thr;    /* throughput */
    V3 L;      /* radiance gathered by this camera sample */
    Real prev_bsdf_pdf;
    int depth;
    bool specular_bounce;
};
RT_DEV void path_begin(PathState& ps, V3 ro, V3 rd, Real tm) {
    ps.ro = ro, ps.rd = rd, ps.tm = tm;
    ps.thr = mk(1.0, 1.0, 1.0);
    ps.L = mk(0.0, 0.0, 0.0);
    ps.prev_bsdf_pdf = 0.0;
    ps.depth = 0;
    ps.specular_bounce = false;
}

/* a next-event-estimation connection waiting for its shadow ray */
struct ShadowReq {
    bool valid;
    V3 wi;      /* shadow_ray = ray(rec.p, wi, time 0), mis_path_integrator.h:210 */
    Real tmax;  /* ls.dist - 0.001 */
    V3 contrib; /* clamp_radiance(throughput * L_direct) if unshadowed */
};

/*
 * MISPathIntegrator::Li, first half of one loop iteration after a hit
 * (mis_path_integrator.h:69-103): emitted radiance with its MIS weight, then the light
 * sample of sample_lights_mis (:191-229).  The BSDF value and pdf are evaluated before the
 * shadow test (the reference evaluates them after it; they draw no random numbers), so the
 * connection can be resolved later by a separate shadow-ray stage.
 */
/* The same two halves serve three integrators (INTEG):
 *   RTR_INTEGRATOR_MIS  mis_path_integrator.h:69-146   (MIS-weighted emission + light sample, scatter() fallback)
 *   RTR_INTEGRATOR_NEE  direct_light_integrator.h:56-95 (emission only at depth 0 / after specular, light sample
 *                       without MIS weight and with the per-channel rescale of :133-139, no fallback)
 *   RTR_INTEGRATOR_PBR  pbr_path_integrator.h:38-68     (emission unweighted, no light sample, no fallback) */
/* the two parts of shade_a_mis, callable on their own (the material-sorted megakernel runs the first on the lane that
 * owns the path and the second on the lane that shades it): PART bit 0 = emission, bit 1 = light sample */
template <int MS = RT_MS_FULL, int INTEG = RTR_INTEGRATOR_MIS, int PART = 3>
RT_DEV void shade_a_mis(const DScene& sc, PathState& ps, const Hit& rec, const MatCtx& mc, V3 wo, uint32_t& rng,
                        ShadowReq& rq) {
    RT_REGION(RG_SHADE_A);
    const bool have_lights = sc.n_lights > 0;
    rq.valid = false;
    if (INTEG == RTR_INTEGRATOR_PBR) {
        if (PART & 1) ps.L = add(ps.L, mul(ps.thr, mat_emitted(mc, rec))); /* pbr_path_integrator.h:40-41 */
        return;
    }
    if (!(PART & 1)) {
    } else if (INTEG == RTR_INTEGRATOR_NEE) { /* direct_light_integrator.h:56-59 */
        if (ps.depth == 0 || ps.specular_bounce) ps.L = add(ps.L, mul(ps.thr, mat_emitted(mc, rec)));
    } else {
        V3 emitted = mat_emitted(mc, rec);
        if (len2(emitted) > 0) { /* mis_path_integrator.h:72-94 */
            V3 L_emit;
            if (ps.depth == 0 || ps.specular_bounce) {
                L_emit = mul(ps.thr, emitted);
            } else if (have_lights) {
                Real mis_weight = power_heuristic(ps.prev_bsdf_pdf, compute_light_pdf<MS>(sc, ps.ro, ps.rd));
                L_emit = scl(mis_weight, mul(ps.thr, emitted));
            } else {
                L_emit = mul(ps.thr, emitted);
            }
            ps.L = add(ps.L, ps.depth == 0 ? L_emit : clamp_radiance(L_emit));
        }
    }
    /* material::is_specular() is never overridden, so NEE runs at every hit */
    if ((PART & 2) && have_lights) {
        const int light_idx = rng_int(rng, 0, sc.n_lights - 1);
        const Real light_select_pdf = 1.0 / sc.n_lights;
        const Real uy = rng_next(rng); /* vec2 u(r(), r()): u.y takes the first draw (g++ order) */
        const Real ux = rng_next(rng);
        LightSample ls;
        if (sc.n_lights == 1) { /* the usual case: the record comes in through scalar loads, not 34 VGPRs */
            const rtr_light l0 = ld_const(sc.lights, 0);
            ls = light_sample<MS>(l0, rec.p, ux, uy, rng, sc.image_bytes);
        } else {
            const rtr_light li = ld_const(sc.lights, light_idx);
            ls = light_sample<MS>(li, rec.p, ux, uy, rng, sc.image_bytes);
        }
        if (ls.pdf > 0 && len2(ls.Li) > 0) {
            V3 f;
            Real bsdf_pdf = 0;
            if (INTEG == RTR_INTEGRATOR_MIS && !ls.is_delta)
                mat_eval_pdf<MS>(mc, rec, wo, ls.wi, f, bsdf_pdf); /* mis_path_integrator.h:215, :223 */
            else
                f = mat_eval<MS>(mc, wo, ls.wi);
            Real cos_theta = __builtin_fabs(dot(ls.wi, rec.n));
            V3 L_direct;
            if (INTEG == RTR_INTEGRATOR_NEE) { /* direct_light_integrator.h:125-139 */
                L_direct = divs(scl(cos_theta, mul(f, ls.Li)), ls.is_delta ? light_select_pdf : ls.pdf * light_select_pdf);
                const Real max_radiance = 100.0;
                if (L_direct.x > max_radiance) L_direct = scl(max_radiance / L_direct.x, L_direct);
                if (L_direct.y > max_radiance) L_direct = scl(max_radiance / L_direct.y, L_direct);
                if (L_direct.z > max_radiance) L_direct = scl(max_radiance / L_direct.z, L_direct);
            } else if (ls.is_delta) { /* mis_path_integrator.h:219-221: no BSDF sample can hit a delta light */
                L_direct = divs(scl(cos_theta, mul(f, ls.Li)), light_select_pdf);
            } else { /* mis_path_integrator.h:222-229 */
                Real lpdf = ls.pdf * light_select_pdf;
                Real mis_weight = power_heuristic(lpdf, bsdf_pdf);
                L_direct = divs(scl(mis_weight, scl(cos_theta, mul(f, ls.Li))), lpdf);
            }
            rq.valid = true;
            rq.wi = ls.wi;
            rq.tmax = ls.dist - 0.001;
            rq.contrib = INTEG == RTR_INTEGRATOR_NEE ? mul(ps.thr, L_direct) : clamp_radiance(mul(ps.thr, L_direct));
        }
    }
}

/* second half (mis_path_integrator.h:105-146): BSDF sampling with the legacy scatter()
 * fallback (MIS only), throughput update, Russian roulette.  Returns false when the path ends. */
template <int MS = RT_MS_FULL, int INTEG = RTR_INTEGRATOR_MIS>
RT_DEV bool shade_b_mis(const DScene& sc, PathState& ps, const Hit& rec, const MatCtx& mc, V3 wo, uint32_t& rng,
                        int rr_start) {
    RT_REGION(RG_SHADE_B);
    BSDFSample bs;
    if (!mat_sample<MS>(mc, rec, wo, bs, rng)) { /* :106-118 */
        if (INTEG != RTR_INTEGRATOR_MIS) return false; /* pbr_path_integrator.h:44-46, direct_light_integrator.h:67-69 */
        V3 attenuation, ndir;
        if (!mat_scatter<MS>(mc, ps.rd, rec, attenuation, ndir, rng)) return false;
        ps.thr = mul(ps.thr, attenuation);
        ps.ro = rec.p, ps.rd = ndir;
        ps.specular_bounce = false;
        ps.prev_bsdf_pdf = 0.0;
    } else {
        if (bs.pdf < 1e-8 && !bs.is_specular) return false;
        ps.specular_bounce = bs.is_specular;
        ps.prev_bsdf_pdf = bs.is_specular ? 0.0 : bs.pdf;
        Real cos_theta = __builtin_fabs(dot(bs.wi, rec.n));
        if (bs.is_specular)
            ps.thr = mul(ps.thr, bs.f);
        else
            ps.thr = mul(ps.thr, divs(scl(cos_theta, bs.f), bs.pdf));
        ps.ro = rec.p, ps.rd = bs.wi;
    }
    if (ps.depth >= rr_start) { /* :137-146 */
        Real p_survive = clampd(max3(ps.thr), 0.05, 0.95);
        if (rng_next(rng) > p_survive) return false;
        ps.thr = divs(ps.thr, p_survive);
    }
    return true;
}

/* RRPathInterator::Li after a hit (rr_path_integrator.h:36-55): two-sided legacy emission,
 * legacy scatter(), roulette clamp [0.005, 0.95] tested before the ray moves on. */
template <int MS = RT_MS_FULL>
RT_DEV bool shade_rr(const DScene& sc, PathState& ps, const Hit& rec, uint32_t& rng, int rr_start) {
    const MatCtx mc = mat_prepare<MS>(sc, rec);
    RT_REGION(RG_SHADE_RR);
    ps.L = add(ps.L, mul(ps.thr, mat_emitted_legacy(mc)));
    V3 attenuation, ndir;
    if (!mat_scatter<MS>(mc, ps.rd, rec, attenuation, ndir, rng)) return false;
    ps.thr = mul(ps.thr, attenuation);
    if (ps.depth >= rr_start) {
        Real p_survive = clampd(max3(ps.thr), 0.005, 0.95);
        if (rng_next(rng) > p_survive) return false;
        ps.thr = divs(ps.thr, p_survive);
    }
    ps.ro = rec.p, ps.rd = ndir;
    return true;
}

/* PathIntegrator (path_integrator.h:22-44): the reference recurses, emitted + attenuation * Li(next);
 * here the same terms are summed front to back (throughput-weighted), which differs only in
 * rounding (<= 1e-15 relative).  No roulette; depth limited by max_depth. */
template <int MS = RT_MS_FULL>
RT_DEV bool shade_path(const DScene& sc, PathState& ps, const Hit& rec, uint32_t& rng) {
    const MatCtx mc = mat_prepare<MS>(sc, rec);
    ps.L = add(ps.L, mul(ps.thr, mat_emitted_legacy(mc)));
    V3 attenuation, ndir;
    if (!mat_scatter<MS>(mc, ps.rd, rec, attenuation, ndir, rng)) return false;
    ps.thr = mul(ps.thr, attenuation);
    ps.ro = rec.p, ps.rd = ndir;
    return true;
}

/*
 * One whole loop iteration of Integrator::Li in registers (megakernel / unit tests):
 * closest hit, shading, inline shadow ray.  Returns false when the camera sample is
 * finished (ps.L is then its radiance).
 */
template <int INTEG, int TRAV>
__device__ __forceinline__ bool bounce(const DScene& sc, PathState& ps, uint32_t& rng, const Stack st,
                                       int max_depth, int rr_start, PathCounters& cnt) {
    Hit rec;
    rec.u = 0, rec.v = 0;
    ++cnt.closest;
    if (!cast_closest<TRAV>(sc, ps.ro, ps.rd, ps.tm, rec, rng, st)) {
        ps.L = add(ps.L, miss_radiance<INTEG>(sc, ps.thr, ps.ro, ps.rd, ps.depth, ps.specular_bounce, ps.prev_bsdf_pdf));
        return false;
    }
    bool go;
    if (INTEG == RTR_INTEGRATOR_RR) {
        go = shade_rr(sc, ps, rec, rng, rr_start);
    } else if (INTEG == RTR_INTEGRATOR_PATH) {
        go = shade_path(sc, ps, rec, rng);
        if (!go) return false;
        if (++ps.depth < max_depth) return true;
        return false; /* path_integrator.h:27-29: the call at depth 0 contributes nothing */
    } else {
        V3 wo = neg(unit(ps.rd));
        ShadowReq rq;
        const MatCtx mc = mat_prepare<RT_MS_FULL>(sc, rec);
        shade_a_mis<RT_MS_FULL, INTEG>(sc, ps, rec, mc, wo, rng, rq);
        if (rq.valid) {
            ++cnt.shadow;
            if (!cast_shadow<TRAV>(sc, rec.p, rq.wi, rq.tmax, rng, st)) ps.L = add(ps.L, rq.contrib);
        } /* else the reference adds clamp_radiance(throughput * 0) = +0 (:99-103): no effect */
        go = shade_b_mis<RT_MS_FULL, INTEG>(sc, ps, rec, mc, wo, rng, rr_start);
    }
    if (!go) return false;
    return ++ps.depth < max_depth;
}
