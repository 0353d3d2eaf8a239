/*
 * rt_test_kernels.h -- device unit kernels of librtr_hip_test.so (include/rtr_hip_test.h): the library's own device
 * functions (rt_device.h) over golden-vector records, one lane per record, plus counter-calibration and
 * instruction-level checks.  Test infrastructure: nothing of this is linked into librtr_hip.so.
 */
#pragma once

#include "rt_kernels.h"
#include "rtr_testrec.h"

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
    if (TRAV == RT_TRAV_FAST || TRAV == RT_TRAV_TOP) {
        int ref, inst;
        h = trace_fast<false, true, false, TRAV == RT_TRAV_TOP>(sc, sub_scene0(sc), ld3(r.o), ld3(r.d), r.time, r.t_min, tmax, ref, inst, st, 0);
        if (h) fast_finish<true>(sc, ld3(r.o), ld3(r.d), r.time, tmax, ref, inst, rec);
    } else if (rt_is_program(TRAV)) {
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
    int ia = threadIdx.x, ib = 3, ic = 0;
    unsigned long long m = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile(RT_REP32("v_fma_f64 %0, %1, %2, %1\n") : "+v"(d) : "v"(b), "v"(c));
        if (KIND == 1) asm volatile(RT_REP32("v_add_f64 %0, %1, %2\n") : "+v"(d) : "v"(b), "v"(c));
        if (KIND == 2) asm volatile(RT_REP32("v_mul_f64 %0, %1, %2\n") : "+v"(d) : "v"(b), "v"(c));
        if (KIND == 3) asm volatile(RT_REP32("v_rcp_f64 %0, %1\n") : "+v"(d) : "v"(b));
        if (KIND == 4) asm volatile(RT_REP32("v_rsq_f64 %0, %1\n") : "+v"(d) : "v"(b));
        if (KIND == 5) asm volatile(RT_REP32("v_cmp_lt_f64 %0, %1, %2\n") : "=s"(m) : "v"(b), "v"(c));
        if (KIND == 6) asm volatile(RT_REP32("v_cndmask_b32 %0, %1, %2, vcc\n") : "=v"(ic) : "v"(ib), "v"(ia) : "vcc");
        if (KIND == 7) asm volatile(RT_REP32("v_mov_b32 %0, %1\n") : "+v"(ia) : "v"(ib));
        if (KIND == 8) asm volatile(RT_REP32("v_fma_f32 %0, %1, %2, %1\n") : "+v"(fa) : "v"(fb), "v"(fc));
        if (KIND == 9) asm volatile(RT_REP32("s_and_b64 %0, %0, exec\n") : "+s"(m) : : "scc");
        if (KIND == 10) asm volatile(RT_REP32("v_div_scale_f64 %0, vcc, %1, %2, %1\n") : "+v"(d) : "v"(b), "v"(c) : "vcc");
        if (KIND == 11) asm volatile(RT_REP32("v_div_fixup_f64 %0, %1, %2, %1\n") : "+v"(d) : "v"(b), "v"(c));
        if (KIND == 13) asm volatile(RT_REP32("v_cndmask_b32_e64 %0, %1, %2, %3\n") : "=v"(ic) : "v"(ib), "v"(ia), "s"(m));
        if (KIND == 14) asm volatile(RT_REP32("v_min_f32 %0, %1, %2\n") : "=v"(fa) : "v"(fb), "v"(fc));
        if (KIND == 15) asm volatile(RT_REP32("v_cmp_lt_f32 vcc, %1, %2\nv_cndmask_b32 %0, %1, %2, vcc\n") : "=v"(fa) : "v"(fb), "v"(fc) : "vcc");
        if (KIND == 16) { int i0, i1, i2, i3; asm volatile(RT_REP32("v_cndmask_b32 %0, %4, %5, vcc\nv_cndmask_b32 %1, %4, %5, vcc\nv_cndmask_b32 %2, %4, %5, vcc\nv_cndmask_b32 %3, %4, %5, vcc\n") : "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3) : "v"(ib), "v"(ia) : "vcc"); ic = i0 + i1 + i2 + i3; }
        if (KIND == 17) asm volatile(RT_REP32("v_cmp_lt_f64 vcc, %1, %2\nv_cndmask_b32 %0, %3, %4, vcc\n") : "=v"(ic) : "v"(b), "v"(c), "v"(ib), "v"(ia) : "vcc");
        if (KIND == 12) /* the rectangle test's mix: one compare into a scalar pair and the scalar AND that uses it */
            asm volatile(RT_REP32("v_cmp_lt_f64 %0, %1, %2\ns_and_b64 %0, %0, exec\n") : "=s"(m) : "v"(b), "v"(c) : "scc");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (d == 12345.678 || fa == 1.5f || ia == -77 || ic == -78 || m == 0x1234567ull) out[2] = 1; /* keep the results alive */
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], t1 - t0);
        atomicAdd(&out[1], 1ull);
    }
}

__global__ void __launch_bounds__(RTR_BLOCK) k_stream8(const double* __restrict__ in, double* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * RTR_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * RTR_BLOCK)
        out[i] = in[i] + 1.0;
}
