/*
 * rtr_test.hip -- librtr_hip_test.so: the entry points of include/rtr_hip_test.h.  Test infrastructure, built next to
 * librtr_hip.so and linked against it; it reaches a context only through the seam of csrc/rt_debug.h.
 */
#include "rt_debug.h"
#include "rt_test_kernels.h"
#include "rtr_hip_test.h"

#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct TestState {
    void* buf = nullptr;
    size_t cap = 0;
    bool reference_order = false;
};
std::mutex g_mu;
std::map<rtr_context*, TestState> g_state; /* contexts are few and live as long as a test session */

TestState& state_of(rtr_context* c) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_state[c];
}
int fail(rtr_context* c, int code, const std::string& msg) {
    rtr_debug_set_error(c, msg.c_str());
    return code;
}
#define TCHK(ctx, expr)                                                                                  \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return fail(ctx, RTR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int ensure(rtr_context* c, TestState& t, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (t.cap >= bytes) return RTR_OK;
    if (t.buf) TCHK(c, hipFree(t.buf));
    t.buf = nullptr, t.cap = 0;
    if (hipMalloc(&t.buf, bytes) != hipSuccess) return fail(c, RTR_ERR_NOMEM, "hipMalloc of the test record buffer");
    t.cap = bytes;
    return RTR_OK;
}
/* view of the context + the records on the device */
int begin(rtr_context* c, const void* recs, int64_t n, size_t rec_size, rtr_debug_view& v, TestState*& t, bool need_scene = true) {
    if (!c) return RTR_ERR_INVALID;
    t = &state_of(c);
    if (need_scene) {
        if (int rc = rtr_debug_view_get(c, t->reference_order ? RTR_FLAG_REFERENCE_ORDER : 0, &v, sizeof v)) return rc;
    }
    if (n < 0 || (n > 0 && !recs)) return fail(c, RTR_ERR_INVALID, "bad record array");
    if (int rc = rtr_synchronize(c)) return rc;
    if (int rc = ensure(c, *t, (size_t)n * rec_size)) return rc;
    if (n) TCHK(c, hipMemcpy(t->buf, recs, (size_t)n * rec_size, hipMemcpyHostToDevice));
    return RTR_OK;
}
int end(rtr_context* c, hipStream_t stream, TestState* t, void* recs, int64_t n, size_t rec_size) {
    TCHK(c, hipGetLastError());
    TCHK(c, hipStreamSynchronize(stream));
    if (n) TCHK(c, hipMemcpy(recs, t->buf, (size_t)n * rec_size, hipMemcpyDeviceToHost));
    return RTR_OK;
}
dim3 grid_of(int64_t n) { return dim3((unsigned)((n + RTR_BLOCK - 1) / RTR_BLOCK)); }

template <typename K>
int set_lds(rtr_context* c, K kernel, size_t bytes) {
    if (bytes > 160 * 1024)
        return fail(c, RTR_ERR_UNSUPPORTED, "this traversal of the scene needs a deeper stack than 160 KiB of LDS holds");
    if (bytes > 64 * 1024)
        TCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return RTR_OK;
}
/* a bare context: stream, device and CU count without a scene */
int bare(rtr_context* c, hipStream_t& stream, int& n_cus, TestState*& t) {
    if (!c) return RTR_ERR_INVALID;
    t = &state_of(c);
    if (int rc = rtr_synchronize(c)) return rc;
    stream = nullptr; /* the kernels below are self-contained: the null stream orders them after everything */
    hipDeviceProp_t prop;
    int dev = 0;
    TCHK(c, hipGetDevice(&dev));
    TCHK(c, hipGetDeviceProperties(&prop, dev));
    n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return RTR_OK;
}

} // namespace

extern "C" {

int rtr_test_hits(rtr_context* c, rtr_hit_record* recs, int64_t n) {
    rtr_debug_view v;
    TestState* t;
    int rc = begin(c, recs, n, sizeof *recs, v, t);
    if (rc || n == 0) return rc;
    TCHK(c, hipSetDevice(v.device));
    DScene ds = v.ds;
    ds.needs_uv = 1; /* the vectors pin u,v although no flattened texture of these scenes reads them */
    auto* d = static_cast<rtr_hit_record*>(t->buf);
    const size_t lds = v.stack_bytes;
#define RTR_HITS(T)                                                                                          \
    do {                                                                                                      \
        if ((rc = set_lds(c, k_test_hits<T>, lds))) return rc;                                                 \
        hipLaunchKernelGGL(k_test_hits<T>, grid_of(n), dim3(RTR_BLOCK), lds, v.stream, ds, d, (long long)n);   \
    } while (0)
    if (v.trav == RT_TRAV_FAST)
        RTR_HITS(RT_TRAV_FAST);
    else if (v.trav == RT_TRAV_TOP)
        RTR_HITS(RT_TRAV_TOP);
    else if (v.trav == RT_TRAV_PROGRAM)
        RTR_HITS(RT_TRAV_PROGRAM_EXT);
    else if (v.trav == RT_TRAV_MEDIA)
        RTR_HITS(RT_TRAV_MEDIA);
    else
        RTR_HITS(RT_TRAV_EXACT);
#undef RTR_HITS
    return end(c, v.stream, t, recs, n, sizeof *recs);
}

int rtr_test_materials(rtr_context* c, rtr_mat_record* recs, int64_t n) {
    rtr_debug_view v;
    TestState* t;
    int rc = begin(c, recs, n, sizeof *recs, v, t);
    if (rc || n == 0) return rc;
    for (int64_t k = 0; k < n; ++k)
        if (recs[k].material < 0 || recs[k].material >= v.n_materials) return fail(c, RTR_ERR_INVALID, "material index out of range");
    TCHK(c, hipSetDevice(v.device));
    hipLaunchKernelGGL(k_test_materials, grid_of(n), dim3(RTR_BLOCK), 0, v.stream, v.ds, static_cast<rtr_mat_record*>(t->buf),
                       (long long)n);
    return end(c, v.stream, t, recs, n, sizeof *recs);
}

int rtr_test_lights(rtr_context* c, rtr_light_record* recs, int64_t n) {
    rtr_debug_view v;
    TestState* t;
    int rc = begin(c, recs, n, sizeof *recs, v, t);
    if (rc || n == 0) return rc;
    for (int64_t k = 0; k < n; ++k)
        if (recs[k].light < 0 || recs[k].light >= v.ds.n_lights) return fail(c, RTR_ERR_INVALID, "light index out of range");
    TCHK(c, hipSetDevice(v.device));
    hipLaunchKernelGGL(k_test_lights, grid_of(n), dim3(RTR_BLOCK), 0, v.stream, v.ds, static_cast<rtr_light_record*>(t->buf),
                       (long long)n);
    return end(c, v.stream, t, recs, n, sizeof *recs);
}

/* Integrator::Li per camera sample with the RNG state at exit and the segment counts: the product's own per-ray kernel
 * (rtr_li_samples runs the same one and drops those fields) */
int rtr_test_li(rtr_context* c, const rtr_render_params* p, rtr_li_record* recs, int64_t n) {
    if (!c || !p) return RTR_ERR_INVALID;
    if (n < 0 || (n > 0 && !recs)) return fail(c, RTR_ERR_INVALID, "bad record array");
    std::vector<int32_t> ijs((size_t)n * 3);
    for (int64_t k = 0; k < n; ++k) ijs[3 * k] = recs[k].i, ijs[3 * k + 1] = recs[k].j, ijs[3 * k + 2] = recs[k].s;
    std::vector<rtr_debug_li_out> out((size_t)n);
    rtr_render_params q = *p;
    if (state_of(c).reference_order) q.flags |= RTR_FLAG_REFERENCE_ORDER;
    int rc = rtr_debug_li(c, &q, ijs.data(), out.data(), n);
    if (rc) return rc;
    for (int64_t k = 0; k < n; ++k) {
        for (int a = 0; a < 3; ++a) recs[k].L[a] = out[(size_t)k].L[a];
        recs[k].rng_exit = out[(size_t)k].rng_exit;
        recs[k].n_closest = out[(size_t)k].n_closest, recs[k].n_shadow = out[(size_t)k].n_shadow;
    }
    return RTR_OK;
}

int rtr_test_reference_order(rtr_context* c, int on) {
    if (!c) return RTR_ERR_INVALID;
    state_of(c).reference_order = on != 0;
    return RTR_OK;
}

int rtr_test_stream8(rtr_context* c, int64_t n_doubles, int repeat) {
    if (!c || n_doubles <= 0 || repeat <= 0) return RTR_ERR_INVALID;
    hipStream_t stream;
    int n_cus;
    TestState* t;
    int rc = bare(c, stream, n_cus, t);
    if (rc) return rc;
    if ((rc = ensure(c, *t, (size_t)n_doubles * 16))) return rc;
    double* in = static_cast<double*>(t->buf);
    double* out = in + n_doubles;
    TCHK(c, hipMemsetAsync(in, 0, (size_t)n_doubles * 16, stream));
    for (int r = 0; r < repeat; ++r)
        hipLaunchKernelGGL(k_stream8, dim3((unsigned)(n_cus * 16)), dim3(RTR_BLOCK), 0, stream, in, out, (long long)n_doubles);
    TCHK(c, hipGetLastError());
    TCHK(c, hipStreamSynchronize(stream));
    return RTR_OK;
}

int rtr_test_sincos_exhaustive(rtr_context* c, uint64_t* mismatches) {
    if (!c || !mismatches) return RTR_ERR_INVALID;
    hipStream_t stream;
    int n_cus;
    TestState* t;
    int rc = bare(c, stream, n_cus, t);
    if (rc) return rc;
    if ((rc = ensure(c, *t, 8))) return rc;
    TCHK(c, hipMemsetAsync(t->buf, 0, 8, stream));
    hipLaunchKernelGGL(k_test_sincos, dim3((unsigned)(n_cus * 16)), dim3(RTR_BLOCK), 0, stream, static_cast<unsigned long long*>(t->buf));
    TCHK(c, hipGetLastError());
    unsigned long long h = 0;
    TCHK(c, hipMemcpy(&h, t->buf, 8, hipMemcpyDeviceToHost));
    *mismatches = h;
    return RTR_OK;
}

int rtr_test_issue_rates(rtr_context* c, double* cycles_per_inst, int n) {
    if (!c || !cycles_per_inst || n < 0) return RTR_ERR_INVALID;
    hipStream_t stream;
    int n_cus;
    TestState* t;
    int rc = bare(c, stream, n_cus, t);
    if (rc) return rc;
    if ((rc = ensure(c, *t, 32))) return rc;
    auto* d = static_cast<unsigned long long*>(t->buf);
    const int iters = 4096;
    const dim3 grid((unsigned)(n_cus * 4));
    for (int k = 0; k < n && k < 18; ++k) {
        TCHK(c, hipMemsetAsync(d, 0, 32, stream));
#define RTR_RATE(K) case K: hipLaunchKernelGGL(k_test_issue_rate<K>, grid, dim3(RTR_BLOCK), 0, stream, d, iters, 1.25); break
        switch (k) {
            RTR_RATE(0); RTR_RATE(1); RTR_RATE(2); RTR_RATE(3); RTR_RATE(4); RTR_RATE(5); RTR_RATE(6);
            RTR_RATE(7); RTR_RATE(8); RTR_RATE(9); RTR_RATE(10); RTR_RATE(11); RTR_RATE(12);
            RTR_RATE(13); RTR_RATE(14); RTR_RATE(15); RTR_RATE(16); RTR_RATE(17);
        }
#undef RTR_RATE
        TCHK(c, hipGetLastError());
        unsigned long long h[2] = {0, 0};
        TCHK(c, hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        cycles_per_inst[k] = h[1] ? (double)h[0] / (double)h[1] / (32.0 * iters * (k == 12 || k == 15 || k == 17 ? 2 : (k == 16 ? 4 : 1))) : 0.0;
    }
    return RTR_OK;
}

int rtr_test_shared_division(rtr_context* c, uint64_t* mismatches) {
    if (!c || !mismatches) return RTR_ERR_INVALID;
    hipStream_t stream;
    int n_cus;
    TestState* t;
    int rc = bare(c, stream, n_cus, t);
    if (rc) return rc;
    if ((rc = ensure(c, *t, 8))) return rc;
    TCHK(c, hipMemsetAsync(t->buf, 0, 8, stream));
    const unsigned blocks = 4096, per_thread = (unsigned)((1ull << 32) / ((unsigned long long)blocks * RTR_BLOCK));
    hipLaunchKernelGGL(k_test_shared_div, dim3(blocks), dim3(RTR_BLOCK), 0, stream, static_cast<unsigned long long*>(t->buf), per_thread);
    TCHK(c, hipGetLastError());
    unsigned long long h = 0;
    TCHK(c, hipMemcpy(&h, t->buf, 8, hipMemcpyDeviceToHost));
    *mismatches = h;
    return RTR_OK;
}

} /* extern "C" */
