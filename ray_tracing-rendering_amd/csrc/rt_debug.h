/*
 * rt_debug.h -- the seam between librtr_hip.so and librtr_hip_test.so (NOT part of include/: a renderer integration
 * never sees it).  The test library runs its own unit kernels over the device functions of rt_device.h; what it needs
 * from a context is where the uploaded scene lives and which stream and traversal a call would use.
 */
#pragma once

#include "rt_device.h"
#include "rtr_hip.h"

struct rtr_debug_view {
    DScene ds;          /* device pointers of the uploaded scene */
    hipStream_t stream; /* the context's stream */
    int device, n_cus, n_materials;
    int trav;           /* RT_TRAV_* a call with `flags` uses (RT_TRAV_FLAT reported as RT_TRAV_FAST: same results) */
    size_t stack_bytes; /* LDS traversal stack per workgroup of that traversal */
};
struct rtr_debug_li_out { /* per camera sample: what rtr_li_samples drops */
    double L[3];
    uint32_t rng_exit;
    int32_t n_closest, n_shadow, pad;
};
extern "C" {
int rtr_debug_view_get(rtr_context* ctx, int flags, rtr_debug_view* view, size_t size_of_view);
int rtr_debug_li(rtr_context* ctx, const rtr_render_params* params, const int32_t* ijs, rtr_debug_li_out* out, int64_t n);
void rtr_debug_set_error(rtr_context* ctx, const char* msg); /* rtr_last_error() of a failing rtr_test_* call */
}
