/*
 * rtr_hip_test.h -- entry points of librtr_hip_test.so, a SEPARATE library next to librtr_hip.so (it links against
 * it): device unit kernels that run the product's own device functions (csrc/rt_device.h) over golden-vector records
 * (include/rtr_testrec.h), one lane per record, plus counter-calibration and instruction-level checks.  They exist
 * so tests can compare the HIP path with the oracle below the whole-image level; none of this code is in the
 * product library, and a renderer integration only needs rtr_hip.h.
 */
#ifndef RTR_HIP_TEST_H
#define RTR_HIP_TEST_H

#include "rtr_hip.h"
#include "rtr_testrec.h"

#ifdef __cplusplus
extern "C" {
#endif

/* In-place on HOST arrays of records: inputs are read, outputs overwritten. */
int rtr_test_hits(rtr_context* ctx, rtr_hit_record* recs, int64_t n);
int rtr_test_materials(rtr_context* ctx, rtr_mat_record* recs, int64_t n);
int rtr_test_lights(rtr_context* ctx, rtr_light_record* recs, int64_t n);
int rtr_test_li(rtr_context* ctx, const rtr_render_params* params, rtr_li_record* recs, int64_t n);

/* Make rtr_test_hits (which takes no render params) use the reference-order traversal. */
int rtr_test_reference_order(rtr_context* ctx, int on);

/* Counter calibration for profiles/: stream `n_doubles` doubles through the access shape of the wavefront
 * stages (one 8-byte word per lane, consecutive lanes consecutive words: out[i] = in[i] + 1), `repeat` times.
 * The launch reads and writes exactly n_doubles * 8 bytes each per repetition, which is what rocprofv3's
 * FETCH_SIZE / WRITE_SIZE of the same run are compared with (MI355X_MICROARCH.md, HBM: widths other than
 * 16 bytes per lane are uncalibrated).  Returns RTR_OK or a negative status. */
int rtr_test_stream8(rtr_context* ctx, int64_t n_doubles, int repeat);

/* The samplers take sin and cos of phi = 2 pi r for r = s * 2^-32, s any state of the 32-bit generator
 * (vec3.h:261-269, material.h:268-275); the device evaluates both with ONE sincos().  This walks ALL 2^32 values of
 * s and counts those where sincos(phi) and the pair sin(phi), cos(phi) differ in any bit: *mismatches must be 0. */
int rtr_test_sincos_exhaustive(rtr_context* ctx, uint64_t* mismatches);

/* Measured issue costs for bench.py's `valu_slot_utilisation`: shader cycles a wave spends per instruction of one class
 * while four waves share each SIMD (so a pipe-bound class reads four times its pipe cost), classes in this order:
 * v_fma_f64, v_add_f64, v_mul_f64, v_rcp_f64, v_rsq_f64, v_cmp_lt_f64, v_cndmask_b32, v_mov_b32, v_fma_f32, s_and_b64,
 * v_div_scale_f64, v_div_fixup_f64, {v_cmp_lt_f64 + dependent s_and_b64}, v_cndmask_b32 with a scalar-pair mask, v_min_f32,
 * {v_cmp_lt_f32 + v_cndmask_b32}, v_cndmask_b32 into four destinations, {v_cmp_lt_f64 + v_cndmask_b32}.  Fills
 * cycles_per_inst[0 .. n) (n <= 18). */
int rtr_test_issue_rates(rtr_context* ctx, double* cycles_per_inst, int n);

/* The primitive tests divide many numerators by the same ray-direction component through a shared refined reciprocal
 * (rt_device.h: div_shared) instead of the compiler's eleven-instruction division.  This compares the two on 2^32
 * operand pairs from the range the short form is used in; *mismatches (quotients that differ in any bit) must be 0. */
int rtr_test_shared_division(rtr_context* ctx, uint64_t* mismatches);

#ifdef __cplusplus
}
#endif
#endif /* RTR_HIP_TEST_H */
