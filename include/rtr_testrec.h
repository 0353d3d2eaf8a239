/*
 * rtr_testrec.h -- packed record layouts of the golden vectors (the .bin files under tests/golden).
 *
 * Shared by oracle/ref_harness.cpp (which fills them by calling the reference),
 * oracle/rt_oracle.cpp (CPU restatement) and the library's rtr_test_* entry points
 * (device unit kernels), so one numpy dtype in tests/ describes all three.
 * Inputs first, outputs after; all doubles are IEEE binary64.
 */
#ifndef RTR_TESTREC_H
#define RTR_TESTREC_H

#include <stdint.h>

#pragma pack(push, 1)

/* one camera sample through Integrator::Li */
typedef struct rtr_li_record {
    int32_t i, j, s;            /* in: pixel and sample index */
    uint32_t rng_exit;          /* out: xorshift32 state after Li returned (pins the draw count) */
    double L[3];                /* out: radiance */
    int32_t n_closest, n_shadow; /* out: scene.hit calls with t_max = inf / finite */
} rtr_li_record; /* 48 bytes */

/* one hittable::hit call on the scene root */
typedef struct rtr_hit_record {
    double o[3], d[3], time, t_min, t_max; /* in */
    uint32_t rng_in, rng_out;              /* in / out (constant_medium draws inside hit) */
    int32_t hit, front_face, material, pad; /* out */
    double t, p[3], n[3], u, v;             /* out; u,v are NaN where the reference leaves them unset */
} rtr_hit_record; /* 168 bytes */

/* material::sample / eval / pdf / emitted on a synthetic hit_record */
typedef struct rtr_mat_record {
    int32_t material, front_face; /* in */
    uint32_t rng_in, rng_out;     /* in / out */
    double p[3], n[3], u, v, wo[3], wi_in[3]; /* in */
    int32_t sample_ok, is_specular, is_transmission, pad; /* out */
    double s_wi[3], s_f[3], s_pdf; /* out: BSDFSample */
    double eval[3], pdf;           /* out: eval(rec,wo,wi_in), pdf(rec,wo,wi_in) */
    double emitted[3];             /* out: emitted(rec,wo) */
} rtr_mat_record; /* 256 bytes */

/* Light::sample(p,u) and Light::pdf(p,dir) */
typedef struct rtr_light_record {
    int32_t light, pad;            /* in */
    double p[3], u[2], dir[3];     /* in */
    double Li[3], wi[3], pdf, dist; /* out */
    int32_t is_delta, pad2;
    double pdf_dir;
} rtr_light_record; /* 152 bytes */

#pragma pack(pop)

/* RNG known-answer block per seed (doubles): seed, 16 x random_double, 8 x random_int(0,9),
 * vec3::random(-1,1) xyz, vec2(r,r) xy, random_in_unit_disk xyz, random_in_unit_sphere xyz,
 * random_unit_vector xyz, random_cosine_direction xyz, final state */
#define RTR_RNG_BLOCK_DOUBLES (1 + 16 + 8 + 3 + 2 + 3 + 3 + 3 + 3 + 1)

#endif /* RTR_TESTREC_H */
