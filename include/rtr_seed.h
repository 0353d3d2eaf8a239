/*
 * rtr_seed.h -- the per-sample RNG seeding scheme shared by the oracle, the
 * host layer and the device kernels.
 *
 * The reference seeds its thread_local xorshift32 (core/rtweekend.h:24-34) from
 * std::hash<std::thread::id> and hands tiles to threads dynamically, so its
 * output is not reproducible.  Parity on "identical RNG seeds"
 * therefore fixes the generator state at the start of every camera sample as a
 * pure function of (render seed, pixel, sample index).  The state is never 0
 * (xorshift32 would stay at 0).
 */
#ifndef RTR_SEED_H
#define RTR_SEED_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RTR_HD __host__ __device__
#else
#define RTR_HD
#endif

/* murmur3 32-bit finalizer */
static inline RTR_HD uint32_t rtr_mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

static inline RTR_HD uint32_t rtr_sample_seed_inline(uint32_t seed, int32_t image_width, int32_t i, int32_t j,
                                                     int32_t s) {
    uint32_t pix = (uint32_t)j * (uint32_t)image_width + (uint32_t)i;
    uint32_t h = rtr_mix32(pix * 0x9E3779B1u + (uint32_t)s * 0x85EBCA77u + 0xC2B2AE3Du + seed * 0x27D4EB2Fu);
    return h ? h : 0x9E3779B9u;
}

#endif /* RTR_SEED_H */
