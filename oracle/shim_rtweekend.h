/*
 * shim_rtweekend.h -- TEST INFRASTRUCTURE (oracle), never part of the product.
 *
 * Pre-included (g++ -include) in front of the UNMODIFIED reference sources when
 * building oracle/_ref/ref_harness.  It claims the include guard of the
 * reference's core/rtweekend.h so that header's body is skipped everywhere, and
 * provides the same public names (core/rtweekend.h:11-50) on top of a generator
 * whose state the harness can set: the reference seeds its thread_local
 * xorshift32 from the thread id (rtweekend.h:26-27), which makes its output
 * non-reproducible.  The update rule (13,17,5) and the 2^-32 scale
 * are those of rtweekend.h:29-33.
 */
#ifndef RTWEEKEND_H
#define RTWEEKEND_H

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <limits>
#include <memory>
#include <random>
#include <thread>

using std::make_shared;
using std::make_unique;
using std::shared_ptr;
using std::sqrt;
using std::unique_ptr;

constexpr double infinity = std::numeric_limits<double>::infinity();
constexpr double pi = 3.1415926535897932385;

inline constexpr double degrees_to_radians(double degrees) { return degrees * pi / 180.0; }

/* settable generator state, one per thread */
inline uint32_t& rtr_ref_rng_state() {
    static thread_local uint32_t state =
        static_cast<uint32_t>(std::hash<std::thread::id>{}(std::this_thread::get_id()));
    return state;
}

inline double random_double() {
    uint32_t& s = rtr_ref_rng_state();
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s * 2.3283064365386963e-10;
}

inline double random_double(double min, double max) noexcept { return min + (max - min) * random_double(); }

inline double clamp(double x, double min, double max) noexcept {
    if (x < min) return min;
    if (x > max) return max;
    return x;
}

inline int random_int(int min, int max) { return static_cast<int>(random_double(min, max + 1)); }

#endif
