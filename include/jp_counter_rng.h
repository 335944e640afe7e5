/* include/jp_counter_rng.h -- the counter-based sampler shared by the GPU kernels, the host mirror and
 * the test oracle ("same RNG seeds", BASELINE.json north_star; SURVEY.md section 8c tier T1, Appendix B).
 *
 * The reference consumes one sequential std::mt19937_64 stream per 20-row band (sampler.h:26,
 * integrator.cc:66) -- not reproducible by a parallel device.  This header defines the replacement
 * stream that is plugged into the reference through its own FSampler virtual interface
 * (sampler.h:64-105) and used verbatim on the device:
 *
 *   key   = jp_rng_key(seed, x, y, s)      set at GetCameraSample((x,y)) for sample index s
 *   value = jp_rng_float(key, dim)         dim = 0,1,2,... one per draw, in the draw order of
 *                                          FPathIntegratorIteration::Li (integrator.cc:316-403)
 * Within a 2-draw call the FIRST draw is .x (the stock sampler's order is compiler-dependent,
 * sampler.h:49-52; this definition removes the ambiguity).
 * Values are multiples of 2^-24 in [0,1): exactly representable, never 1.0f.
 * Plain C99 / C++ / HIP device code (JP_HD expands to __host__ __device__ under hipcc).
 */
#ifndef JP_COUNTER_RNG_H
#define JP_COUNTER_RNG_H
#include <stdint.h>

#if defined(__HIPCC__)
#define JP_HD __host__ __device__
#else
#define JP_HD
#endif

/* 32-bit finalizer ("lowbias32", C. Wellons): bijective, full avalanche. */
static inline JP_HD uint32_t jp_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

static inline JP_HD uint32_t jp_rng_key(uint32_t seed, uint32_t x, uint32_t y, uint32_t sample)
{
    uint32_t k = jp_mix32(seed ^ 0x9E3779B9U);
    k = jp_mix32(k + x);
    k = jp_mix32(k + y);
    k = jp_mix32(k + sample);
    return k;
}

static inline JP_HD uint32_t jp_rng_u32(uint32_t key, uint32_t dim)
{
    return jp_mix32(key + dim * 0x9E3779B9U);
}

static inline JP_HD float jp_rng_float(uint32_t key, uint32_t dim)
{
    return (float)(jp_rng_u32(key, dim) >> 8) * (1.0f / 16777216.0f);
}
#endif
