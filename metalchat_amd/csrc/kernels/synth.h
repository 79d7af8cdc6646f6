// Counter-based synthetic weights shared by the host (mc_synth_* in the C ABI, used by tests to
// regenerate any element) and the device fill kernels (bench.py at full model size).  Integer
// hashing plus correctly rounded float ops only, so both sides produce identical bits.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MC_HD __host__ __device__ __forceinline__
#else
#define MC_HD inline
#endif

namespace mcsynth {

MC_HD uint64_t
mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

MC_HD uint64_t
key(uint64_t seed, uint32_t matrix_id, uint32_t a, uint32_t b)
{
    return mix(mix(seed + 0x9E3779B97F4A7C15ull * (matrix_id + 1)) + ((uint64_t)a << 32 | b));
}

// int4 in [-7,7] (bits 4) or int8 in [-127,127] (bits 8), ZERO MEAN: the most negative code is
// folded onto 0.  A uniform draw over [-8,7] has mean -0.5, i.e. every row of Wd carries the
// common component -0.5 s; against a hidden row with a non-zero mean that adds the same
// -0.5 s * sum(x) ~ -4 mean(x) to every output of a 4096-wide linear, the mean of the residual stream
// grows ~100x per layer and a 32-layer model reaches NaN logits -- the benchmark would then time
// arithmetic on NaNs and the no-max-shift softmax the synthetic scales are sized for is moot.
MC_HD int32_t
weight(uint64_t seed, uint32_t matrix_id, uint32_t row, uint32_t col, int32_t bits)
{
    const uint64_t h = key(seed, matrix_id, row, col);
    const int32_t q = bits == 4 ? (int32_t)((h >> 40) & 15) - 8 : (int32_t)((h >> 40) & 255) - 128;
    return q == (bits == 4 ? -8 : -128) ? 0 : q;
}

// U(0.5,1.5) / (sqrt(in) * 2^(bits-1))
MC_HD float
scale(uint64_t seed, uint32_t matrix_id, uint32_t row, uint32_t group, int32_t in_features,
      int32_t bits)
{
    const uint64_t h = key(seed ^ 0x5CA1E5ull, matrix_id, row, group);
    const float u = (float)(uint32_t)((h >> 40) & 0xFFFFFF) * (1.0f / 16777216.0f);
    const float denom = sqrtf((float)in_features) * (bits == 4 ? 8.0f : 128.0f);
    return (0.5f + u) / denom;
}

// kind 0: U(0.5, 1.5) (norm weights); kind 1: ~N(0,1)*0.02 (sum of four uniforms);
// kind 2: U(-1,1)/sqrt(n) with n passed in `index2` (plain T linear weights)
MC_HD float
value(uint64_t seed, uint32_t matrix_id, uint32_t index, int32_t kind, uint32_t n = 1)
{
    const uint64_t h = key(seed ^ 0xA11CEull, matrix_id, index, (uint32_t)kind);
    if (kind == 0) return 0.5f + (float)(uint32_t)((h >> 40) & 0xFFFFFF) * (1.0f / 16777216.0f);
    if (kind == 1) {
        const float s = (float)(uint32_t)(h & 0xFFFF) + (float)(uint32_t)((h >> 16) & 0xFFFF) +
                        (float)(uint32_t)((h >> 32) & 0xFFFF) + (float)(uint32_t)((h >> 48) & 0xFFFF);
        // variance of the sum of 4 U(0,65536) = 4*65536^2/12 -> sigma = 65536/sqrt(3)
        return (s - 131070.0f) * (0.02f * 1.7320508f / 65536.0f);
    }
    const float u = (float)(uint32_t)((h >> 40) & 0xFFFFFF) * (1.0f / 8388608.0f) - 1.0f;
    return u / sqrtf((float)n);
}

} // namespace mcsynth
