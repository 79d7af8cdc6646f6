// Device-side fill of synthetic weights straight into the packed HBM layouts (see gemv.h).
// `map` selects how a destination row maps to a (matrix, source row) pair:
//   0 plain: (m0, r)      1 qkv: rows [0,n0) -> (m0, r'), [n0,n0+n1) -> (m0+1, r'), rest -> (m0+2, ..)
//                              with r' the NATURAL row of packed row r (rotation partners adjacent:
//                              packed head*hd + 2j + e <-> natural head*hd + j + e*hd/2; hd = map >> 8)
//   2 interleave: row 2i -> (m0, i), row 2i+1 -> (m0+2, i)            (w1 | w3)
#include "common.h"
#include "synth.h"

using namespace mc;

__device__ __forceinline__ void
map_row(uint32_t map, uint32_t m0, uint32_t n0, uint32_t n1, uint32_t r, uint32_t& m, uint32_t& sr)
{
    if ((map & 0xFF) == 1) {
        const uint32_t hd = map >> 8;
        if (r < n0 + n1) {
            const uint32_t lr = r < n0 ? r : r - n0;
            const uint32_t head = lr / hd, w = lr % hd;
            m = r < n0 ? m0 : m0 + 1;
            sr = head * hd + (w >> 1) + (w & 1) * (hd / 2);
        } else { m = m0 + 2; sr = r - n0 - n1; }
    } else if (map == 2) {
        m = (r & 1) ? m0 + 2 : m0;
        sr = r >> 1;
    } else {
        m = m0; sr = r;
    }
}

// one thread per output dword: 8 nibbles (I4) or 4 bytes (I8)
extern "C" __global__ void
mc_synth_fill_q(uint32_t* w, uint64_t seed, uint32_t m0, uint32_t map, uint32_t n0, uint32_t n1,
                uint32_t rows, uint32_t in, int32_t bits)
{
    const uint32_t per = bits == 4 ? 8 : 4;
    const size_t dwords_per_row = in / per;
    const size_t total = (size_t)rows * dwords_per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / dwords_per_row), c0 = (uint32_t)(i % dwords_per_row) * per;
        uint32_t m, sr;
        map_row(map, m0, n0, n1, r, m, sr);
        uint32_t v = 0;
        if (bits == 4) {
            const uint32_t perm[8] = {0, 2, 4, 6, 1, 3, 5, 7};
#pragma unroll
            for (uint32_t p = 0; p < 8; p++)
                v |= (uint32_t)(mcsynth::weight(seed, m, sr, c0 + perm[p], 4) + 8) << (4 * p);
        } else {
#pragma unroll
            for (uint32_t p = 0; p < 4; p++)
                v |= (uint32_t)(uint8_t)(int8_t)mcsynth::weight(seed, m, sr, c0 + p, 8) << (8 * p);
        }
        w[i] = v;
    }
}

// scales in row quads [ceil(rows/4)][ngroups][4], stored bf16 (sbytes 2) or f32 (sbytes 4)
extern "C" __global__ void
mc_synth_fill_scales(void* s, uint64_t seed, uint32_t m0, uint32_t map, uint32_t n0, uint32_t n1,
                     uint32_t rows, uint32_t ngroups, uint32_t in, int32_t bits, int32_t sbytes)
{
    const size_t total = (size_t)rows * ngroups;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / ngroups), g = (uint32_t)(i % ngroups);
        uint32_t m, sr;
        map_row(map, m0, n0, n1, r, m, sr);
        const float v = mcsynth::scale(seed, m, sr, g, (int32_t)in, bits);
        // row-quad layout of gemv.h: [ceil(rows/4)][ngroups][4]
        const size_t o = ((size_t)(r / 4) * ngroups + g) * 4 + (r & 3);
        if (sbytes == 2) static_cast<bf16_t*>(s)[o] = f2bf(v);
        else static_cast<float*>(s)[o] = v;
    }
}

// T-typed values: out[i] = T(value(seed, m(row), index, kind)); `cols` = row length for the row map
extern "C" __global__ void
mc_synth_fill_T(void* out, uint64_t seed, uint32_t m0, uint32_t map, uint32_t n0, uint32_t n1,
                uint32_t rows, uint32_t cols, int32_t kind, int32_t tbytes)
{
    const size_t total = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / cols), c = (uint32_t)(i % cols);
        uint32_t m, sr;
        map_row(map, m0, n0, n1, r, m, sr);
        const float v = mcsynth::value(seed, m, sr * cols + c, kind, cols);
        if (tbytes == 2) static_cast<bf16_t*>(out)[i] = f2bf(v);
        else static_cast<float*>(out)[i] = v;
    }
}
