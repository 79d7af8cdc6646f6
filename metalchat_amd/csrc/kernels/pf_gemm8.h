// The prompt GEMM of long prompts (round 5): Y[M][N] = T(X[M][K] Wd[N][K]^T) on 256 x 256 output tiles, eight waves in two groups
// that alternate between the LDS and the matrix pipe.  Replaces nn::linear's bmm on len > 1 rows (include/metalchat/nn/linear.h:70-81,
// kernel/bmm.metal:25-82) for the quantised linears too: the W operand is Wd = T(T(q) T(s)) (kernel/mul.metal:78-82), dequantised
// into LDS once per K-slab by the waves that are not multiplying -- no copy of W in HBM.
//
// Why a new loop (round 4 measured it): the 128 x 128 / 256 x 128 tile loop of pf_gemm_big_body reaches 600-725 TFLOP/s WITH and
// WITHOUT its dequantisation (plain bfloat weights: 180 us for 512 x 28672 x 4096 against 168 with int4): every wave stages, waits,
// passes a barrier and multiplies in lockstep, so the matrix pipe idles while LDS is written and read.  Here:
//   * tile 256 (rows of X) x 256 (rows of W), K in tiles of 64; per wave 128 x 64 outputs = 8 x 4 accumulators of
//     v_mfma_f32_16x16x32_bf16 (128 registers).  Operands swapped -- A = W fragment, B = X fragment -- so that a lane ends up with four
//     CONSECUTIVE columns n of one row m: 8-byte stores, and the (w1, w3) pair of the fused w1|w3 matrix in one lane.
//   * waves 0-3 (group 0) and 4-7 (group 1) share the four SIMDs pairwise.  A PHASE is half a K tile (k-step of 32): one group reads its
//     twelve 16-byte fragments from LDS and issues the next loads while the other issues 32 MFMAs from the fragments it read in the
//     phase before; one s_barrier; roles swap.  The matrix pipe of every SIMD always has one wave multiplying.
//   * LDS: two images of a K tile, each four half tiles of 128 rows x 64 k (16 KiB: X rows 0-127 / 128-255, W rows 0-127 / 128-255), rows
//     of 128 bytes, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7) -- conflict-free for the fragment reads by the bank rule
//     of ds_read_b128 (MI355X_MICROARCH.md, LDS).  Filled by LDS-DMA (buffer_load ... lds, 16 bytes per lane): the destination is linear
//     per wave instruction (8 rows), the swizzle sits on the SOURCE address; rows past M or N read as zeros (the buffer's bounds).
//   * one half tile is staged per phase, by the group that is not multiplying, into the image whose last reader finished a phase
//     earlier; `s_waitcnt vmcnt(4)` at the end of every reading phase leaves ONE half tile of the wave in flight across the barrier;
//     a half tile is read three phases after it was requested at the earliest (~ 0.7 us).
//
//   phase p = 4 t + q of K tile t        group 0                         group 1                       staged (by the reading group)
//     q = 0                              read (t, k 0-31)                multiply (t - 1, k 32-63)      W rows 0-127 of tile t + 1
//     q = 1                              multiply (t, k 0-31)            read (t, k 0-31)               W rows 128-255 of tile t + 1
//     q = 2                              read (t, k 32-63)               multiply (t, k 0-31)           X rows 128-255 of tile t + 1
//     q = 3                              multiply (t, k 32-63)           read (t, k 32-63)              X rows 0-127 of tile t + 2
//   (group 0 reads X rows 0-127 only, last in q = 2: that half is free one phase before the other three.)
#pragma once

#include "common.h"

namespace mc {
namespace g8 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_char;

constexpr uint32_t BM = 256, BN = 256, BK = 64;
constexpr uint32_t HALF_BYTES = 128 * BK * 2;     // 16 KiB: 128 rows of 128 bytes
constexpr uint32_t IMG_BYTES = 4 * HALF_BYTES;    // X lo, X hi, W lo, W hi
constexpr uint32_t LDS_BYTES = 2 * IMG_BYTES;     // 128 KiB

enum { W_T = 0, W_I8 = 1, W_I4 = 2 };
enum { E_STORE = 0, E_RES = 1, E_PART = 2, E_ACT = 3 };

// which output tile a workgroup takes: XCD k (= workgroups with equal linear index % 8) owns a band of row tiles of X where there
// are enough of them, and walks the column tiles -- consecutive workgroups of an XCD share the W tile (prefill_kernels.hip pf_tile_of)
__device__ __forceinline__ void
tile_of(uint32_t& nt, uint32_t& mt)
{
    const uint32_t nx = gridDim.x, ny = gridDim.y;
    nt = blockIdx.x;
    mt = blockIdx.y;
    if ((nx * ny) % 8u) return;
    const uint32_t p = blockIdx.y * nx + blockIdx.x, k = p & 7u, j = p >> 3;
    if (ny % 8u == 0) {
        const uint32_t per = ny / 8u;
        mt = k * per + j % per;
        nt = j / per;
    } else if (8u % ny == 0 && nx % (8u / ny) == 0) {
        const uint32_t g = 8u / ny;
        mt = k / g;
        nt = j * g + k % g;
    }
}

struct args {
    const void* w;        // W_T: bfloat [N][K]; W_I4 / W_I8: the decode GEMV's packed rows
    const void* scales;   // row quads [ceil(N / 4)][K / group][4] bfloat
    const bf16_t* X;      // [M][K]
    void* Y;              // E_STORE / E_RES: bfloat [M][N]; E_PART: float [splits][M][N]; E_ACT: bfloat [M][N / 2]
    const void* res;      // E_RES: bfloat [M][N]; E_ACT: the table of exponentials (prefill_kernels.hip mc_exp_table_bfloat)
    uint32_t M, N, K, group;
};

template <int WF, int EPI, typename ActFn>
__device__ __forceinline__ void
body(const args& a, ActFn&& act)
{
    static_assert(WF == W_T, "quantised operands: pf_gemm8_q.h");
    __shared__ __attribute__((aligned(1024))) char lds_[LDS_BYTES];
    lds_char* const lds = (lds_char*)lds_;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t grp = wave >> 2, wq = wave & 3u; // group; X half = grp, W half = wq >> 1, its rows (wq & 1) * 64 ..
    const uint32_t l15 = lane & 15, lg = lane >> 4;
    uint32_t tile_n, tile_m;
    tile_of(tile_n, tile_m);
    const uint32_t n0 = tile_n * BN, m0 = tile_m * BM;
    const uint32_t M = a.M, N = a.N, K = a.K;
    // split-K (E_PART): workgroup z walks K tiles [z * tper, ...)
    const uint32_t TT = K / BK, tper = EPI == E_PART ? (TT + gridDim.z - 1) / gridDim.z : TT;
    const uint32_t tbeg = EPI == E_PART ? blockIdx.z * tper : 0u;
    const uint32_t T = tbeg < TT ? min(tper, TT - tbeg) : 0u;

    // ---- staging: a half tile = 16 wave instructions of 8 rows x 128 bytes; the four waves of the staging group take four each
    // (instruction i of wave wq: rows (4 wq + i) 8 .. + 7).  Lane l: row l >> 3, position l & 7 of the row, i.e. source chunk
    // (l & 7) ^ ((row >> 1) & 7) = (l & 7) ^ (((i & 1) * 4 + (l >> 4)) & 7).  The LDS-DMA is inline asm: hipcc then neither counts it
    // nor waits vmcnt(0) in front of every LDS read that might alias its destination (it did, with the builtin: the loop drained the
    // queue twelve times per phase) -- the waits are counted by hand below.
    auto rsrc_of = [](const void* p, size_t bytes) {
        const uint64_t v = (uint64_t)p;
        return u32x4{(uint32_t)v, (uint32_t)(v >> 32), (uint32_t)min(bytes, (size_t)0xFFFFFFFFu), 0x00020000u};
    };
    const u32x4 xrs = rsrc_of(a.X, (size_t)M * K * 2), wrs = rsrc_of(a.w, (size_t)N * K * 2);
    const uint32_t srow = wq * 32u + (lane >> 3);                       // row of instruction 0 inside the half tile
    const uint32_t sc0 = ((lane & 7u) ^ (lane >> 4)) * 16u;              // even instructions; odd ones: ^ 64
    const uint32_t xv0 = (m0 + srow) * K * 2u + sc0, wv0 = (n0 + srow) * K * 2u + sc0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds + wq * 4096u;
    // half: 0 / 1 = X rows 0-127 / 128-255, 2 / 3 = W rows 0-127 / 128-255 (wave-uniform); t: K tile of this workgroup's range
    auto stage = [&](uint32_t img, uint32_t half, uint32_t t) {
        // (a tile past the end of the K range: the last one again -- what it writes is never read, and the loop keeps its counted waits)
        const uint32_t k0 = (tbeg + (t < T ? t : T - 1u)) * BK;
        const uint32_t dst = lds0 + img * IMG_BYTES + half * HALF_BYTES;
        const bool isw = half >= 2u;
        const uint32_t base = (isw ? wv0 : xv0) + ((half & 1u) * 128u) * K * 2u + k0 * 2u;
        const u32x4 rs = isw ? wrs : xrs; // (scalar selects: the descriptor stays in SGPRs)
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            const uint32_t vo = (base + i * 8u * K * 2u) ^ ((i & 1u) * 64u);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "s"(dst + i * 1024u), "v"(vo), "s"(rs)
                         : "memory");
        }
    };

    // ---- fragments: 16 bytes of row (16 j + l15) of a half tile, k = 32 ks + 8 lg .. + 7: chunk 4 ks + lg at position ^ (l15 >> 1)
    const uint32_t f0 = l15 * 128u + ((lg ^ (l15 >> 1)) & 7u) * 16u;
    const uint32_t f1 = l15 * 128u + (((4u + lg) ^ (l15 >> 1)) & 7u) * 16u;
    const uint32_t wbase = (2u + (wq >> 1)) * HALF_BYTES + (wq & 1u) * 8192u, xbase = grp * HALF_BYTES;
    u32x4 wf[4], xf[8];
    auto read_frags = [&](uint32_t img, uint32_t ks) {
        lds_char* p = lds + img * IMG_BYTES + (ks ? f1 : f0);
#pragma unroll
        for (int j = 0; j < 4; j++) wf[j] = *(const __attribute__((address_space(3))) u32x4*)(p + wbase + j * 2048);
#pragma unroll
        for (int i = 0; i < 8; i++) xf[i] = *(const __attribute__((address_space(3))) u32x4*)(p + xbase + i * 2048);
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto multiply = [&] {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j]), __builtin_bit_cast(bf16x8, xf[i]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // end of a phase in which this wave read and staged: its fragment reads done (the half tile may be overwritten a phase later),
    // all but its youngest half tile landed
    auto end_read = [&] {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto end_mul = [&] {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    if (T != 0) {
        // ---- prologue: tile 0 whole and X lo of tile 1, every wave its quarter of each half tile (both groups issue the same
        // instructions here: the same bytes twice, once per launch)
        stage(0, 2, 0);
        stage(0, 3, 0);
        stage(0, 0, 0);
        stage(0, 1, 0);
        stage(1, 0, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); // tile 0 landed (this wave's share), X lo of tile 1 in flight
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ONE instruction stream for both groups; group 1 runs it one barrier -- one phase -- behind group 0 (and group 0 passes one
        // more barrier at the end), so that wherever group 0 multiplies group 1 reads and stages, and the other way round.  What a
        // wave stages in its two reading phases of tile t is the table's column for ITS group.
        if (grp != 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        for (uint32_t t = 0; t < T; t++) {
            const uint32_t img = t & 1u, nimg = img ^ 1u;
            read_frags(img, 0);
            stage(nimg, 2u + grp, t + 1);                      // group 0: W lo, group 1: W hi of tile t + 1
            end_read();
            multiply();
            end_mul();
            read_frags(img, 1);
            stage(grp ? img : nimg, grp ? 0u : 1u, t + 1 + grp); // group 0: X hi of tile t + 1, group 1: X lo of tile t + 2
            end_read();
            multiply();
            end_mul();
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- epilogue: lane (l15, lg) holds, in acc[i][j][r], row m = 16 i + l15 of the wave's 128, column n = 16 j + 4 lg + r of its 64
    const uint32_t mw = m0 + grp * 128u + l15, nw = n0 + wq * 64u + lg * 4u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t m = mw + 16u * i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t n = nw + 16u * j;
            if (n >= N) continue; // (N is a multiple of 4: the host's condition)
            const f32x4 v = acc[i][j];
            if constexpr (EPI == E_PART) {
                *reinterpret_cast<f32x4*>(static_cast<float*>(a.Y) + ((size_t)blockIdx.z * M + m) * N + n) = v;
            } else if constexpr (EPI == E_ACT) {
                // columns (2 c, 2 c + 1) of the fused w1|w3 output are the pair (w1 x, w3 x) of output column c
                const float o0 = act(BF::rt(v[0]), BF::rt(v[1])), o1 = act(BF::rt(v[2]), BF::rt(v[3]));
                *reinterpret_cast<uint32_t*>(static_cast<bf16_t*>(a.Y) + (size_t)m * (N / 2) + (n >> 1)) = pack_bf16x2(o0, o1);
            } else {
                float o[4] = {BF::rt(v[0]), BF::rt(v[1]), BF::rt(v[2]), BF::rt(v[3])};
                if constexpr (EPI == E_RES) {
                    const u32x2 r = *reinterpret_cast<const u32x2*>(static_cast<const bf16_t*>(a.res) + (size_t)m * N + n);
                    o[0] = __uint_as_float(r.x << 16) + o[0];
                    o[1] = __uint_as_float(r.x & 0xFFFF0000u) + o[1];
                    o[2] = __uint_as_float(r.y << 16) + o[2];
                    o[3] = __uint_as_float(r.y & 0xFFFF0000u) + o[3];
                }
                *reinterpret_cast<u32x2*>(static_cast<bf16_t*>(a.Y) + (size_t)m * N + n) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            }
        }
    }
}

} // namespace g8
} // namespace mc
