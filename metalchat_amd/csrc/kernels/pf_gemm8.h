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
//   * one half tile is staged per phase, by the group that is not multiplying, into a ring of half-tile slots; a counted
//     `s_waitcnt vmcnt(N)` at the end of every reading phase leaves the wave's youngest requests in flight across the barrier; a half
//     tile is read 3 (8 slots) or 5 (10 slots: all 160 KiB) phases after it was requested at the earliest.
//
//   phase p = 4 t + q of K tile t        group 0                         group 1                       staged (by the reading group)
//     q = 0                              read (t, k 0-31)                multiply (t - 1, k 32-63)      W rows 0-127 of tile t + 1
//     q = 1                              multiply (t, k 0-31)            read (t, k 0-31)               W rows 128-255 of tile t + 1
//     q = 2                              read (t, k 32-63)               multiply (t, k 0-31)           X rows 128-255 of tile t + 1
//     q = 3                              multiply (t, k 32-63)           read (t, k 32-63)              X rows 0-127 of tile t + 2
//   (group 0 reads X rows 0-127 only, last in q = 2: that half is free one phase before the other three.)
#pragma once

#include "common.h"

#include <type_traits>

namespace mc {
namespace g8 {

// byte k of a dword -> float in one full-rate instruction (hipcc builds v_bfe_u32 + v_cvt_f32_u32 otherwise)
template <int K_>
__device__ __forceinline__ float
ubyte_f32(uint32_t v)
{
    float d;
    if (K_ == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(d) : "v"(v));
    if (K_ == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(d) : "v"(v));
    if (K_ == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(d) : "v"(v));
    if (K_ == 3) asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(d) : "v"(v));
    return d;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_char;

constexpr uint32_t BN = 256, BK = 64; // (rows of X per tile: the body's BM, 256 or 128)
constexpr uint32_t HALF_BYTES = 128 * BK * 2;     // 16 KiB: 128 rows of 128 bytes
constexpr uint32_t IMG_BYTES = 4 * HALF_BYTES;    // X lo, X hi, W lo, W hi
constexpr uint32_t LDS_BYTES = 2 * IMG_BYTES;     // 128 KiB

#ifndef MC_G8_INTERLEAVE
#define MC_G8_INTERLEAVE 1 // the multiplying phase of the quantised loop: MFMAs and the dequantisation's vector instructions interleaved by sched_group_barrier
#endif
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

// NS: half-tile slots of the LDS ring (8 = two images of a K tile, 128 KiB; 10 = all 160 KiB: every request goes out two phases
// earlier).  DIAG (lab builds): 1 = nothing is staged inside the loop (what the LDS reads, the MFMAs and the barriers take alone),
// 2 = no MFMAs (what the staging and the reads take alone).
// BM: rows of X per tile.  256: the tile above.  128 (mc_pf_gemm8h_*): every wave 64 x 64 outputs, the X half tiles 64 rows -- for the
// launches whose 256-row tiles would be too few for the chip: twice the workgroups before K is split, i.e. half the fp32 partial sums
// written and read back (a 512-row prompt of Llama-3-8B moved 368 MB of them per block with 256-row tiles).
// MF: the MFMA -- 16: v_mfma_f32_16x16x32_bf16 (32 per phase and wave), 32: v_mfma_f32_32x32x16_bf16 (16 per phase: the same matrix-pipe
// time from half the instructions, i.e. half the issue-port cycles the MFMAs hold -- what the quantised loop's dequantisation competes for).
template <int WF, int EPI, int NS = 8, int DIAG = 0, int BM = 256, int MF = 16, typename ActFn>
__device__ __forceinline__ void
body(const args& a, ActFn&& act)
{
    static_assert(NS == 8 || NS == 10, "slots");
    static_assert(WF == W_T || NS == 8, "the quantised loop's schedule is written for two images");
    static_assert(BM == 256 || (BM == 128 && NS == 8), "rows of X per tile");
    static_assert(MF == 16 || MF == 32, "MFMA shape");
    constexpr int MI = BM / (2 * MF);        // MF-row tiles of X per wave (its BM / 2 rows)
    constexpr int NI = 64 / MF;              // MF-row tiles of W per wave (its 64 rows)
    constexpr int KS = MF == 16 ? 1 : 2;     // MFMA k-steps per 32-deep phase
    constexpr int AR = MF == 16 ? 4 : 16;    // accumulator registers per tile
    typedef float accv __attribute__((ext_vector_type(AR)));
    constexpr uint32_t NXI = BM / 64;        // LDS-DMAs per wave of an X half tile (BM / 2 rows)
    constexpr int D = NS - 8;                  // phases a request goes out earlier than the two-image schedule's
    constexpr int LEAD = 1 + D;                // events of tile 1 the prologue requests behind tile 0 (the loop's phase p requests event 4 + LEAD + p)
    __shared__ __attribute__((aligned(1024))) char lds_[NS * HALF_BYTES];
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

    // ---- staging.  EVENTS: half tile c (0 = X rows 0-127, 1 = W rows 0-127, 2 = W rows 128-255, 3 = X rows 128-255) of K tile u is
    // event 4 u + c and lives in ring slot (4 u + c) % NS.  Tile 0 and the first LEAD events of tile 1 are requested in the prologue;
    // phase p of the loop requests event 4 + LEAD + p = p + NS - 3, by the group that reads in that phase: event e goes out in phase
    // e - NS + 3 and is first read in phase >= 4 (e / 4), i.e. 3 + D phases later at the earliest (the W hi and X hi events).  A slot is
    // free again when its event's last reader is done: X lo of tile u after phase 4 u + 2, the other three after 4 u + 3 -- and event
    // e + NS, the slot's next tenant, is requested in phase e + 3: 4 u + 3 for the X lo of tile u (e = 4 u), >= 4 u + 4 for the others.
    // A half tile = 16 wave instructions of 8 rows x 128 bytes; the four waves of the staging group take four each (instruction i
    // of wave wq: rows (4 wq + i) 8 .. + 7).  Lane l: row l >> 3, position l & 7 of the row, i.e. source chunk
    // (l & 7) ^ ((row >> 1) & 7) = (l & 7) ^ (((i & 1) * 4 + (l >> 4)) & 7).  The LDS-DMA is inline asm: hipcc then neither counts it
    // nor waits vmcnt(0) in front of every LDS read that might alias its destination (it did, with the builtin: the loop drained the
    // queue twelve times per phase) -- the waits are counted by hand below.
    auto rsrc_of = [](const void* p, size_t bytes) {
        const uint64_t v = (uint64_t)p;
        return u32x4{(uint32_t)v, (uint32_t)(v >> 32), (uint32_t)min(bytes, (size_t)0xFFFFFFFFu), 0x00020000u};
    };
    const u32x4 xrs = rsrc_of(a.X, (size_t)M * K * 2), wrs = rsrc_of(a.w, (size_t)N * K * 2);
    const uint32_t srow = wq * 32u + (lane >> 3);                       // row of instruction 0 inside a W half tile (X: wq BM / 8 + ...)
    const uint32_t sc0 = ((lane & 7u) ^ (lane >> 4)) * 16u;              // even instructions; odd ones: ^ 64
    const uint32_t xv0 = (m0 + wq * (BM / 8u) + (lane >> 3)) * K * 2u + sc0, wv0 = (n0 + srow) * K * 2u + sc0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds + wq * 4096u, lds0x = (uint32_t)(uintptr_t)lds + wq * (NXI * 1024u);
    // event e (wave-uniform): tile e / 4 of this workgroup's K range, half tile e % 4
    auto stage = [&](uint32_t e) {
        const uint32_t u = e >> 2, c = e & 3u;
        // (a tile past the end of the K range: the last one again -- what it writes is never read, and the loop keeps its counted waits)
        const uint32_t k0 = (tbeg + (u < T ? u : T - 1u)) * BK;
        const bool isw = c == 1u || c == 2u;
        const uint32_t dst = (isw ? lds0 : lds0x) + (e % (uint32_t)NS) * HALF_BYTES;
        const uint32_t hi = c >> 1; // (the upper half of the operand's rows: c = 2, 3)
        const uint32_t base = (isw ? wv0 : xv0) + (hi * (isw ? 128u : BM / 2u)) * K * 2u + k0 * 2u;
        const u32x4 rs = isw ? wrs : xrs; // (scalar selects: the descriptor stays in SGPRs)
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            if (BM == 128 && i >= NXI && !isw) break; // (wave-uniform: an X half tile of 64 rows is two instructions per wave)
            const uint32_t vo = (base + i * 8u * K * 2u) ^ ((i & 1u) * 64u);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "s"(dst + i * 1024u), "v"(vo), "s"(rs)
                         : "memory");
        }
    };

    // ---- fragments: 16 bytes of row (16 j + l15) of a half tile, k = 32 ks + 8 lg .. + 7: chunk 4 ks + lg at position ^ (l15 >> 1)
    // (32 x 32 x 16: lane (r = lane & 31, h = lane >> 5) holds row r, k = 16 kk + 8 h .. + 7 of k-step kk: chunk 4 ks + 2 kk + h.  The same
    //  positions are conflict-free: rows 0-3, 12-15 and 20-27 of one chunk -- a lane group of ds_read_b128 -- have (r >> 1) & 7 = 0 0 1 1 6 6 7 7
    //  2 2 3 3 4 4 5 5 in both row parities)
    const uint32_t l31 = lane & 31, lh = lane >> 5;
    const uint32_t frow = MF == 16 ? l15 : l31, fsw = frow >> 1;
    auto fpos = [&](uint32_t ks, uint32_t kk) { // byte offset of the lane's fragment of k-step (ks, kk) inside its tile of MF rows
        const uint32_t chunk = MF == 16 ? 4u * ks + lg : 4u * ks + 2u * kk + lh;
        return frow * 128u + ((chunk ^ fsw) & 7u) * 16u;
    };
    const uint32_t f0 = fpos(0, 0), f1 = fpos(1, 0), f0b = fpos(0, 1), f1b = fpos(1, 1);
    const uint32_t wsub = (wq & 1u) * 8192u;
    u32x4 wf[NI][KS], xf[MI][KS];
    // tile t: the W half of this wave is event 4 t + 1 + (wq >> 1), its X half event 4 t + 3 grp
    auto read_frags = [&](uint32_t t, uint32_t ks) {
        const uint32_t ws = (4u * t + 1u + (wq >> 1)) % (uint32_t)NS, xs = (4u * t + 3u * grp) % (uint32_t)NS;
        lds_char* pw = lds + ws * HALF_BYTES + wsub;
        lds_char* px = lds + xs * HALF_BYTES;
#pragma unroll
        for (int kk = 0; kk < KS; kk++) {
            const uint32_t f = ks ? (kk ? f1b : f1) : (kk ? f0b : f0);
#pragma unroll
            for (int j = 0; j < NI; j++) wf[j][kk] = *(const __attribute__((address_space(3))) u32x4*)(pw + f + j * (MF * 128));
#pragma unroll
            for (int i = 0; i < MI; i++) xf[i][kk] = *(const __attribute__((address_space(3))) u32x4*)(px + f + i * (MF * 128));
        }
    };
    accv acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++)
#pragma unroll
            for (int e = 0; e < AR; e++) acc[i][j][e] = 0.0f;
    auto multiply = [&](bool prio = true) {
        if constexpr (DIAG == 2) {
#pragma unroll
            for (int i = 0; i < MI; i++) asm volatile("" ::"v"(xf[i][0]), "v"(xf[i][KS - 1]));
#pragma unroll
            for (int j = 0; j < NI; j++) asm volatile("" ::"v"(wf[j][0]), "v"(wf[j][KS - 1]));
            return;
        }
        if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < KS; kk++)
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NI; j++) {
                    if constexpr (MF == 16)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j][kk]), __builtin_bit_cast(bf16x8, xf[i][kk]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[j][kk]), __builtin_bit_cast(bf16x8, xf[i][kk]), acc[i][j], 0, 0, 0);
                }
        if (prio) __builtin_amdgcn_s_setprio(0);
    };
    // end of a phase in which this wave read and staged: its fragment reads done (the slot may be overwritten a phase later), all
    // but its youngest 1 + D / 2 half tiles landed
    // (BM = 128, two images: the youngest half tile is a W one -- four instructions -- behind the reading step of k 0-31 and an X one --
    //  two -- behind that of k 32-63)
    auto end_read = [&](uint32_t ks) {
        if constexpr (D != 0) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if constexpr (BM == 256) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else {
            if (ks == 0) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto end_mul = [&] {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (WF != W_T) {
        // ---- quantised W (int4 / int8 rows of the decode GEMV's layout): X travels as above, in QUARTERS (64 rows: two LDS-DMAs per
        // wave and phase); a W quarter (64 rows x 64 k = the rows ONE wave position wq multiplies) is loaded to registers -- 8 (int4) or
        // 16 (int8) bytes per lane: row lane / 4 of the wave's 16, weights 16 (lane % 4) .. + 15 of the tile -- by the group that reads,
        // and two of its reading phases later dequantised EXACTLY (Wd = T(T(q) T(s)), kernel/mul.metal:78-82: one conversion, one fma
        // and half a pack per weight; (n - 8) s and q s are exact in fp32) and written into the W slot with two 16-byte LDS stores: ~ 50
        // vector instructions per wave and phase, issued by the group that is NOT multiplying.  Per phase p = 4 t + 2 ks + grp:
        //     X quarter j = (p + 5) % 4 of tile (p + 5) / 4 (X lo rows 0-63, 64-127, X hi rows 0-63, 64-127) requested;
        //     W quarter r = p % 4 of tile p / 4 + 1 dequantised and written (its slot's last reader -- tile p / 4 - 1 -- finished a phase
        //     ago at the latest; its first reader comes in phase 4 (p / 4 + 1) at the earliest); W quarter (p + 2) % 4 of tile (p + 2) / 4 + 1
        //     requested.  One wait per phase: `vmcnt(2)` behind the two new LDS-DMAs = everything of the previous reading phase landed.
        const uint32_t K2 = WF == W_I4 ? K / 2u : K;                       // bytes per W row
        const uint32_t RUNB = WF == W_I4 ? 8u : 16u;                       // bytes of a lane's 16 weights
        const uint32_t ngroups = a.group ? K / a.group : 1u, glog = a.group ? 31u - (uint32_t)__builtin_clz(a.group) : 31u;
        const u32x4 qrs = rsrc_of(a.w, (size_t)N * K2), srs = rsrc_of(a.scales, (size_t)((N + 3u) / 4u) * ngroups * 8u);
        const uint32_t wrow16 = wq * 16u + (lane >> 2);                    // the lane's row inside a W quarter
        const uint32_t qv0 = (n0 + wrow16) * K2 + (lane & 3u) * RUNB;      // + (64 r) K2 + k0 bytes
        // (+ (16 r ngroups + g) 8 per quarter; g = group of the tile's first weight, the lane's 16-run may lie in a later one when group < 64)
        const uint32_t sv0 = (((n0 >> 2) + 4u * wq + (lane >> 4)) * ngroups * 4u + ((lane >> 2) & 3u)) * 2u +
                             (a.group ? (((lane & 3u) * 16u) >> glog) * 8u : 0u);
        const uint32_t ww0 = wrow16 * 128u + (((2u * (lane & 3u)) ^ ((lane >> 3) & 7u)) * 16u); // first chunk; the second: ^ 16
        constexpr uint32_t NXQ = BM / 128; // LDS-DMAs per wave of an X quarter
        typedef typename std::conditional<WF == W_I4, u32x2, u32x4>::type wraw_t;
        wraw_t wA, wB;
        uint32_t sA, sB;
        auto stage_xq = [&](uint32_t u, uint32_t j) { // X quarter j of tile u
            const uint32_t c = (j >> 1) ? 3u : 0u, qh = j & 1u;
            const uint32_t k0 = (tbeg + (u < T ? u : T - 1u)) * BK;
            // (a quarter = BM / 4 rows; a wave's share of it: BM / 16 rows = NXQ instructions, NXQ KiB)
            const uint32_t dst = (uint32_t)(uintptr_t)lds + wq * (NXQ * 1024u) + ((4u * u + c) % 8u) * HALF_BYTES + qh * (BM * 32u);
            const uint32_t base = (m0 + wq * (BM / 16u) + (lane >> 3)) * K * 2u + sc0 + ((j >> 1) * (BM / 2u) + qh * (BM / 4u)) * K * 2u + k0 * 2u;
#pragma unroll
            for (uint32_t i = 0; i < NXQ; i++) {
                // (the source chunk's swizzle (row >> 1) & 7: + 4 on every second group of 8 rows -- instruction 1 of a wave, and with
                //  32-row quarters, BM = 128, the odd waves' only instruction)
                const uint32_t vo = (base + i * 8u * K * 2u) ^ (((i + (BM == 128 ? wq : 0u)) & 1u) * 64u);
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "s"(dst + i * 1024u), "v"(vo), "s"(xrs)
                             : "memory");
            }
        };
        // W quarter wp % 4 of tile wp / 4 + 1 (the quarter that is WRITTEN in phase wp) into a register set (asm: counted by hand)
        auto load_wq = [&](uint32_t u, uint32_t r, wraw_t& wraw, uint32_t& sraw) {
            const uint32_t k0 = (tbeg + (u < T ? u : T - 1u)) * BK;
            const uint32_t vo = qv0 + (64u * r) * K2 + (WF == W_I4 ? k0 / 2u : k0);
            const uint32_t so = sv0 + (16u * r * ngroups + (a.group ? k0 >> glog : 0u)) * 8u;
            if constexpr (WF == W_I4) asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(wraw) : "v"(vo), "s"(qrs) : "memory");
            else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(wraw) : "v"(vo), "s"(qrs) : "memory");
            asm volatile("buffer_load_ushort %0, %1, %2, 0 offen" : "=v"(sraw) : "v"(so), "s"(srs) : "memory");
        };
        auto write_wq = [&](uint32_t u, uint32_t r, const wraw_t& wraw, uint32_t sraw) { // -> the W slot of tile u, rows 64 (r & 1) .. of half r >> 1
            const float sc = __uint_as_float(sraw << 16);
            uint32_t o[8];
            if constexpr (WF == W_I4) {
                const float c8 = -8.0f * sc;
                const uint32_t v[2] = {wraw.x, wraw.y};
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    // nibble p of a dword = weight {0,2,4,6,1,3,5,7}[p] of its 8-run: byte b of lo / hi = nibble 2 b / 2 b + 1
                    const uint32_t lo = v[d] & 0x0F0F0F0Fu, hi = (v[d] >> 4) & 0x0F0F0F0Fu;
                    o[4 * d + 0] = pack_bf16x2(__builtin_fmaf(ubyte_f32<0>(lo), sc, c8), __builtin_fmaf(ubyte_f32<2>(lo), sc, c8));
                    o[4 * d + 1] = pack_bf16x2(__builtin_fmaf(ubyte_f32<0>(hi), sc, c8), __builtin_fmaf(ubyte_f32<2>(hi), sc, c8));
                    o[4 * d + 2] = pack_bf16x2(__builtin_fmaf(ubyte_f32<1>(lo), sc, c8), __builtin_fmaf(ubyte_f32<3>(lo), sc, c8));
                    o[4 * d + 3] = pack_bf16x2(__builtin_fmaf(ubyte_f32<1>(hi), sc, c8), __builtin_fmaf(ubyte_f32<3>(hi), sc, c8));
                }
            } else {
                // a byte b = q as int8: b ^ 0x80 = q + 128 as an unsigned byte, (q + 128) s - 128 s = q s exactly (<= 16 significant bits)
                const float c128 = -128.0f * sc;
                const uint32_t v[4] = {wraw.x ^ 0x80808080u, wraw.y ^ 0x80808080u, wraw.z ^ 0x80808080u, wraw.w ^ 0x80808080u};
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    o[2 * d + 0] = pack_bf16x2(__builtin_fmaf(ubyte_f32<0>(v[d]), sc, c128), __builtin_fmaf(ubyte_f32<1>(v[d]), sc, c128));
                    o[2 * d + 1] = pack_bf16x2(__builtin_fmaf(ubyte_f32<2>(v[d]), sc, c128), __builtin_fmaf(ubyte_f32<3>(v[d]), sc, c128));
                }
            }
            lds_char* dst = lds + ((4u * u + 1u + (r >> 1)) % 8u) * HALF_BYTES + (r & 1u) * 8192u + ww0;
            *(__attribute__((address_space(3))) u32x4*)dst = u32x4{o[0], o[1], o[2], o[3]};
            *(__attribute__((address_space(3))) u32x4*)((lds_char*)((uint32_t)(uintptr_t)dst ^ 16u)) = u32x4{o[4], o[5], o[6], o[7]};
        };
        if (T != 0) {
            // ---- prologue: X of tile 0 (every wave its share of the four quarters: both groups issue the same instructions), W of
            // tile 0 (group 0: quarters 0 and 2, group 1: 1 and 3), the quarter of phase 0 -- in which nobody multiplies yet -- by group
            // 1, then what phase -1 would have requested: an X quarter, and the W quarter this wave writes in its first multiplying phase
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) stage_xq(0u, j);
            load_wq(0u, grp, wA, sA);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(wA), "+v"(sA)::"memory");
            write_wq(0u, grp, wA, sA);
            __builtin_amdgcn_sched_barrier(0);
            load_wq(0u, 2u + grp, wA, sA);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(wA), "+v"(sA)::"memory");
            write_wq(0u, 2u + grp, wA, sA);
            __builtin_amdgcn_sched_barrier(0);
            if (grp != 0) {
                load_wq(1u, 0u, wA, sA);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(wA), "+v"(sA)::"memory");
                write_wq(1u, 0u, wA, sA);
            }
            __builtin_amdgcn_sched_barrier(0);
            stage_xq(1u, 0u);
            load_wq(1u, 1u + grp, wB, sB);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (grp != 0) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // a reading step of the stream: phase p.  `ld`: the register set loaded here (for the quarter written in phase p + 3),
            // `keep`: the other one, loaded in the wave's previous reading phase and consumed in the multiplying phase that follows
            auto read_step = [&](uint32_t t, uint32_t ks, wraw_t& wld, uint32_t& sld, wraw_t& wkeep, uint32_t& skeep) {
                const uint32_t p = 4u * t + 2u * ks + grp;
                read_frags(t, ks);
                stage_xq((p + 5u) >> 2, (p + 5u) & 3u);
                load_wq(((p + 3u) >> 2) + 1u, (p + 3u) & 3u, wld, sld);
                // the previous reading phase's requests landed (in flight: this phase's NXQ LDS-DMAs and two loads)
                if constexpr (BM == 256) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" : "+v"(wkeep), "+v"(skeep)::"memory");
                else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" : "+v"(wkeep), "+v"(skeep)::"memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            };
            // a multiplying step: phase p + 1 behind reading step p; the W quarter of this phase from `w`, `sc`
            auto mul_step = [&](uint32_t t, uint32_t ks, const wraw_t& w, uint32_t sc) {
                const uint32_t p = 4u * t + 2u * ks + grp + 1u;
                __builtin_amdgcn_s_setprio(1);
                write_wq((p >> 2) + 1u, p & 3u, w, sc);
                multiply(false); // (s_setprio is a scheduling fence: with it around the MFMAs alone they stay behind all the vector work)
#if MC_G8_INTERLEAVE
                // one MFMA, then two of the dequantisation's vector instructions, ...
#pragma unroll
                for (int i = 0; i < MI * NI * KS; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, (BM == 256 ? 2 : 4) * (MF == 16 ? 1 : 2), 0);
                }
#endif
                __builtin_amdgcn_s_setprio(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            };
            for (uint32_t t = 0; t < T; t++) {
                read_step(t, 0, wA, sA, wB, sB);
                mul_step(t, 0, wB, sB);
                read_step(t, 1, wB, sB, wA, sA);
                mul_step(t, 1, wA, sA);
            }
            if (grp == 0) __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(wA), "+v"(sA), "+v"(wB), "+v"(sB)::"memory");
        }
    } else
    if (T != 0) {
        // ---- prologue: tile 0 and the first LEAD events of tile 1, every wave its quarter of each half tile (both groups issue the
        // same instructions here: the same bytes twice, once per launch)
#pragma unroll
        for (uint32_t e = 0; e < 4u + LEAD; e++) stage(e);
        // tile 0 landed (this wave's share); in flight: X lo of tile 1 (D = 0), + W lo and W hi of tile 1 (D = 2)
        if constexpr (D != 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr (BM == 256) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ONE instruction stream for both groups; group 1 runs it one barrier -- one phase -- behind group 0 (and group 0 passes one
        // more barrier at the end), so that wherever group 0 multiplies group 1 reads and stages, and the other way round: the
        // stream's reading step (t, ks) is phase 4 t + 2 ks for group 0 and 4 t + 2 ks + 1 for group 1.
        if (grp != 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        for (uint32_t t = 0; t < T; t++) {
            read_frags(t, 0);
            if (DIAG != 1) stage(4u + LEAD + 4u * t + grp);
            end_read(0);
            multiply();
            end_mul();
            read_frags(t, 1);
            if (DIAG != 1) stage(4u + LEAD + 4u * t + 2u + grp);
            end_read(1);
            multiply();
            end_mul();
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- epilogue.  16 x 16 x 32: lane (l15, lg) holds, in acc[i][j][r], row m = 16 i + l15 of the wave's rows, column n = 16 j + 4 lg + r of its
    // 64.  32 x 32 x 16: lane (l31, lh) holds, in acc[i][j][4 g + r], row 32 i + l31, column 32 j + 8 g + 4 lh + r.  Either way four consecutive
    // columns of one row per group of four registers.
    const uint32_t mw = m0 + grp * (BM / 2u) + frow, nw = n0 + wq * 64u + (MF == 16 ? lg : lh) * 4u;
#pragma unroll
    for (int i = 0; i < MI; i++) {
        const uint32_t m = mw + (uint32_t)MF * i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NI; j++)
#pragma unroll
            for (int g = 0; g < AR / 4; g++) {
                const uint32_t n = nw + (uint32_t)MF * j + 8u * g;
                if (n >= N) continue; // (N is a multiple of 4: the host's condition)
                const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
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
