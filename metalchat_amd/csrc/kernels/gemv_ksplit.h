// Long rows, few of them: the K range of a row pair cut over FOUR waves of one workgroup (round 6: mc_gemv_i4_bfloat_lin12k4_p0_e{0,1}).
//
// Why: Gemma-7B's w2 (include/metalchat/nn/transformer.h:59, K = 24576: rows of 12 KiB, 1536 row pairs) gives the linear-order kernels
// (gemv.h) one pair per wave on 192 of the 256 CUs; a wave then streams 24 KiB on its own -- 24 dependent tile round trips with 2 KiB in
// flight -- and the launch lasts 10.9 us for 38.9 MB (0.45 of 8 TB/s; "bound by what ONE wave streams").  Here every CU has a
// workgroup and every wave 18 KiB: workgroup g owns pairs 6 g .. 6 g + 5, wave w the K quarter w % 4 (chunks 3 q .. 3 q + 2 of every row:
// its 12 KB of the activation row sit in 48 registers for the whole launch) of the three pairs of its half (w / 4); the four quarter sums of
// a row meet in LDS and are added IN QUARTER ORDER by the lane that finishes the row.  Measured: ~ 10.0 us per launch, 677 -> 689 tokens/s on the
// Gemma-7B shapes (a CU still takes in the whole 48 KB row next to its 152 KB of weights: 200 KB against 251 KB on 192 CUs before).
//
// Numerics: the arithmetic per weight is the linear-order kernels' (mac4b_n: Wd = T(T(q) T(s)) exactly, products accumulated by the 4x4x4
// MFMA chunk by chunk, the lane's own element, one wave sum times 2^37); what changes is the ORDER of a row's fp32 additions: four chains of
// three chunks and ((s0 + s1) + s2) + s3 instead of one chain of twelve -- the same class of difference as `_lin3s_` and the int8 matrix-pipe
// kernels against the classic family: within one bf16 step of the oracle on <= 1 % of the outputs (tests/test_lin_kernels_gpu.py).
#pragma once

#include "gemv.h"

namespace mc {
namespace gemv {

// NCH = KiB per row (K = 2048 NCH); KS = 4 quarters of NCH / 4 chunks; eight waves; pairs per workgroup <= 8 (the host: out_rows / 2 a whole
// multiple of the grid or the remainder dealt to the first workgroups, as lin_deal does)
template <int NCH, int EPI>
__device__ __forceinline__ void
body_ksplit4(const void* __restrict__ wp, const void* __restrict__ sp, const void* __restrict__ xp, void* __restrict__ yp, const void* __restrict__ resp,
             uint32_t out_rows, uint32_t group)
{
    static_assert(NCH % 4 == 0 && (EPI == EPI_STORE || EPI == EPI_RESID), "whole chunks per quarter; plain store or residual add");
    constexpr int QC = NCH / 4;                       // chunks (KiB of weights) of a row per quarter
    constexpr uint32_t K = 2048u * NCH, ROWB = 1024u * NCH, CHUNK_LDS = 2048 * 2 / 16 * 17;
    constexpr int HP = 4;                             // pairs per half of a workgroup, at most
    asm volatile("" ::"s"(wp), "s"(sp), "s"(xp), "s"(yp), "s"(resp), "s"(out_rows), "s"(group), "s"(gridDim.x)); // one round of scalar loads (gemv.h)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;                                                         // the row, padded: packet p in slot p + p / 16
    float* part = reinterpret_cast<float*>(smem + (size_t)NCH * CHUNK_LDS); // [2 HP pairs][2 rows][4 quarters]
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t q = wave & 3u, half = wave >> 2;
    typedef uint32_t rowv4 __attribute__((ext_vector_type(4)));
    // ---- the wave pair (q, q + 4) stages quarter q of the row: 256 QC packets over 128 threads, requested first
    constexpr int NXP = (256 * QC + 127) / 128;
    rowv4 xr[NXP];
    {
        const uint32_t t2 = (half << 6) | lane; // 0 .. 127
#pragma unroll
        for (int i = 0; i < NXP; i++) {
            const uint32_t p = 256u * QC * q + min(t2 + 128u * (uint32_t)i, 256u * QC - 1u);
            xr[i] = reinterpret_cast<const rowv4*>(xp)[p];
        }
    }
    asm volatile("s_barrier" ::: "memory"); // (the row's requests stay ahead of the weight requests in the CU's memory pipe: gemv.h)
    // ---- the deal: the launch's pairs in equal contiguous shares per workgroup (a remainder to the first ones), a workgroup's share in two halves
    const uint32_t NP = out_rows / 2, G = gridDim.x, wq_ = NP / G, wrem = NP - wq_ * G;
    const uint32_t nb = wq_ + (blockIdx.x < wrem ? 1u : 0u), sb0 = blockIdx.x * wq_ + min(blockIdx.x, wrem);
    const uint32_t na = (nb + 1u) >> 1;                                  // pairs of half 0; half 1 takes the rest
    const uint32_t pb = sb0 + (half ? na : 0u), cnt = half ? nb - na : na; // this wave: quarter q of pairs pb .. pb + cnt - 1 (cnt <= HP)
    const uint32_t glog = group ? 31u - __builtin_clz(group) : 31u, ngroups = group ? K >> glog : 1u;
    const char* wbase = static_cast<const char*>(wp) + (size_t)(QC * q) * 1024u + lane * 16u;
    const char* sbase = static_cast<const char*>(sp);
    // ---- the wave's stream: tile t = chunk t % QC of row (t / QC) % 2 of pair t / (2 QC), in address order; a ring of RING tiles, every slot
    // refilled the moment its tile has been multiplied (gemv.h: straight-line code, every load unconditional -- a pair the wave does not have reads
    // one broadcast line: masks, not selects)
#ifndef MC_K4_RING
#define MC_K4_RING 4 // tiles (KiB per wave) in flight.  Gemma-7B shapes, same box, alternating, tokens/s (the one-pair-per-wave kernel: 675.2 / 677.2):
                     // 2: 670.4 / 672.2, 3: 688.7, 4: 685.0 / 684.8 / 689.4, 6: 684.4 / 681.6, 10: 670.0 / 672.5 (profiles/r06_ab_k4_ring*.log)
#endif
    constexpr int RING = MC_K4_RING, TPP = 2 * QC, NT = HP * TPP;
    uint4 ring[RING];
    uint32_t scs[HP][QC];
    auto req_tile = [&](uint4& dst, int t) {
        const uint32_t i = (uint32_t)(t / TPP), r = (uint32_t)((t / QC) % 2), c = (uint32_t)(t % QC);
        const uint32_t lm = 0u - (uint32_t)(i < cnt ? 1u : 0u);
        const uint64_t lm64 = ((uint64_t)lm << 32) | lm;
        const char* a = wbase + ((((uint64_t)(pb + i) * 2u + r) * ROWB + (uint64_t)c * 1024u) & lm64);
        const rowv4 v = __builtin_nontemporal_load(reinterpret_cast<const rowv4*>(a));
        dst = make_uint4(v.x, v.y, v.z, v.w);
    };
    // scales: row quads [ceil(out / 4)][ngroups][4] bf16; the dword (rows 2 pr, 2 pr + 1) of the lane's group of chunk QC q + c
    auto req_scales = [&](uint32_t (&sc)[QC], uint32_t i) {
        const uint32_t lm = 0u - (uint32_t)(i < cnt ? 1u : 0u);
        const uint64_t lm64 = ((uint64_t)lm << 32) | lm;
        const uint32_t pr = (pb + i) & lm;
        const char* s0 = sbase + (((((uint64_t)(pr >> 1) * ngroups) * 4 + (pr & 1u) * 2) * 2) & lm64);
#pragma unroll
        for (int c = 0; c < QC; c++) {
            const uint32_t g = group ? ((2048u * (uint32_t)(QC * q + c) + 32u * lane) >> glog) : 0u;
            sc[c] = *reinterpret_cast<const uint32_t*>(s0 + ((g * 8u) & lm));
        }
    };
    req_scales(scs[0], 0u);
#pragma unroll
    for (int t = 0; t < RING; t++) req_tile(ring[t], t);
#pragma unroll
    for (int i = 1; i < HP; i++) req_scales(scs[i], (uint32_t)i);
    uint32_t eo_res[HP];
    if constexpr (EPI == EPI_RESID) { // (the residual dwords of the pairs this wave's lanes may finish: requested early, gemv.h)
#pragma unroll
        for (int i = 0; i < HP; i++) eo_res[i] = reinterpret_cast<const uint32_t*>(resp)[min(pb + (uint32_t)i, NP - 1u)];
    }
    // ---- the quarter into LDS
    {
        const uint32_t t2 = (half << 6) | lane;
#pragma unroll
        for (int i = 0; i < NXP; i++) {
            const uint32_t l = t2 + 128u * (uint32_t)i;
            if (l < 256u * QC) {
                const uint32_t p = 256u * QC * q + l;
                reinterpret_cast<rowv4*>(xs)[p + (p >> 4)] = xr[i];
            }
        }
    }
    lds_barrier(); // (LDS only: the weights stay in flight)
    const uint32_t lane_tr = (((lane >> 4) * 4 + (lane & 3)) * 17 + ((lane >> 2) & 3) * 4) * 16;
    const m4b_lane m4bk = m4b_lane_consts(lane);
    typedef __attribute__((address_space(3))) mf_s4 lds_s4;
    uint2 xq[QC][8];
#pragma unroll
    for (int c = 0; c < QC; c++) {
        lds_s4* xt = (lds_s4*)(xs + (QC * q + c) * CHUNK_LDS + lane_tr);
#pragma unroll
        for (int e = 0; e < 8; e++) xq[c][e] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(xt + e));
    }
    // ---- a quarter row = QC chunks into one accumulator, the lane's own element, one wave sum (gemv.h do_pair's order inside the quarter)
    mf_f4 acc[1] = {mf_f4{0, 0, 0, 0}};
    float ra = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const uint32_t i = (uint32_t)(t / TPP);
        const int r = (t / QC) % 2, c = t % QC;
        if (i < cnt) { // (wave-uniform; no load inside)
            const uint32_t raw = scs[t / TPP][c];
            mac4b_n<1>(acc, ring[t % RING], m4b_prepare(r ? (raw & 0xFFFF0000u) : (raw << 16), m4bk), xq[c]);
            if (c == QC - 1) {
                const uint32_t e = lane & 3;
                const float mine = e == 0 ? acc[0][0] : (e == 1 ? acc[0][1] : (e == 2 ? acc[0][2] : acc[0][3]));
                const float rs = wave_sum_dpp(mine) * 0x1p37f; // 2^M4B_Q: the sum was formed at 2^-Q (mac4b_n)
                acc[0] = mf_f4{0, 0, 0, 0};
                if (r == 0) {
                    ra = rs;
                } else if (lane == 0) {
                    const uint32_t pl = (half ? na : 0u) + i; // the pair's index inside the workgroup
                    part[(pl * 2u + 0u) * 4u + q] = ra;
                    part[(pl * 2u + 1u) * 4u + q] = rs;
                }
            }
        }
        if (t + RING < NT) req_tile(ring[t % RING], t + RING);
    }
    lds_barrier();
    // ---- the rows: quarters added in order, one rounding to T; lane i of the first wave of each half finishes pair i of that half
    if (q == 0u && lane < cnt) {
        const uint32_t pl = (half ? na : 0u) + lane;
        const float* p0 = part + (pl * 2u) * 4u;
        const float a = ((p0[0] + p0[1]) + p0[2]) + p0[3], b = ((p0[4] + p0[5]) + p0[6]) + p0[7];
        float va = BF::rt(a), vb = BF::rt(b);
        if constexpr (EPI == EPI_RESID) { // add in T (kernel/arithmetic.metal:13-46)
            const uint32_t rr = lane == 0 ? eo_res[0] : (lane == 1 ? eo_res[1] : (lane == 2 ? eo_res[2] : eo_res[3]));
            va = asf(rr << 16) + va;
            vb = asf(rr & 0xFFFF0000u) + vb;
        }
        reinterpret_cast<uint32_t*>(yp)[pb + lane] = pack_bf16x2(va, vb);
    }
}

} // namespace gemv
} // namespace mc
