// Decode attention AND the output projection of include/metalchat/nn/attention.h:191-205 in one launch:
//   mc_attn_wo_i4_bfloat = mc_attn_fused_bfloat (scores, softmax, P.V: decode_kernels.hip) + hand-off C + the Wo GEMV with its
//   residual (transformer.h:132-133), for int4 weights on bfloat rows (the arithmetic of mc_gemv_i4_bfloat_lin{1,2,4}_p0_e{0,1}).
//
// Why: the Wo GEMV streams 34 KB per CU (1.4 us) and lasts 5.0 us -- its launch ramp, its prologue and its first tiles' latency
// are all it consists of.  Here its weights are requested behind the scores (one row pair per wave: they sit in registers
// before they are needed: the two hand-offs of the attention cover them), the attention row reaches every workgroup through one more in-launch hand-off (C: the 16-value
// chunks the reduce of hand-off B finishes are published as {2 x bf16, tag} granules and swept by all eight waves into LDS), and
// what is left of the GEMV is its arithmetic and its epilogue.
//
// Numerics: bit for bit the two launches it replaces -- the attention phases are the same code (attn_fused_bf, NW = 8: the scores
// stay on four waves, P.V column blocks and reduce chunks are dealt over eight: who adds changes, not what is added), the row in
// LDS is the row mc_attn_fused_bfloat leaves in HBM, and a row pair is multiplied packet by packet as the linear-order kernel does
// (mac4b_n over the chunks in order, one wave reduction, T(sum), the residual added in T).
// The residual row is read and the output row written IN PLACE (`res` == `y` == the hidden row, as the Wo GEMV is launched):
// every workgroup writes behind hand-off C, which completes only when every workgroup of the launch has finished its chunks
// of the reduce -- and nothing of this launch reads the hidden row except the lane that is about to overwrite the pair it read.
#include "gemv.h"

namespace {

using namespace mc;
using namespace mc::gemv;

// LNCH = KiB of packed weights per Wo row (K = H * hd = 2048 LNCH)
template <int HD, int LNCH>
__device__ __forceinline__ void
attn_wo_body(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vt, bf16_t* __restrict__ attn_out,
             unsigned long long* psum_g, unsigned long long* slab_g, unsigned long long* row_g, step_state* st, uint32_t n_rep,
             uint32_t KV, uint32_t max_seq, float scale, uint32_t nsplit, uint32_t layer_tag, const void* __restrict__ wo_w,
             const void* __restrict__ wo_s, const bf16_t* res, bf16_t* y, uint32_t out_rows, uint32_t group, uint32_t has_res,
             uint32_t fastpath, unsigned long long* tl)
{
    constexpr uint32_t K = 2048u * LNCH;
    constexpr uint32_t CHUNK_LDS = 2048 * 2 / 16 * 17;  // a chunk of the row in LDS: 16 bytes of padding per 256 (gemv.h Q_M4D)
    constexpr uint32_t ROWB = K / 2;                     // bytes of packed weights per row
    __shared__ __attribute__((aligned(16))) char xs[LNCH * CHUNK_LDS];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- the Wo row pairs of this wave: contiguous spans dealt as the linear-order kernels deal theirs, at most PMAX each (the
    // host takes this kernel only then).  Their weights are requested BEHIND the scores (attn_fused_bf's hook): at the start of the
    // launch they queue in front of the K tile in the CU's memory pipe (measured: 720 tokens/s against 732 with the Wo GEMV as a
    // launch of its own); behind the scores the hand-offs of the attention cover them.
    constexpr int PMAX = 2;
    const uint32_t NP = out_rows / 2;
    uint32_t pb, pe;
    static_assert(MC_LIN_FAVOUR < 63, "with at most 16 pairs per workgroup (the host's condition) no wave may get a third");
    lin_deal<8>(NP, wave, 8u, pb, pe); // (gemv.h: the same pairs per workgroup, a remainder to waves 0-3 first)
    const uint32_t glog = group ? 31u - __builtin_clz(group) : 31u, ngroups = group ? K >> glog : 1u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint4 ww[PMAX][2][LNCH];
    uint32_t ws[PMAX][LNCH], wres[PMAX];
    // Who asks when (round 4): a wave's vector-memory results return in issue order, so a wave that polls a hand-off behind its own
    // 8 KB of weight requests sees its first granule only when those have arrived.  Waves 0-3 gather the denominators of hand-off
    // A (one query head each): they request their pairs BEHIND that wait (point 1); waves 4-7 wait for nothing there and request
    // behind the scores (point 0), as every wave did in round 3.
#ifndef MC_WO_REQ_SPLIT
#define MC_WO_REQ_SPLIT 1
#endif
    auto request_wo = [&](int point) {
        const bool poller = wave < n_rep; // (attn_fused_bf: head = wave, wave + NW, ... gathers the denominators of hand-off A)
        if (MC_WO_REQ_SPLIT ? (point == 0) == poller : point != 0) return;
#pragma unroll
        for (int i = 0; i < PMAX; i++) {
            if (pb + i >= pe) break; // (wave-uniform)
            const uint32_t pr = pb + i;
            const char* wrow = static_cast<const char*>(wo_w) + (size_t)pr * 2 * ROWB + lane * 16;
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int c = 0; c < LNCH; c++) {
                    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + (size_t)r * ROWB + c * 1024));
                    ww[i][r][c] = make_uint4(v.x, v.y, v.z, v.w);
                }
            // scales: row quads [ceil(out / 4)][ngroups][4] bf16 (gemv.h); the dword (rows 2 pr, 2 pr + 1) of the lane's group
            const char* srow = static_cast<const char*>(wo_s) + (((size_t)(pr >> 1) * ngroups) * 4 + (pr & 1u) * 2) * 2;
#pragma unroll
            for (int c = 0; c < LNCH; c++) {
                const uint32_t g = group ? ((2048u * c + 32u * lane) >> glog) : 0u;
                ws[i][c] = *reinterpret_cast<const uint32_t*>(srow + g * 8u);
            }
            wres[i] = has_res ? reinterpret_cast<const uint32_t*>(res)[pr] : 0u;
        }
    };

    // ---- attention (decode_kernels.hip); every finished 16-value chunk goes out as eight {2 x bf16, tag} granules
    const uint32_t epoch_tag = st->epoch * 256u + layer_tag;
    auto publish = [&](uint32_t head, uint32_t db, uint32_t col, float v) {
        const float vn = __shfl_down(v, 1, 64);
        if (lane < 16) {
            attn_out[(size_t)head * HD + db * 16 + col] = f2bf(v); // (the row in HBM as mc_attn_fused_bfloat leaves it: taps, tools)
            if ((col & 1u) == 0) granule_store(row_g + ((size_t)head * HD + db * 16 + col) / 2, epoch_tag, pack_bf16x2(v, vn));
        }
    };
    attn_fused_bf<HD, 1, 8>(q, kc, vt, psum_g, slab_g, st, n_rep, KV, max_seq, scale, nsplit, layer_tag, tl, publish, request_wo, fastpath);
    // tl != null (tools/attn_wo_timeline.py only): thread 0 of every workgroup leaves s_memrealtime stamps of its phases
    auto stamp = [&](int i) {
        if (tl && threadIdx.x == 0) tl[(size_t)blockIdx.x * 8 + i] = __builtin_amdgcn_s_memrealtime();
    };

    // ---- hand-off C: the whole attention row (K bf16 = K / 2 granules) into LDS, padded as the transposed reads want it
    {
        constexpr int NG = (int)(K / 2 / 512); // granules per thread
        uint32_t val[NG];
        handoff_wait w;
        for (;;) { // (one watched granule and ~ 0.4 us between looks while the row is not there: chain_kernels.hip)
            const bool seen = (uint32_t)(granule_load(row_g + tid + 512u * (NG - 1)) >> 32) == epoch_tag;
            if (__all(seen) || w.expired(st, 0xC0000000u | layer_tag)) break;
#ifndef MC_HANDOFF_C_SLEEP
#define MC_HANDOFF_C_SLEEP 16
#endif
            __builtin_amdgcn_s_sleep(MC_HANDOFF_C_SLEEP);
        }
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < NG; i++) {
                const unsigned long long g = granule_load(row_g + tid + 512u * i);
                ok = ok && (uint32_t)(g >> 32) == epoch_tag;
                val[i] = (uint32_t)g;
            }
            if (__all(ok) || w.expired(st, 0xC0000000u | layer_tag)) break;
        }
#pragma unroll
        for (int i = 0; i < NG; i++) {
            const uint32_t g = tid + 512u * i, p = g >> 2; // granule g = elements 2 g, 2 g + 1 = dword g % 4 of packet g / 4
            *reinterpret_cast<uint32_t*>(xs + (p + (p >> 4)) * 16 + (g & 3u) * 4) = val[i];
        }
    }
    __syncthreads();
    stamp(6);

    // ---- Wo: mc_gemv_i4_bfloat_lin{LNCH}_p0_e{0,1} for the wave's pairs
    const uint32_t lane_tr = (((lane >> 4) * 4 + (lane & 3)) * 17 + ((lane >> 2) & 3) * 4) * 16;
    const m4b_lane m4bk = m4b_lane_consts(lane);
    typedef __attribute__((address_space(3))) mf_s4 lds_s4;
#pragma unroll
    for (int i = 0; i < PMAX; i++) {
        if (pb + i >= pe) break;
        float rsum[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            mf_f4 acc[1] = {mf_f4{0, 0, 0, 0}};
#pragma unroll
            for (int c = 0; c < LNCH; c++) {
                lds_s4* xt = (lds_s4*)(xs + c * CHUNK_LDS + lane_tr);
                uint2 x[8];
#pragma unroll
                for (int e = 0; e < 8; e++) x[e] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(xt + e));
                const uint32_t raw = ws[i][c];
                mac4b_n<1>(acc, ww[i][r][c], m4b_prepare(r ? (raw & 0xFFFF0000u) : (raw << 16), m4bk), x);
            }
            const uint32_t e = lane & 3;
            const float mine = e == 0 ? acc[0][0] : (e == 1 ? acc[0][1] : (e == 2 ? acc[0][2] : acc[0][3]));
            rsum[r] = wave_sum_dpp(mine) * 0x1p37f; // 2^M4B_Q: the sum was formed at 2^-Q (mac4b_n)
        }
        if (lane == 0) {
            float va = BF::rt(rsum[0]), vb = BF::rt(rsum[1]);
            if (has_res) { // add in T (kernel/arithmetic.metal:13-46)
                va = __uint_as_float(wres[i] << 16) + va;
                vb = __uint_as_float(wres[i] & 0xFFFF0000u) + vb;
            }
            reinterpret_cast<uint32_t*>(y)[pb + i] = pack_bf16x2(va, vb);
        }
    }
    stamp(7);
}

} // namespace

#define MC_ATTN_WO(NAME, HD, LNCH)                                                                                                        \
    extern "C" __global__ void __launch_bounds__(512)                                                                                    \
    NAME(const bf16_t* q, const bf16_t* kc, const bf16_t* vt, bf16_t* attn_out, unsigned long long* psum_g, unsigned long long* slab_g, \
         unsigned long long* row_g, step_state* st, uint32_t n_rep, uint32_t n_kv, uint32_t max_seq, float scale, uint32_t nsplit,      \
         uint32_t layer_tag, const void* wo_w, const void* wo_s, const bf16_t* res, bf16_t* y, uint32_t out_rows, uint32_t group,       \
         uint32_t has_res, uint32_t fastpath, unsigned long long* tl)                                                                    \
    {                                                                                                                                    \
        attn_wo_body<HD, LNCH>(q, kc, vt, attn_out, psum_g, slab_g, row_g, st, n_rep, n_kv, max_seq, scale, nsplit, layer_tag, wo_w,     \
                               wo_s, res, y, out_rows, group, has_res, fastpath, tl);                                                    \
    }
// mc_attn_wo_i4_bfloat_hd{head_dim}_k{KiB per Wo row}
MC_ATTN_WO(mc_attn_wo_i4_bfloat_hd128_k2, 128, 2)  // Llama-3-8B: 32 heads x 128
// (hd128_k4 -- Llama-3-70B's 64 heads, 131 KB of Wo per CU -- was built and bit-identical too, and SLOWER than the two launches:
//  20.1 us against 9.3 + 9.7, profiles/r03_kernel_stats_70b*.csv: what this kernel gains is the Wo weights waiting in registers
//  when the row arrives, and two row pairs of 4 KiB rows per wave are 64 registers requested behind the scores, not 16)
MC_ATTN_WO(mc_attn_wo_i4_bfloat_hd64_k1, 64, 1)    // 32 heads x 64
MC_ATTN_WO(mc_attn_wo_i4_bfloat_hd256_k2, 256, 2)  // Gemma-7B shapes: 16 heads x 256
