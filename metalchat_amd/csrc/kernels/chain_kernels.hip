// The END of one transformer block and the START of the next in one launch:
//   mc_w2_qkv_i4_bfloat = w2 GEMV + residual of block l (transformer.h:59, 138-139; mc_gemv_i4_bfloat_lin{7,14}_p0_e1)
//                       + hand-off + attention_norm + wq|wk|wv GEMV + RoPE + cache write of block l + 1
//                         (transformer.h:130, attention.h:170-177; mc_gemv_i4_bfloat_lin{2,4}_p1_e4).
//
// Why: the wq|wk|wv launch lasts 6.8 us and streams 49 KB per CU (2 us) -- launch ramp, row staging, first-tile latency and the
// epilogue are most of it.  A chained phase gains what it can have requested BEFORE its row arrives (tools/experiments/README.md:
// chaining w1|w3, whose 230 KB per CU cannot be, lost 1 us), and wq|wk|wv fits: 1.5 (8B) / 2.5 (70B) row pairs per wave are
// 32 / 96 registers.  They are requested when the wave has requested its last w2 tile (gemv.h body's tail hook), together with the
// norm weights, the step state and the rotation of the wave's pairs; they arrive while the w2 phase reduces, adds the residual,
// stores and publishes its pair of the hidden row ({2 x bf16, tag} granules, CH_ROW_OUT) and while the workgroup waits for the
// slowest pair of the chip.  What is left of the second GEMV is the sweep of the row, the rmsnorm, two or three row pairs of
// arithmetic per wave from registers and the epilogue.
//
// Numerics: bit for bit the two launches.  Phase 1 IS the stand-alone kernel's code.  Phase 2 holds packet p of the row in the
// thread that holds it in mc_gemv_i4_bfloat_lin{2,4}_p1_e4 (the sum of squares is added in the same order), normalises with the
// same expression, multiplies a row chunk by chunk with mac4b_n, reduces with the same wave sum and finishes a pair with the
// same function (gemv.h qkv_rope_finish).  tests/test_context_gpu.py compares hidden rows, logits, caches and tokens for equality.
#include "gemv.h"

namespace {

using namespace mc;
using namespace mc::gemv;

// LQ = KiB per wq|wk|wv row (dim / 2048); PMAX = most row pairs a wave can own (ceil(rows / 2 / (8 * workgroups)), host-checked)
template <int LQ, int PMAX>
struct qkv_phase {
    static constexpr uint32_t K = 2048u * LQ, ROWB = K / 2, CHUNK_LDS = 2048 * 2 / 16 * 17;
    static constexpr uint32_t NPK = 256u * LQ, BD = 512;
    static constexpr int NXP = (int)((NPK + BD - 1) / BD);
    typedef uint32_t rowv4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

    uint4 ww[PMAX][2][LQ];
    uint32_t ws[PMAX][LQ];
    rowv4 nr[NXP];
    qkv_epilogue qe;
    uint32_t slot, rrow, pb, pe, eo_pair;
    float eo_c, eo_s;

    // at the START of the launch (scalar loads of the descriptor and the step state, nothing of this kernel has stored yet -- behind
    // stores hipcc reads such fields with vector loads and waits for them with vmcnt(0), i.e. for every weight in flight): the wave's
    // range of row pairs, the write slot, the rotation of the pair each lane will finish
    __device__ __forceinline__ void
    early(const qkv_epilogue* qep, uint32_t out_rows)
    {
        const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const uint32_t NP = out_rows / 2, nw_total = gridDim.x * 8u, gw = blockIdx.x * 8u + wave;
        const uint32_t pq = NP / nw_total, prem = NP - pq * nw_total;
        pb = gw * pq + min(gw, prem);
        pe = pb + pq + (gw < prem ? 1u : 0u);
        qe = *qep;
        const __attribute__((address_space(1))) int32_t* stp = (const __attribute__((address_space(1))) int32_t*)qe.state;
        slot = (uint32_t)stp[3];
        rrow = (uint32_t)stp[6];
        eo_pair = min(pb + lane, NP - 1);
        typedef const __attribute__((address_space(1))) float* gfloat_p;
        const uint32_t hd = qe.hd, row = 2 * eo_pair;
        const uint32_t j = row < (qe.H + qe.KV) * hd ? (row % hd) / 2 : 0u;
        eo_c = ((gfloat_p)qe.fcos)[(size_t)rrow * (hd / 2) + j];
        eo_s = ((gfloat_p)qe.fsin)[(size_t)rrow * (hd / 2) + j];
        uint32_t never;
        asm volatile("s_mov_b32 %0, 0" : "=s"(never));
        if (never) asm volatile("" ::"v"(eo_c), "v"(eo_s), "s"(slot), "s"(qe.H), "s"(qe.KV), "s"(qe.hd), "s"(qe.max_seq)); // (issued HERE)
    }

    // the weights, scales and norm weights: requested from the tail of the phase in front.  Straight-line code, every load
    // unconditional (a pair the wave does not own reads one broadcast line): hipcc then counts its waits, and the epilogue of the
    // phase in front does not wait for these loads.
    __device__ __forceinline__ void
    request(const void* __restrict__ qkv_w, const void* __restrict__ qkv_s, const void* __restrict__ norm_w, uint32_t group)
    {
        const uint32_t tid = threadIdx.x, lane = tid & 63;
        const uint32_t glog = group ? 31u - __builtin_clz(group) : 31u, ngroups = group ? K >> glog : 1u;
#pragma unroll
        for (int i = 0; i < PMAX; i++) {
            const uint32_t pr = pb + i, lm = 0u - (uint32_t)(pr < pe);
            const uint64_t lm64 = ((uint64_t)lm << 32) | lm;
            const char* wrow = static_cast<const char*>(qkv_w) + (((uint64_t)pr * 2 * ROWB) & lm64) + ((lane * 16) & lm);
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int c = 0; c < LQ; c++) {
                    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + (((uint32_t)r * ROWB + c * 1024) & lm)));
                    ww[i][r][c] = make_uint4(v.x, v.y, v.z, v.w);
                }
            // scales: row quads [ceil(out / 4)][ngroups][4] bf16 (gemv.h); the dword (rows 2 pr, 2 pr + 1) of the lane's group
            const char* srow = static_cast<const char*>(qkv_s) + (((((uint64_t)(pr >> 1) * ngroups) * 4 + (pr & 1u) * 2) * 2) & lm64);
#pragma unroll
            for (int c = 0; c < LQ; c++) {
                const uint32_t g = group ? ((2048u * c + 32u * lane) >> glog) : 0u;
                ws[i][c] = *reinterpret_cast<const uint32_t*>(srow + ((g * 8u) & lm));
            }
        }
#pragma unroll
        for (int i = 0; i < NXP; i++) nr[i] = reinterpret_cast<const rowv4*>(norm_w)[min(tid + i * BD, NPK - 1)];
    }

    // xs: this phase's LDS (LQ * CHUNK_LDS bytes for the row + 32 floats); stamp(k): time stamps of tools/chain_timeline.py
    template <typename Stamp>
    __device__ __forceinline__ void
    run(char* xs, const unsigned long long* hid_g, uint32_t tag, step_state* st, uint32_t out_rows, float eps, float mu, Stamp&& stamp)
    {
        const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        float* red = reinterpret_cast<float*>(xs + LQ * CHUNK_LDS);
        auto live = [&](int i) { return NPK % BD == 0 || i < NXP - 1 || tid + i * BD < NPK; };
        // ---- the hidden row: packet p = granules 4 p .. 4 p + 3, in the thread that holds packet p in the stand-alone kernel
        rowv4 xr[NXP];
        {
            // Waves that finished their part of the phase in front early wait here while others still stream: a wave first watches
            // ONE granule (its last), sleeping ~ 0.4 us between looks, and sweeps its packets only when that one has arrived --
            // with every thread re-reading four granules every 0.1 us the waiting waves asked the fabric for 4 MB per round and
            // the weight requests of this phase took up to 5 us to issue (tools/chain_timeline.py)
            handoff_wait w;
            const uint32_t last = 4u * min(tid + (NXP - 1) * BD, NPK - 1) + 3u;
            for (;;) {
                const bool seen = (uint32_t)(granule_load(hid_g + last) >> 32) == tag;
                if (__all(seen) || w.expired(st, 0xD0000000u | (tag & 0xFFu))) break;
                __builtin_amdgcn_s_sleep(16);
            }
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < NXP; i++) {
                    const uint32_t pc = min(tid + i * BD, NPK - 1);
                    uint32_t v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const unsigned long long g = granule_load(hid_g + 4u * pc + e);
                        ok = ok && (uint32_t)(g >> 32) == tag;
                        v[e] = (uint32_t)g;
                    }
                    xr[i] = rowv4{v[0], v[1], v[2], v[3]};
                }
                if (__all(ok) || w.expired(st, 0xD0000000u | (tag & 0xFFu))) break;
            }
        }
        stamp(3);
        // ---- attention_norm (kernel/rmsnorm.metal:52-95), the additions in the order of gemv.h's build-time prologue
        {
            float ss = 0.0f;
#pragma unroll
            for (int i = 0; i < NXP; i++) {
                const uint32_t vv[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w};
                float s1 = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float a = asf(vv[e] << 16), b = asf(vv[e] & 0xFFFF0000u);
                    s1 += a * a;
                    s1 += b * b;
                }
                ss += live(i) ? s1 : 0.0f;
            }
            const float wsum_ = wave_sum_dpp(ss);
            if (lane == 0) red[wave] = wsum_;
            __syncthreads();
            float tot = 0.0f;
#pragma unroll
            for (uint32_t i = 0; i < 8u; i++) tot += red[i];
            const float inv = 1.0f / sqrtf(tot / (float)K + eps);
            rowv4* xlv = reinterpret_cast<rowv4*>(xs);
#pragma unroll
            for (int i = 0; i < NXP; i++) {
                const uint32_t vv[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w};
                const uint32_t wn[4] = {nr[i].x, nr[i].y, nr[i].z, nr[i].w};
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float a = (mu + asf(wn[e] << 16)) * asf(vv[e] << 16) * inv;
                    const float b = (mu + asf(wn[e] & 0xFFFF0000u)) * asf(vv[e] & 0xFFFF0000u) * inv;
                    o[e] = pack_bf16x2(a, b);
                }
                const uint32_t p = tid + i * BD;
                if (live(i)) xlv[p + (p >> 4)] = rowv4{o[0], o[1], o[2], o[3]};
            }
        }
        __syncthreads();
        stamp(4);
        // ---- the wave's row pairs from registers (the arithmetic of the linear-order kernels: mac4b_n chunk after chunk)
        const uint32_t lane_tr = (((lane >> 4) * 4 + (lane & 3)) * 17 + ((lane >> 2) & 3) * 4) * 16;
        const m4b_lane m4bk = m4b_lane_consts(lane);
        typedef __attribute__((address_space(3))) mf_s4 lds_s4;
        float ra[PMAX], rb[PMAX];
#pragma unroll
        for (int i = 0; i < PMAX; i++) {
            ra[i] = rb[i] = 0.0f;
            if (pb + i >= pe) break;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                mf_f4 acc[1] = {mf_f4{0, 0, 0, 0}};
#pragma unroll
                for (int c = 0; c < LQ; c++) {
                    lds_s4* xt = (lds_s4*)(xs + c * CHUNK_LDS + lane_tr);
                    uint2 x[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) x[e] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(xt + e));
                    const uint32_t raw = ws[i][c];
                    mac4b_n<1>(acc, ww[i][r][c], m4b_prepare(r ? (raw & 0xFFFF0000u) : (raw << 16), m4bk), x);
                }
                const uint32_t e = lane & 3;
                const float mine = e == 0 ? acc[0][0] : (e == 1 ? acc[0][1] : (e == 2 ? acc[0][2] : acc[0][3]));
                const float rs = wave_sum_dpp(mine) * 0x1p37f; // 2^M4B_Q: the sum was formed at 2^-Q (mac4b_n)
                if (r == 0) ra[i] = rs;
                else rb[i] = rs;
            }
        }
        // ---- rotation + cache write: lane i finishes pair pb + i (its cos / sin arrived long ago)
        float a = ra[0], b = rb[0];
#pragma unroll
        for (int i = 1; i < PMAX; i++) {
            a = lane == (uint32_t)i ? ra[i] : a;
            b = lane == (uint32_t)i ? rb[i] : b;
        }
        if (pb + lane < pe) {
            const uint32_t row = 2 * (pb + lane);
            qkv_rope_finish<BF>(qe, slot, row, row + 1 < out_rows, a, b, eo_c, eo_s);
        }
        stamp(5);
    }
};

} // namespace

// mc_w2_qkv_i4_bfloat_w{KiB per w2 row}_q{KiB per wq|wk|wv row}; ... = the w2 family's ring configuration (gemv_kernels.hip)
#define MC_W2_QKV(NAME, LQ, PMAX, ...)                                                                                                   \
    extern "C" __global__ void __launch_bounds__(512)                                                                                    \
    NAME(const void* w2_w, const void* w2_s, const void* gate, void* hidden, const void* res, uint32_t dim, uint32_t ffn,               \
         uint32_t w2_group, unsigned long long* hid_g, step_state* st, uint32_t layer_tag, uint32_t lds_off, const void* qkv_w,          \
         const void* qkv_s, const void* norm_w, const qkv_epilogue* qe, uint32_t qkv_rows, uint32_t qkv_group, float eps, float mu,      \
         unsigned long long* tl)                                                                                                         \
    {                                                                                                                                    \
        extern __shared__ __attribute__((aligned(16))) char smem[];                                                                      \
        const uint32_t tag = st->epoch * 256u + layer_tag;                                                                               \
        /* tl != null (tools/chain_timeline.py only): eight time stamps per wave */                                                      \
        auto stamp = [&](int k) {                                                                                                        \
            if (tl && (threadIdx.x & 63) == 0) tl[(size_t)(blockIdx.x * 8u + (threadIdx.x >> 6)) * 8 + k] = __builtin_amdgcn_s_memrealtime(); \
        };                                                                                                                               \
        stamp(0);                                                                                                                        \
        qkv_phase<LQ, PMAX> next;                                                                                                        \
        next.early(qe, qkv_rows);                                                                                                        \
        auto hook = [&] {                                                                                                                \
            stamp(1);                                                                                                                    \
            next.request(qkv_w, qkv_s, norm_w, qkv_group);                                                                               \
        };                                                                                                                               \
        body<WF_I4, BF, Q_M4D, PRO_NONE, EPI_RESID, 4, __VA_ARGS__, 8, 0, 0, CH_ROW_OUT>(                                                \
            w2_w, w2_s, gate, hidden, res, nullptr, dim, ffn, w2_group, eps, mu, nullptr, nullptr, 0u, 0.0f, chain_args{hid_g, tag},     \
            hook);                                                                                                                       \
        stamp(2);                                                                                                                        \
        next.run(smem + lds_off, hid_g, tag, st, qkv_rows, eps, mu, stamp);                                                              \
    }
MC_W2_QKV(mc_w2_qkv_i4_bfloat_w7_q2, 2, 2, MC_LIN7_CFG)   // Llama-3-8B: ffn 14336, dim 4096, 6144 rows of wq|wk|wv
MC_W2_QKV(mc_w2_qkv_i4_bfloat_w14_q4, 4, 3, MC_LIN14_CFG) // Llama-3-70B: ffn 28672, dim 8192, 10240 rows
