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


// ------------------------------------------------------------------------------------------
// wq|wk|wv INSIDE the attention launch (round 4, mc_attn_qkv_wo_*): the query-source policy of attn_fused_bf (decode_kernels.hip)
// that computes the step's queries and its K / V row in the launch that consumes them.
//
// Why: as a launch of its own the wq|wk|wv GEMV lasts 6.2 us for 2.1 us of stream -- a launch boundary, the wave start, a
// dependent prologue and an epilogue whose results make a round trip through HBM, then the attention launch pays the same again
// before its first score.  Attention needs nothing of wq|wk|wv but the rows of ITS kv head: the 32 workgroups of kv head g (in
// practice one XCD: blockIdx.x % KV) compute exactly those -- the n_rep query heads, the K row and the V row of head g, 384 row
// pairs of the packed matrix, 12 per workgroup, 1 or 2 per wave, all of a wave's weights requested up front (<= 32 registers) --
// and hand them to one another as {2 x bf16, tag} granules through the XCD's L2 (hand-off Q: handoff.h "the XCD-local fast
// path"; a member on another XCD is served by the fabric copy).  Nothing crosses kv heads, so the chip-wide exchange of the
// two-launch form (the rotated queries through HBM behind a kernel boundary) becomes 8 local ones.
//
// Numerics: bit for bit mc_gemv_i4_bfloat_lin{QN}_p1_e4 -- the rmsnorm prologue in the stand-alone order (per-thread packet sums,
// wave_sum_dpp, the eight wave sums in wave order), mac4b_n chunk by chunk into one accumulator per row, one wave sum, the
// epilogue of gemv.h finish_pair (EPI_QKV_ROPE: T(row sum), rotation in fp32 with two roundings, cache write).  Which workgroup
// multiplies a pair does not change a bit of it.  The packed matrix is read where it lies: pair j of group g is packed pair
// pp(j) (q heads of the group, then its k head, then its v head), 4 KiB of contiguous rows whichever wave takes it.
// ------------------------------------------------------------------------------------------
#ifndef MC_QKV_ROW_DEAL
#define MC_QKV_ROW_DEAL 0 // 1: the remainder pairs dealt in rows (below).  Same box, three alternating runs: 784 tokens/s with, 798 without
#endif
// WB = 1 (round 4, mc_attn_qkv_wo_w_bfloat_*): PLAIN bfloat weights (nn::linear; Llama-3.2-1B, the reference's default model) --
// rows of QN KiB hold 512 QN weights, no scales, the row in LDS in natural order, a row multiplied packet by packet with
// v_dot2_f32_bf16 as gemv.h mac<WF_T> does, one pair per wave at most.  Every difference is a compile-time branch: the int4
// instantiation is the code it was.
// WB = 2 (round 5, mc_attn_qkv_wo_i8_bfloat_*): int8 weights (quantization::linear at 8 bits) -- rows of QN KiB hold 1024 QN weights, one
// bfloat scale per row and group, the row in LDS in natural order, a packet of 16 weights dequantised and multiplied as gemv.h
// mac<Q_EXACT> (Wd = T(T(q) T(s)) per weight, fp32 sums by v_dot2c); up to two pairs per wave.  The stand-alone int8 kernels multiply on
// the matrix pipe (gemv.h mac8b_n): the same products in another order -- parity with the oracle, not identity with those launches.
// STG (round 5): where the K / V tile requests of the attention go.  0: in front of the wq|wk|wv phase (round 4).  A wave stalls at ISSUE once
// the CU's memory pipe is full, so requests written in front of the multiplications hold those back until most of the bytes have
// arrived (decode_kernels.hip attn_fused_bf, STAGED); 1: the K tile behind the first pair's multiplication, the V tile behind the second;
// 2: ... the V tile behind the polls of hand-off Q (a wave's loads return in order: polls behind its V tile see their granules when
// that has arrived).  Same box, alternating (tools/ab_hsaco.sh, tools/configs_run.py; 3-4 rounds each): Llama-3-8B int4 800.0 -> 808.9 tokens/s
// with 1 (789.2 with 2); int8 at S = 8192 502.0 -> 507.8 with 2 (501.5 with 1); plain bfloat weights (one pair per wave: one phase) 1513 ->
// 1503 with 2 on TinyLlama, 1527 -> 1526 with 1 (four alternating runs): left at 0.
#ifndef MC_QX_STG_I4
#define MC_QX_STG_I4 1
#endif
#ifndef MC_QX_STG_I8
#define MC_QX_STG_I8 2
#endif
#ifndef MC_QX_STG_W
#define MC_QX_STG_W 0
#endif
template <int HD, int QN, int WB = 0, int STG = (WB == 0 ? MC_QX_STG_I4 : (WB == 1 ? MC_QX_STG_W : MC_QX_STG_I8))>
struct qkv_in_launch {
    // (WB = 0, QN = 4 -- round 5, mc_attn_qkv_i4_bfloat_hd128_q4: Llama-3-70B's rows of 4 KiB (K = 8192): two packets of the hidden row per
    //  thread, up to THREE pairs per wave (20 per workgroup), 640 pairs per kv head gathered in two passes)
    static_assert(WB ? QN == 4 : (QN == 2 || QN == 4), "K = 4096 / 8192 int4, 4096 int8, 2048 bfloat; 512 threads");
    static constexpr bool WIDE = WB == 0 && QN == 4;
#ifndef MC_I8_SCORER_WAVES
#define MC_I8_SCORER_WAVES 8 // (int8, wide ranges) all eight waves compute scores: decode_kernels.hip attn_fused_bf SW; 507.7 -> 517.9 tokens/s same box, three alternating runs
#endif
    static constexpr int SCORER_WAVES = MC_I8_SCORER_WAVES; // (wide ranges score on all eight waves; 64-slot ranges on four either way)
    static constexpr bool LDS = true, PIN_V = true, STAGED = STG != 0;
    static constexpr int K_STEPS = WB == 1 ? 1 : (WIDE ? 3 : 2), V_STEP = STG == 2 ? K_STEPS : K_STEPS - 1; // (the polls are step K_STEPS)
    static constexpr int TL_STRIDE = 16, TL_BASE = 3; // stamps: 0 start, 1 row staged, 2 pairs published, 3.. attn_fused_bf's 0..
    static constexpr uint32_t KQ = WB == 1 ? 512u * QN : (WB == 2 ? 1024u * QN : 2048u * QN), ROWBQ = 1024u * QN, CHUNK_LDS = 2048 * 2 / 16 * 17, HALF = HD / 2;
    static constexpr uint32_t NPK = KQ / 8; // 16-byte packets of the hidden row
    static constexpr uint32_t WPK = WB == 1 ? 8u : (WB == 2 ? 16u : 32u); // weights of a lane's 16-byte packet
    static constexpr int PMAXQ = WB == 1 ? 1 : (WIDE ? 3 : 2);
    static constexpr int NXP = WIDE ? 2 : 1; // packets of the hidden row per thread
    typedef const __attribute__((address_space(3))) bf16_t* lds_row;
    typedef __attribute__((address_space(3))) bf16_t* lds_row_w;
    typedef uint32_t rowv4 __attribute__((ext_vector_type(4)));
    lds_row q_s, k_s, v_s;
    // ---- what the launch was given
    const void *xp, *normp, *qw, *qs;
    const float *fcos, *fsin;
    bf16_t *kc, *vt;
    unsigned long long* qkv_g; // [KV][(n_rep + 2) HD / 2] granules, the XCD-local copies behind them
    step_state* st;
    char* xs;     // the row in LDS (padded for the transposed reads: gemv.h Q_M4D)
    float* red;   // 16 floats of scratch
    uint32_t n_rep, KV, max_seq, nsplit, group, layer_tag, fastpath;
    uint32_t kv_shift; // virtual kv heads (decode_kernels.hip attn_fused_bf): n_rep and KV are the VIRTUAL counts, cache head = kv >> kv_shift
    float eps, mu;
    unsigned long long* tl;
    // ---- what at_start() leaves for before_scores()
    rowv4 xr[NXP], nr[NXP];
    uint4 ww[PMAXQ][2][QN];
    uint32_t wsc[PMAXQ][QN];
    float eo_c, eo_s;
    uint32_t j0, cnt, slot, tag, rrow_; // cnt: pairs this wave finishes (lane i < cnt: one each)

    __device__ __forceinline__ void stamp(int i) const
    {
        if (tl && threadIdx.x == 0) tl[(size_t)blockIdx.x * TL_STRIDE + i] = __builtin_amdgcn_s_memrealtime();
    }
    // packed pair (gemv.h EPI_QKV_ROPE: q heads, k heads, v heads; rotation partners adjacent) of pair j of kv head `kv`
    __device__ __forceinline__ uint32_t pp_of(uint32_t kv, uint32_t j) const
    {
        const uint32_t hq = n_rep * HALF, H = n_rep * KV, kvc = kv >> kv_shift, KVC = KV >> kv_shift; // (the K and V rows of the CACHE head)
        return j < hq ? kv * hq + j : (j < hq + HALF ? H * HALF + kvc * HALF + (j - hq) : (H + KVC) * HALF + kvc * HALF + (j - hq - HALF));
    }
    // The deal of the workgroup's PW pairs over its eight waves.  Waves w and w + 4 share SIMD w, and a wave alone on its SIMD
    // multiplies at ~ 60 % of the rate two reach together (MFMA and VALU issue overlap only ACROSS waves), so
    // whole pairs -- (2, 1) per SIMD for Llama-3-8B's 12 -- leave the second pair of wave w without a partner.  Where the
    // remainder is four pairs (PW % 8 == 4) they are dealt in ROWS: wave w < 4 takes e whole pairs and row 0 of a shared pair,
    // wave w + 4 takes e whole pairs and row 1 of that pair -- three rows each for the 12 -- hands its row sum over through LDS,
    // and wave w finishes the shared pair.  Who multiplies a row does not change a bit of it.
    // full: whole pairs of the wave, [j0, j0 + full); js: the shared pair (shared: the wave has a row of it), row sh_row.
    uint32_t full, js, shared, sh_row;
    // whole pair i of the wave: its 2 x QN KiB of weights and its scales (every load unconditional -- a load behind a branch costs
    // every counted s_waitcnt vmcnt(N) of the launch.  A pair the wave does not have reads one broadcast line of the matrix:
    // masks, not selects -- gemv.h ltile)
    __device__ __forceinline__ void request_pair(int i)
    {
        const uint32_t lane = threadIdx.x & 63, kv = blockIdx.x % KV;
        const uint32_t glog = group ? 31u - __builtin_clz(group) : 31u, ngroups = group ? KQ >> glog : 1u;
        const uint32_t lm = 0u - (uint32_t)((uint32_t)i < full ? 1u : 0u);
        const uint64_t lm64 = ((uint64_t)lm << 32) | lm;
        const uint32_t pp = pp_of(kv, j0 + i) & lm;
        const char* wrow = static_cast<const char*>(qw) + (((uint64_t)pp * 2 * ROWBQ) & lm64) + ((lane * 16) & lm);
#pragma unroll
        for (int rr = 0; rr < 2; rr++)
#pragma unroll
            for (int c = 0; c < QN; c++) {
                const rowv4 v = __builtin_nontemporal_load(reinterpret_cast<const rowv4*>(wrow + (((size_t)rr * ROWBQ + c * 1024) & lm64)));
                ww[i][rr][c] = make_uint4(v.x, v.y, v.z, v.w);
            }
        if constexpr (WB != 1) {
            const char* srow = static_cast<const char*>(qs) + (((size_t)(pp >> 1) * ngroups) * 4 + (pp & 1u) * 2) * 2;
#pragma unroll
            for (int c = 0; c < QN; c++) {
                const uint32_t g = group ? ((64u * WPK * c + WPK * lane) >> glog) : 0u;
                wsc[i][c] = *reinterpret_cast<const uint32_t*>(srow + ((g * 8u) & lm));
            }
        }
    }
    // the wave's row of the shared pair, into slot [PMAXQ - 1][0] (the wave has at most PMAXQ - 1 whole pairs then)
    __device__ __forceinline__ void request_shared_row()
    {
        const uint32_t lane = threadIdx.x & 63, kv = blockIdx.x % KV;
        const uint32_t glog = group ? 31u - __builtin_clz(group) : 31u, ngroups = group ? KQ >> glog : 1u;
        const uint32_t lm = 0u - shared;
        const uint64_t lm64 = ((uint64_t)lm << 32) | lm;
        const uint32_t pp = pp_of(kv, js) & lm;
        const char* wrow = static_cast<const char*>(qw) + ((((uint64_t)pp * 2 + sh_row) * ROWBQ) & lm64) + ((lane * 16) & lm);
#pragma unroll
        for (int c = 0; c < QN; c++) {
            const rowv4 v = __builtin_nontemporal_load(reinterpret_cast<const rowv4*>(wrow + (((size_t)c * 1024) & lm64)));
            ww[PMAXQ - 1][0][c] = make_uint4(v.x, v.y, v.z, v.w);
        }
        const char* srow = static_cast<const char*>(qs) + (((size_t)(pp >> 1) * ngroups) * 4 + (pp & 1u) * 2) * 2;
#pragma unroll
        for (int c = 0; c < QN; c++) {
            const uint32_t g = group ? ((2048u * c + 32u * lane) >> glog) : 0u;
            wsc[PMAXQ - 1][c] = *reinterpret_cast<const uint32_t*>(srow + ((g * 8u) & lm));
        }
    }
    // What is asked for when: a CU takes in ~ 25 GB/s and its waves stall at ISSUE once ~ 32 KB are outstanding (the first build
    // requested all 81 KB of weights and tiles up front: 3.2 - 3.8 us from the start to "row staged", the row's own round trip
    // being 0.6).  So: the row and every wave's FIRST pair (32 KB per CU) by the first instructions; the row staged while those
    // arrive; then the rest of the wave's rows and the K / V tiles, which arrive while the first pairs are multiplied.
    __device__ __forceinline__ void at_start()
    {
        const uint32_t tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        // the row first (gemv.h, the build-time prologue of the linear-order kernels), the step state behind it
        const uint32_t pk = WB == 1 ? tid & (NPK - 1u) : tid; // (bfloat weights: 256 packets, threads 256.. read them again and add nothing)
#pragma unroll
        for (int i = 0; i < NXP; i++) {
            xr[i] = reinterpret_cast<const rowv4*>(xp)[pk + 512u * i];
            nr[i] = reinterpret_cast<const rowv4*>(normp)[pk + 512u * i];
        }
        stamp(0);
        const __attribute__((address_space(1))) int32_t* stp = (const __attribute__((address_space(1))) int32_t*)st;
        slot = (uint32_t)stp[3];
        rrow_ = (uint32_t)stp[6];
        tag = (uint32_t)stp[9] * 256u + layer_tag;
        asm volatile("s_barrier" ::: "memory"); // (the row's requests stay ahead of the weight requests in the CU's memory pipe)
        const uint32_t split = blockIdx.x / KV;
        const uint32_t PG = (n_rep + 2u) * HALF, PW = PG / nsplit, e = PW >> 3, r = PW & 7u;
        if (MC_QKV_ROW_DEAL && !WIDE && WB == 0 && r == 4u && e + 1u <= (uint32_t)PMAXQ) {
            const uint32_t w4 = wave & 3u;
            full = e;
            j0 = split * PW + (wave < 4u ? w4 * (e + 1u) : 4u * (e + 1u) + w4 * e);
            js = split * PW + w4 * (e + 1u) + e;
            shared = 1u;
            sh_row = wave >> 2;
        } else {
            full = e + (wave < r ? 1u : 0u);
            j0 = split * PW + wave * e + min(wave, r);
            js = 0u;
            shared = 0u;
            sh_row = 0u;
        }
        // (the pairs this wave FINISHES: its whole pairs, and the shared pair on the wave that holds its row 0)
        cnt = full + (shared && sh_row == 0u ? 1u : 0u);
        request_pair(0);
    }
    __device__ __forceinline__ void before_tiles()
    {
        const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        // ---- rmsnorm on the way into LDS (kernel/rmsnorm.metal:52-95; the additions in the stand-alone kernel's order)
        {
            // (gemv.h, the build-time prologue: a sum per packet, the packets' sums added in order)
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
                ss += s1;
            }
            const float wsum_ = wave_sum_dpp(WB == 1 && tid >= NPK ? 0.0f : ss);
#if MC_ABL_NORM_NOXWAVE
            float tot = wsum_ * 8.0f; // (ablation build, gemv.h: timing only)
#else
            if (lane == 0) red[wave] = wsum_;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // (LDS only: the first pairs stay in flight)
            float tot = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) tot += red[i];
#endif
            const float inv = 1.0f / sqrtf(tot / (float)KQ + eps);
#pragma unroll
            for (int i = 0; i < NXP; i++) {
                const uint32_t vv[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w}, wv[4] = {nr[i].x, nr[i].y, nr[i].z, nr[i].w};
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float a = (mu + asf(wv[e] << 16)) * asf(vv[e] << 16) * inv;
                    const float b = (mu + asf(wv[e] & 0xFFFF0000u)) * asf(vv[e] & 0xFFFF0000u) * inv;
                    o[e] = pack_bf16x2(a, b);
                }
                const uint32_t p = tid + 512u * i;
                if constexpr (WB == 0) reinterpret_cast<rowv4*>(xs)[p + (p >> 4)] = rowv4{o[0], o[1], o[2], o[3]}; // (packet p sits in slot p + p / 16)
                else if (tid < NPK) reinterpret_cast<rowv4*>(xs)[tid] = rowv4{o[0], o[1], o[2], o[3]};              // (natural order)
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stamp(1);
        // ---- the rest of the wave's rows and the rotation of the pair this lane will finish (lane i < cnt: whole pair j0 + i,
        // then the shared pair; its address waits for the step state: asked for in at_start() that wait sat in the middle of
        // the weight requests)
        if constexpr (WIDE) {
            request_pair(1);
            request_pair(2);
        } else if constexpr (PMAXQ > 1) {
            if constexpr (WB == 0) {
                if (shared) request_shared_row();   // (wave-uniform, and the same for every wave of the launch: no wave's loads
                else request_pair(PMAXQ - 1);       //  are "behind a branch" that another path of ITS OWN code does not take)
            } else {
                request_pair(PMAXQ - 1);
            }
        }
        typedef const __attribute__((address_space(1))) float* gfloat_p;
        const uint32_t li = min(lane, cnt ? cnt - 1u : 0u), jm = li < full ? j0 + li : js;
        const uint32_t jj = jm < (n_rep + 1u) * HALF ? jm % HALF : 0u;
        eo_c = ((gfloat_p)fcos)[(size_t)rrow_ * HALF + jj];
        eo_s = ((gfloat_p)fsin)[(size_t)rrow_ * HALF + jj];
    }
    __device__ __forceinline__ void before_scores() { before_scores([](int) {}); }
    template <typename Tiles>
    __device__ __forceinline__ void before_scores(Tiles&& tiles)
    {
        const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const uint32_t kv = blockIdx.x % KV;
        const uint32_t PG = (n_rep + 2u) * HALF;
        // ---- the wave's rows from registers (mc_gemv_i4_bfloat_lin{QN}_p1_e4's arithmetic)
        const uint32_t lane_tr = (((lane >> 4) * 4 + (lane & 3)) * 17 + ((lane >> 2) & 3) * 4) * 16;
        const m4b_lane m4bk = m4b_lane_consts(lane);
        typedef __attribute__((address_space(3))) mf_s4 lds_s4;
        uint2 x[QN][WB == 2 ? 4 : 8];
        if constexpr (WB == 2) {
            // (int8: the lane's packet c of a row is weights 16 (64 c + lane) .. + 15: 32 bytes of the row in LDS)
#pragma unroll
            for (int c = 0; c < QN; c++) {
                const uint4 v0 = *reinterpret_cast<const uint4*>(xs + (size_t)(c * 64 + lane) * 32), v1 = *reinterpret_cast<const uint4*>(xs + (size_t)(c * 64 + lane) * 32 + 16);
                x[c][0] = make_uint2(v0.x, v0.y);
                x[c][1] = make_uint2(v0.z, v0.w);
                x[c][2] = make_uint2(v1.x, v1.y);
                x[c][3] = make_uint2(v1.z, v1.w);
            }
        } else if constexpr (WB == 0) {
#pragma unroll
            for (int c = 0; c < QN; c++) {
                lds_s4* xt = (lds_s4*)(xs + c * CHUNK_LDS + lane_tr);
#pragma unroll
                for (int e = 0; e < 8; e++) x[c][e] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(xt + e));
            }
        } else {
            // (bfloat weights: the lane's packet c of a row is weights 8 (64 c + lane) .. + 7: the same packet of the row in LDS)
#pragma unroll
            for (int c = 0; c < QN; c++) {
                const uint4 v = *reinterpret_cast<const uint4*>(xs + (size_t)(c * 64 + lane) * 16);
                x[c][0] = make_uint2(v.x, v.y);
                x[c][1] = make_uint2(v.z, v.w);
            }
        }
        // one row: QN packets into one accumulator, the lane's own element, one wave sum (the stand-alone kernel's order)
        auto row_sum = [&](const uint4 (&w)[QN], const uint32_t (&sc)[QN], uint32_t hi) {
            if constexpr (WB == 2) {
                float a = 0.0f; // gemv.h mac<Q_EXACT> for int8 on bfloat rows, packets in order, one wave sum
#pragma unroll
                for (int c = 0; c < QN; c++) {
                    xregs<BF, 16> xr;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        xr.v[2 * e] = x[c][e].x;
                        xr.v[2 * e + 1] = x[c][e].y;
                    }
                    mac<Q_EXACT>(a, w[c], asf(hi ? (sc[c] & 0xFFFF0000u) : (sc[c] << 16)), xr, 0.0f, static_cast<fmt<WF_I8, BF>*>(nullptr));
                }
                return wave_sum_dpp(a);
            } else if constexpr (WB != 0) {
                float a = 0.0f; // gemv.h mac<WF_T>: four v_dot2_f32_bf16 per packet, packets in order, one wave sum
#pragma unroll
                for (int c = 0; c < QN; c++) {
                    a = dot2(w[c].x, x[c][0].x, a);
                    a = dot2(w[c].y, x[c][0].y, a);
                    a = dot2(w[c].z, x[c][1].x, a);
                    a = dot2(w[c].w, x[c][1].y, a);
                }
                return wave_sum_dpp(a);
            } else {
            mf_f4 acc[1] = {mf_f4{0, 0, 0, 0}};
#pragma unroll
            for (int c = 0; c < QN; c++) mac4b_n<1>(acc, w[c], m4b_prepare(hi ? (sc[c] & 0xFFFF0000u) : (sc[c] << 16), m4bk), x[c]);
            const uint32_t e = lane & 3;
            const float mine = e == 0 ? acc[0][0] : (e == 1 ? acc[0][1] : (e == 2 ? acc[0][2] : acc[0][3]));
            return wave_sum_dpp(mine) * 0x1p37f; // 2^M4B_Q: the sum was formed at 2^-Q (mac4b_n)
            }
        };
        float my_a = 0.0f, my_b = 0.0f;
#pragma unroll
        for (int i = 0; i < PMAXQ; i++) {
            if (!STAGED && (uint32_t)i >= full) break;
            if ((uint32_t)i < full) { // (wave-uniform; no load inside)
                const float ra = row_sum(ww[i][0], wsc[i], 0u), rb = row_sum(ww[i][1], wsc[i], 1u);
                if (lane == (uint32_t)i) {
                    my_a = ra;
                    my_b = rb;
                }
            }
            if constexpr (STAGED) tiles(i);
        }
        if (shared) { // (wave-uniform, and uniform over the workgroup: the barrier below is reached by every wave or by none)
            const float rs = row_sum(ww[PMAXQ - 1][0], wsc[PMAXQ - 1], sh_row);
            if (sh_row != 0u && lane == 0) red[8 + (wave & 3u)] = rs; // (red[0..7]: the rmsnorm's wave sums, long read)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (sh_row == 0u && lane == full) {
                my_a = rs;
                my_b = red[8 + (wave & 3u)];
            }
        }
        // ---- the epilogue of gemv.h finish_pair (EPI_QKV_ROPE), one lane per pair; the pair goes to the cache (k, v: for the
        // steps to come) and, as ONE granule {first | second << 16}, to the workgroups of this kv head (hand-off Q)
        if (lane < cnt) {
            typedef __attribute__((address_space(1))) bf16_t* gS_p;
            const uint32_t j = lane < full ? j0 + lane : js, hq = n_rep * HALF;
            uint32_t g;
            if (j < hq + HALF) {
                const float x1 = BF::rt(my_a), x2 = BF::rt(my_b);
                const bf16_t o1 = BF::st(eo_c * x1 - eo_s * x2), o2 = BF::st(eo_s * x1 + eo_c * x2);
                if (j >= hq) { // (virtual heads of one cache head write the same bits to the same place)
                    gS_p dst = (gS_p)kc + ((size_t)(kv >> kv_shift) * max_seq + slot) * HD;
                    dst[j - hq] = o1;
                    dst[j - hq + HALF] = o2;
                }
                g = (uint32_t)o1 | ((uint32_t)o2 << 16);
            } else {
                const uint32_t d = 2u * (j - hq - HALF);
                const bf16_t va = BF::st(my_a), vb = BF::st(my_b);
                ((gS_p)vt)[((size_t)(kv >> kv_shift) * HD + d) * max_seq + slot] = va;
                ((gS_p)vt)[((size_t)(kv >> kv_shift) * HD + d + 1) * max_seq + slot] = vb;
                g = (uint32_t)va | ((uint32_t)vb << 16);
            }
            unsigned long long* gp = qkv_g + (size_t)kv * PG + j;
            if (fastpath) granule_store_dual(gp, (size_t)KV * PG, tag, g);
            else granule_store(gp, tag, g);
        }
        stamp(2);
        // ---- hand-off Q: the PG pairs of this kv head into LDS (thread t: pair t), natural order
        constexpr int NPASS = WIDE ? 2 : 1; // (up to 512 NPASS pairs per kv head)
        unsigned long long g[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ps++) {
            g[ps] = 0ull;
            if (512u * ps + wave * 64u < PG) {
                const unsigned long long* gp = qkv_g + (size_t)kv * PG + min(512u * ps + tid, PG - 1u);
                handoff_wait w;
                for (uint32_t look = 0;; look++) {
                    g[ps] = fastpath ? granule_look_dual(gp, (size_t)KV * PG, look) : granule_load(gp);
                    const bool ok = (uint32_t)(g[ps] >> 32) == tag;
                    if (__all(ok) || w.expired(st, 0xD0000000u | layer_tag)) break;
                }
            }
        }
        if constexpr (STAGED) tiles(PMAXQ);
#pragma unroll
        for (int ps = 0; ps < NPASS; ps++) {
            const uint32_t t = 512u * ps + tid;
            if (t < PG) {
                const uint32_t hq = n_rep * HALF;
                const bf16_t lo = (bf16_t)(g[ps] & 0xFFFFu), hi = (bf16_t)((g[ps] >> 16) & 0xFFFFu);
                lds_row_w qw_ = (lds_row_w)q_s;
                if (t < hq + HALF) { // q heads of the group, then its k head: [head][HD], element jj and jj + HD / 2
                    const uint32_t hl = t / HALF, jj = t % HALF;
                    qw_[hl * HD + jj] = lo;
                    qw_[hl * HD + jj + HALF] = hi;
                } else { // v: elements 2 jj, 2 jj + 1
                    const uint32_t jj = t - hq - HALF;
                    *(__attribute__((address_space(3))) uint32_t*)(qw_ + (n_rep + 1u) * HD + 2u * jj) = (uint32_t)g[ps];
                }
            }
        }
        __syncthreads();
        stamp(11);
    }
};


// ------------------------------------------------------------------------------------------
// gemma3 (round 5, mc_attn_qkv_wo_qkn_*): wq|wk|wv of a gemma3 block INSIDE the attention launch -- what qkv_in_launch does for Llama, with
// the three things gemma3 adds (include/metalchat/nn/gemma.h:110-137, nn/attention.h:170-177, nn/transformer.h:126-141):
//   * rows of 1.5 KiB (K = 3072, Gemma-7B): a rotation pair of rows is one 3 KiB SUPER ROW swept against the hidden row staged
//     twice in LDS ([x, x]) -- gemv.h LSPLIT, the arithmetic of mc_gemv_i4_bfloat_lin3s_p{1,2}_e0 addition for addition: one accumulator
//     per 1 KiB packet, the middle packet split between the two rows by a lane mask, one wave sum per row;
//   * P2: the previous linear's post-norm, its residual add and this block's pre-norm in the prologue (gemv.h PRO_POSTNORM: h = T(res +
//     T((mu + post_w) x rsqrt(mean(x^2) + eps))), left in HBM by workgroup 0; row = T((mu + norm_w) h rsqrt(mean(h^2) + eps)));
//   * q_norm / k_norm over WHOLE heads before the rotation: the pairs are handed over RAW (T(row sum), as the GEMV's plain store leaves
//     them) and every workgroup of the kv head normalises and rotates the 2 hd values it gathered -- thread t holds pair t of the
//     head's (n_rep + 2) hd / 2, i.e. exactly the (head, j) that mc_rope_kv_T / q_from_qkv_rows give thread t: the same wave_sum tree,
//     the wave sums added in wave order; the workgroup whose range holds the step's slot writes the K row and the V column.
// At most three pairs per wave (Gemma-7B at S = 2048: 384 pairs per kv head over 16 ranges x 8 waves), (n_rep + 2) hd / 2 <= 512.
template <int HD, int P2>
struct qkv_qkn_in_launch {
    static_assert(HD == 256, "gemma3 with head_dim 256 (hd / 2 threads per head are whole waves)");
#ifndef MC_GQ_SCORER_WAVES
#define MC_GQ_SCORER_WAVES 8 // one 16-slot tile per wave on all eight waves (q_from_qkv_rows<., 512> says the same, MC_QKN_SCORER_WAVES: the two forms stay bit for bit); 657 -> 666 tokens/s same box, three alternating runs
#endif
#ifndef MC_GQ_PIN_V
#define MC_GQ_PIN_V 1 // 0 (tuning): the V tiles requested behind the wq|wk|wv phase instead of in front of it
#endif
#ifndef MC_GQ_STAGED
#define MC_GQ_STAGED 1 // 0 (tuning): the K / V tiles requested up front, in front of the wq|wk|wv phase
#endif
#ifndef MC_GQ_V_STEP
#define MC_GQ_V_STEP 3 // the V tiles: 2 = behind the last pair's multiplication, 3 = behind the polls of hand-off Q (a wave's loads return in
                       // order: polls behind 8 KB of V tiles per wave see their granules only when those have arrived)
#endif
    static constexpr bool LDS = true, PIN_V = MC_GQ_PIN_V != 0, STAGED = MC_GQ_STAGED != 0;
    static constexpr int V_STEP = MC_GQ_V_STEP, K_STEPS = 3;
    static constexpr int SCORER_WAVES = MC_GQ_SCORER_WAVES;
    static constexpr int TL_STRIDE = 16, TL_BASE = 3; // stamps: 0 start, 1 row staged, 2 pairs published, 3.. attn_fused_bf's 0.., 11 rows normalised
    static constexpr uint32_t KQ = 3072u, NPK = KQ / 8, ROWB2 = 3072u, CHUNK_LDS = 2048 * 2 / 16 * 17, HALF = HD / 2, WPH = HALF / 64;
    static constexpr int PMAXQ = 3, NCH = 3;
    typedef const __attribute__((address_space(3))) bf16_t* lds_row;
    typedef __attribute__((address_space(3))) bf16_t* lds_row_w;
    typedef uint32_t rowv4 __attribute__((ext_vector_type(4)));
    lds_row q_s, k_s, v_s;
    // ---- what the launch was given
    const void *xp, *normp, *postp, *resp, *qw, *qs;
    void* h_out;
    const bf16_t *q_norm, *k_norm;
    const float *fcos, *fsin;
    bf16_t *kc, *vt;
    unsigned long long* qkv_g;
    step_state* st;
    char* xs;   // [x, x] in LDS: three chunks of 2048, padded for the transposed reads
    float* red; // 32 floats of scratch
    uint32_t n_rep, KV, max_seq, nsplit, group, layer_tag, fastpath, split_slots;
    float eps, mu;
    unsigned long long* tl;
    // ---- what at_start() leaves for the later phases
    rowv4 xr, nr, pwr, rrr;
    uint4 ww[PMAXQ][NCH];
    uint32_t wsc[PMAXQ][NCH];
    uint32_t nq0, nq1, nk0, nk1; // (the norm weights' bits, zero-extended: 2-byte members of a policy passed by value went through scratch)
    float pc, ps;
    uint32_t j0, full, slot, tag, rrow_;

    __device__ __forceinline__ void stamp(int i) const
    {
        if (tl && threadIdx.x == 0) tl[(size_t)blockIdx.x * TL_STRIDE + i] = __builtin_amdgcn_s_memrealtime();
    }
    // packed pair of pair j of kv head `kv`: q heads of the group, then its k head, then its v head (rotation partners adjacent)
    __device__ __forceinline__ uint32_t pp_of(uint32_t kv, uint32_t j) const
    {
        const uint32_t hq = n_rep * HALF, H = n_rep * KV;
        return j < hq ? kv * hq + j : (j < hq + HALF ? H * HALF + kv * HALF + (j - hq) : (H + KV) * HALF + kv * HALF + (j - hq - HALF));
    }
    // (gemv.h LSPLIT) the lane's 32 weights of packet c belong to the super row's SECOND row when they lie past 3072
    static __device__ __forceinline__ bool second_half(int c, uint32_t lane) { return 2048u * (uint32_t)c + 32u * lane >= 3072u; }
    // pair i of the wave: its 3 KiB of weights and the scales of the lane's groups (every load unconditional; a pair the wave does not
    // have reads one broadcast line: masks, not selects -- gemv.h ltile)
    __device__ __forceinline__ void request_pair(int i)
    {
        const uint32_t lane = threadIdx.x & 63, kv = blockIdx.x % KV;
        const uint32_t glog = group ? 31u - __builtin_clz(group) : 31u, ngroups = group ? KQ >> glog : 1u;
        const uint32_t lm = 0u - (uint32_t)((uint32_t)i < full ? 1u : 0u);
        const uint64_t lm64 = ((uint64_t)lm << 32) | lm;
        const uint32_t pp = pp_of(kv, j0 + i) & lm;
        const char* wrow = static_cast<const char*>(qw) + (((uint64_t)pp * ROWB2) & lm64) + ((lane * 16) & lm);
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const rowv4 v = __builtin_nontemporal_load(reinterpret_cast<const rowv4*>(wrow + (((size_t)c * 1024) & lm64)));
            ww[i][c] = make_uint4(v.x, v.y, v.z, v.w);
        }
        // scales: row quads [ceil(out / 4)][ngroups][4] bf16; the dword (rows 2 pp, 2 pp + 1) of the lane's group
        const char* srow = static_cast<const char*>(qs) + (((size_t)(pp >> 1) * ngroups) * 4 + (pp & 1u) * 2) * 2;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const uint32_t ks = 2048u * (uint32_t)c + 32u * lane, k = ks >= KQ ? ks - KQ : ks;
            const uint32_t g = group ? (k >> glog) : 0u;
            wsc[i][c] = *reinterpret_cast<const uint32_t*>(srow + ((g * 8u) & lm));
        }
    }
    __device__ __forceinline__ void at_start()
    {
        const uint32_t tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const uint32_t pk = min(tid, NPK - 1u); // (384 packets: threads 384.. read the last one again and add nothing)
        xr = reinterpret_cast<const rowv4*>(xp)[pk];
        nr = reinterpret_cast<const rowv4*>(normp)[pk];
        if constexpr (P2 != 0) {
            pwr = reinterpret_cast<const rowv4*>(postp)[pk];
            rrr = reinterpret_cast<const rowv4*>(resp)[pk];
        }
        stamp(0);
        const __attribute__((address_space(1))) int32_t* stp = (const __attribute__((address_space(1))) int32_t*)st;
        slot = (uint32_t)stp[3];
        rrow_ = (uint32_t)stp[6];
        tag = (uint32_t)stp[9] * 256u + layer_tag;
        const uint32_t j = tid % HALF;
        nq0 = q_norm[j]; nq1 = q_norm[j + HALF];
        nk0 = k_norm[j]; nk1 = k_norm[j + HALF];
        asm volatile("s_barrier" ::: "memory"); // (the row's requests stay ahead of the weight requests in the CU's memory pipe)
        const uint32_t split = blockIdx.x / KV;
        const uint32_t PG = (n_rep + 2u) * HALF, PW = PG / nsplit, e = PW >> 3, r = PW & 7u;
        full = e + (wave < r ? 1u : 0u);
        j0 = split * PW + wave * e + min(wave, r);
        request_pair(0);
    }
    __device__ __forceinline__ void before_tiles()
    {
        const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const bool live = tid < NPK;
        // ---- the prologue of mc_gemv_i4_bfloat_lin3s_p{1,2}_* (gemv.h, the build-time prologue): the same sums in the same order
        auto sumsq = [&](const rowv4& v) {
            const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
            float s1 = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float a = asf(vv[e] << 16), b = asf(vv[e] & 0xFFFF0000u);
                s1 += a * a;
                s1 += b * b;
            }
            return live ? s1 : 0.0f;
        };
        auto normalise = [&](const rowv4& v, const rowv4& w, float inv) {
            const uint32_t vv[4] = {v.x, v.y, v.z, v.w}, wv[4] = {w.x, w.y, w.z, w.w};
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float a = (mu + asf(wv[e] << 16)) * asf(vv[e] << 16) * inv;
                const float b = (mu + asf(wv[e] & 0xFFFF0000u)) * asf(vv[e] & 0xFFFF0000u) * inv;
                o[e] = pack_bf16x2(a, b);
            }
            return rowv4{o[0], o[1], o[2], o[3]};
        };
        rowv4 row;
        {
            const float w1 = wave_sum_dpp(sumsq(xr));
            if (lane == 0) red[wave] = w1;
            lds_barrier(); // (LDS only: the first pairs stay in flight)
            float tot = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) tot += red[i];
            const float inv1 = 1.0f / sqrtf(tot / (float)KQ + eps);
            if constexpr (P2 != 0) {
                const rowv4 y = normalise(xr, pwr, inv1);
                const uint32_t aa[4] = {rrr.x, rrr.y, rrr.z, rrr.w}, bb[4] = {y.x, y.y, y.z, y.w};
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; e++)
                    o[e] = pack_bf16x2(asf(aa[e] << 16) + asf(bb[e] << 16), asf(aa[e] & 0xFFFF0000u) + asf(bb[e] & 0xFFFF0000u));
                const rowv4 h = live ? rowv4{o[0], o[1], o[2], o[3]} : rowv4{0, 0, 0, 0};
                if (blockIdx.x == 0 && live) ((__attribute__((address_space(1))) rowv4*)h_out)[tid] = h;
                const float w2 = wave_sum_dpp(sumsq(h));
                if (lane == 0) red[16 + wave] = w2;
                lds_barrier(); // (not __syncthreads(): behind workgroup 0's store it would wait for every pair requested so far)
                float tot2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; i++) tot2 += red[16 + i];
                const float inv2 = 1.0f / sqrtf(tot2 / (float)KQ + eps);
                row = normalise(h, nr, inv2);
            } else {
                row = normalise(xr, nr, inv1);
            }
        }
        if (live) { // the row twice, [x, x]; packet p sits in slot p + p / 16
            rowv4* xl = reinterpret_cast<rowv4*>(xs);
            xl[tid + (tid >> 4)] = row;
            xl[(NPK + tid) + ((NPK + tid) >> 4)] = row;
        }
        lds_barrier();
        stamp(1);
        // ---- the rest of the wave's pairs, then the table row of this thread's rotation (its address waits for the step state)
        request_pair(1);
        request_pair(2);
        typedef const __attribute__((address_space(1))) float* gfloat_p;
        pc = ((gfloat_p)fcos)[(size_t)rrow_ * HALF + tid % HALF];
        ps = ((gfloat_p)fsin)[(size_t)rrow_ * HALF + tid % HALF];
    }
    __device__ __forceinline__ void before_scores() { before_scores([](int) {}); }
    // tiles(step): attn_fused_bf's tile requests (STAGED), placed behind the multiplication of pair `step`: the memory pipe stays full
    // while the wave is never held at issue for long
    template <typename Tiles>
    __device__ __forceinline__ void before_scores(Tiles&& tiles)
    {
        const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const uint32_t kv = blockIdx.x % KV, split = blockIdx.x / KV;
        const uint32_t PG = (n_rep + 2u) * HALF;
        // ---- the wave's super rows from registers (mc_gemv_i4_bfloat_lin3s_*'s arithmetic)
        const uint32_t lane_tr = (((lane >> 4) * 4 + (lane & 3)) * 17 + ((lane >> 2) & 3) * 4) * 16;
        const m4b_lane m4bk = m4b_lane_consts(lane);
        typedef __attribute__((address_space(3))) mf_s4 lds_s4;
        uint2 x[NCH][8];
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            lds_s4* xt = (lds_s4*)(xs + c * CHUNK_LDS + lane_tr);
#pragma unroll
            for (int e = 0; e < 8; e++) x[c][e] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(xt + e));
        }
        float my_a = 0.0f, my_b = 0.0f;
#pragma unroll
        for (int i = 0; i < PMAXQ; i++) {
            if ((uint32_t)i < full) { // (wave-uniform; no load inside)
            float m[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const uint32_t dw = wsc[i][c];
                const uint32_t sfl = second_half(c, lane) ? (dw & 0xFFFF0000u) : (dw << 16);
                mf_f4 one[1] = {mf_f4{0, 0, 0, 0}};
                mac4b_n<1>(one, ww[i][c], m4b_prepare(sfl, m4bk), x[c]);
                const uint32_t e = lane & 3;
                m[c] = e == 0 ? one[0][0] : (e == 1 ? one[0][1] : (e == 2 ? one[0][2] : one[0][3]));
            }
            // packets 0 and 2 are whole rows' worth, the middle one is the first row's in lanes 0..31 and the second row's in lanes 32..63
            const float fa = wave_sum_dpp((m[0] + (lane < 32 ? m[1] : 0.0f)) * 0x1p37f);
            const float fb = wave_sum_dpp((m[2] + (lane < 32 ? 0.0f : m[1])) * 0x1p37f);
            if (lane == (uint32_t)i) {
                my_a = fa;
                my_b = fb;
            }
            }
#ifdef MC_GQ_STAMPS
            stamp(12 + i);
#endif
            tiles(i);
        }
        // ---- the pair RAW (the plain store of the GEMV: T(row sum)) as ONE granule to the workgroups of this kv head (hand-off Q)
        if (lane < full) {
            const uint32_t j = j0 + lane;
            const uint32_t g = (uint32_t)BF::st(my_a) | ((uint32_t)BF::st(my_b) << 16);
            unsigned long long* gp = qkv_g + (size_t)kv * PG + j;
            if (fastpath) granule_store_dual(gp, (size_t)KV * PG, tag, g);
            else granule_store(gp, tag, g);
        }
        stamp(2);
        // ---- hand-off Q: thread t gathers pair t of this kv head
        unsigned long long g = 0ull;
        if (wave * 64u < PG) {
            const unsigned long long* gp = qkv_g + (size_t)kv * PG + min(tid, PG - 1u);
            handoff_wait w;
            for (uint32_t look = 0;; look++) {
                g = fastpath ? granule_look_dual(gp, (size_t)KV * PG, look) : granule_load(gp);
                const bool ok = (uint32_t)(g >> 32) == tag;
                if (__all(ok) || w.expired(st, 0xD0000000u | layer_tag)) break;
            }
        }
        tiles(PMAXQ);
        // ---- q_norm / k_norm + rotation of the head this thread's pair belongs to (mc_rope_kv_T / q_from_qkv_rows::one_pass: thread
        // (head hl, pair j) = thread hl * hd / 2 + j), the V pairs as they are; the cache rows by the workgroup whose range holds the slot
        {
            const uint32_t hl = tid / HALF, j = tid % HALF;
            const bool is_q = hl < n_rep, live = hl < n_rep + 1u, is_v = hl == n_rep + 1u;
            const bool writer = slot / split_slots == split;
            float x1 = asf((uint32_t)g << 16), x2 = asf((uint32_t)g & 0xFFFF0000u);
            if (!live) x1 = x2 = 0.0f;
            const float v = wave_sum(x1 * x1 + x2 * x2);
            if (lane == 0) red[24 + wave] = v;
            lds_barrier();
            float tot = 0.0f;
#pragma unroll
            for (uint32_t i = 0; i < WPH; i++) tot += red[24 + (tid / HALF) * WPH + i];
            const float inv = 1.0f / sqrtf(tot / (float)HD + eps);
            x1 = BF::rt((mu + bf2f((bf16_t)(is_q ? nq0 : nk0))) * x1 * inv);
            x2 = BF::rt((mu + bf2f((bf16_t)(is_q ? nq1 : nk1))) * x2 * inv);
            const bf16_t o1 = BF::st(pc * x1 - ps * x2), o2 = BF::st(ps * x1 + pc * x2);
            lds_row_w qw_ = (lds_row_w)q_s;
            if (live) {
                qw_[hl * HD + j] = o1; // (the K row sits behind the n_rep query heads: k_s = q_s + n_rep * HD)
                qw_[hl * HD + j + HALF] = o2;
                if (!is_q && writer) {
                    typedef __attribute__((address_space(1))) bf16_t* gS_p;
                    gS_p dst = (gS_p)kc + ((size_t)kv * max_seq + slot) * HD;
                    dst[j] = o1;
                    dst[j + HALF] = o2;
                }
            } else if (is_v) {
                const bf16_t va = (bf16_t)(g & 0xFFFFu), vb = (bf16_t)((g >> 16) & 0xFFFFu);
                *(__attribute__((address_space(3))) uint32_t*)(qw_ + (n_rep + 1u) * HD + 2u * j) = (uint32_t)g;
                if (writer) {
                    typedef __attribute__((address_space(1))) bf16_t* gS_p;
                    ((gS_p)vt)[((size_t)kv * HD + 2u * j) * max_seq + slot] = va;
                    ((gS_p)vt)[((size_t)kv * HD + 2u * j + 1u) * max_seq + slot] = vb;
                }
            }
        }
        lds_barrier(); // (behind the cache write: __syncthreads() would drain the vector-memory counter, i.e. wait for the tiles)
        stamp(11);
    }
};

// LNCH = KiB of packed weights per Wo row (K = H * hd = 2048 LNCH); QN != 0: wq|wk|wv (rows of QN KiB) in this launch too -- qx
// WB = 1: plain bfloat weights for Wo and wq|wk|wv (rows of LNCH / QN KiB = 512 LNCH / 512 QN weights; qkv_in_launch above)
// TT: 64-slot score tiles per scoring wave (attn_fused_bf's T): 1 = 64 cache slots per workgroup, 4 = 256 (S = 8192 with one workgroup per CU)
// QKN = 1 (round 5, mc_attn_wo_qkn_*: gemma3): `q` holds the RAW wq|wk|wv rows of the step (the GEMV's plain store); q_norm / k_norm, the rotation
// and the cache write happen in this launch (decode_kernels.hip q_from_qkv_rows<HD, 512>) -- `qnorm_w` = q_norm, `qkv_w` = k_norm
// QKN = 2 / 3 (mc_attn_qkv_wo_qkn_*): gemma3 with wq|wk|wv in the launch too (qkv_qkn_in_launch above; 3: the post-norm prologue) -- `res` = the
// row handed to the block, `qnorm_w` = attention_norm, `gx` = the pointers gemma3 adds
struct gemma_extra {
    const void *post_w, *res_row; // (QKN = 3) the previous linear's post-norm and the residual it adds to
    void* h_out;                  // (QKN = 3) where workgroup 0 leaves that hidden row
    const bf16_t *q_norm, *k_norm;
};
// CH = 1 (round 6, mc_attn_qkv_wo_w13_w_*): a phase follows in the same launch -- every stored pair of the output row is also published as a
// {2 x bf16, tag} granule in `hid_g` (hand-off D)
template <int HD, int LNCH, int QN = 0, int WB = 0, int TT = 1, int QKN = 0, int CH = 0>
__device__ __forceinline__ void
attn_wo_body(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vt, bf16_t* __restrict__ attn_out,
             unsigned long long* psum_g, unsigned long long* slab_g, unsigned long long* row_g, step_state* st, uint32_t n_rep,
             uint32_t KV, uint32_t max_seq, float scale, uint32_t nsplit, uint32_t layer_tag, const void* __restrict__ wo_w,
             const void* __restrict__ wo_s, const bf16_t* res, bf16_t* y, uint32_t out_rows, uint32_t group, uint32_t has_res,
             uint32_t fastpath, unsigned long long* tl,
             // QN != 0: the hidden row is `res`; its norm weight, the packed wq|wk|wv matrix and its scales, the rope tables, the
             // granules of hand-off Q
             const void* __restrict__ qnorm_w = nullptr, const void* __restrict__ qkv_w = nullptr, const void* __restrict__ qkv_s = nullptr,
             const float* fcos = nullptr, const float* fsin = nullptr, unsigned long long* qkv_g = nullptr, float eps = 0.0f, float mu = 0.0f,
             uint32_t kv_shift = 0, gemma_extra gx = gemma_extra(), unsigned long long* hid_g = nullptr)
{
    static_assert(CH == 0 || (QN != 0 && QKN == 0), "a chained phase: behind a llama launch with wq|wk|wv inside");
    constexpr uint32_t K = WB == 1 ? 512u * LNCH : (WB == 2 ? 1024u * LNCH : 2048u * LNCH);
    constexpr uint32_t WPK = WB == 1 ? 8u : (WB == 2 ? 16u : 32u); // weights of a lane's 16-byte packet
    constexpr uint32_t CHUNK_LDS = 2048 * 2 / 16 * 17;  // a chunk of the row in LDS: 16 bytes of padding per 256 (gemv.h Q_M4D)
    constexpr uint32_t ROWB = 1024u * LNCH;              // bytes of weights per row
    // (QN != 0: the hidden row of the wq|wk|wv phase first, the attention row of the Wo phase later: hand-off C lies between them)
    constexpr int QCH = QKN >= 2 ? 3 : QN; // chunks of the row of the wq|wk|wv phase (gemma3: [x, x] = three chunks of 2048)
    __shared__ __attribute__((aligned(16))) char xs[(LNCH > QCH ? LNCH : QCH) * (WB == 1 ? 1024u : (WB == 2 ? 2048u : CHUNK_LDS))];
    static_assert(QN == 0 || QKN == 0, "one source of the step's queries");
    __shared__ float qred[(QN || QKN) ? 32 : 1];
    __shared__ __attribute__((aligned(16))) bf16_t qkv_rows[(QN || QKN) ? 18 * HD : 8]; // queries of up to 16 heads, the K row, the V row
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (QN != 0) the hidden row and the wave's wq|wk|wv pairs are requested by the first instructions of the launch
    typedef qkv_in_launch<HD, QN ? QN : (WB ? 4 : 2), WB> qx_t;
    qx_t qx;
    if constexpr (QN != 0) {
        typedef typename qx_t::lds_row lds_row;
        qx.q_s = (lds_row)qkv_rows;
        qx.k_s = (lds_row)qkv_rows + n_rep * HD;
        qx.v_s = (lds_row)qkv_rows + (n_rep + 1u) * HD;
        qx.xp = res; qx.normp = qnorm_w; qx.qw = qkv_w; qx.qs = qkv_s; qx.fcos = fcos; qx.fsin = fsin;
        qx.kc = const_cast<bf16_t*>(kc); qx.vt = const_cast<bf16_t*>(vt); qx.qkv_g = qkv_g; qx.st = st; qx.xs = xs; qx.red = qred;
        qx.n_rep = n_rep; qx.KV = KV; qx.max_seq = max_seq; qx.nsplit = nsplit; qx.group = group; qx.layer_tag = layer_tag;
        qx.fastpath = fastpath; qx.eps = eps; qx.mu = mu; qx.tl = tl; qx.kv_shift = kv_shift;
        qx.at_start();
    }
    typedef qkv_qkn_in_launch<(QKN >= 2 ? HD : 256), QKN == 3> qg_t;
    qg_t qg;
    if constexpr (QKN >= 2) {
        typedef typename qg_t::lds_row lds_row;
        qg.q_s = (lds_row)qkv_rows;
        qg.k_s = (lds_row)qkv_rows + n_rep * HD;
        qg.v_s = (lds_row)qkv_rows + (n_rep + 1u) * HD;
        qg.xp = res; qg.normp = qnorm_w; qg.postp = gx.post_w; qg.resp = gx.res_row; qg.h_out = gx.h_out; qg.qw = qkv_w; qg.qs = qkv_s;
        qg.q_norm = gx.q_norm; qg.k_norm = gx.k_norm; qg.fcos = fcos; qg.fsin = fsin; qg.kc = const_cast<bf16_t*>(kc); qg.vt = const_cast<bf16_t*>(vt);
        qg.qkv_g = qkv_g; qg.st = st; qg.xs = xs; qg.red = qred; qg.n_rep = n_rep; qg.KV = KV; qg.max_seq = max_seq; qg.nsplit = nsplit;
        qg.group = group; qg.layer_tag = layer_tag; qg.fastpath = fastpath; qg.split_slots = 64u * TT; qg.eps = eps; qg.mu = mu; qg.tl = tl;
        qg.at_start();
    }
    typedef q_from_qkv_rows<(QKN == 1 ? HD : 128), 512> qn_t;
    qn_t qn;
    if constexpr (QKN == 1) {
        typedef typename qn_t::lds_row lds_row;
        qn.q_s = (lds_row)qkv_rows;
        qn.k_s = (lds_row)qkv_rows + n_rep * HD;
        qn.v_s = (lds_row)qkv_rows + (n_rep + 1u) * HD;
        qn.red = qred; qn.qkv = q; qn.q_norm = static_cast<const bf16_t*>(qnorm_w); qn.k_norm = static_cast<const bf16_t*>(qkv_w);
        qn.fcos = fcos; qn.fsin = fsin; qn.kc = const_cast<bf16_t*>(kc); qn.vt = const_cast<bf16_t*>(vt); qn.st = st; qn.n_rep = n_rep; qn.KV = KV;
        qn.max_seq = max_seq; qn.split_slots = 64u * TT; qn.eps = eps; qn.mu = mu;
        qn.at_start();
    }

    // ---- the Wo row pairs of this wave: contiguous spans dealt as the linear-order kernels deal theirs, at most PMAX each (the
    // host takes this kernel only then).  Their weights are requested BEHIND the scores (attn_fused_bf's hook): at the start of the
    // launch they queue in front of the K tile in the CU's memory pipe (measured: 720 tokens/s against 732 with the Wo GEMV as a
    // launch of its own); behind the scores the hand-offs of the attention cover them.
    constexpr int PMAX = WB ? 1 : 2; // (bfloat / int8 weights: a pair is 2 LNCH KiB = 32 registers at LNCH = 4)
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
    // With wq|wk|wv in the launch (QN != 0) the K and V tiles are in long before the scores: every wave requests its pairs in front
    // of the scores (point -1), whose arithmetic (~ 1 us) the 34 KB per CU then have to themselves, and hand-off A is not queued
    // behind anybody's weights.
#ifndef MC_QKV_WO_REQ_EARLY
#define MC_QKV_WO_REQ_EARLY 0
#endif
    auto request_wo = [&](int point) {
        const bool poller = wave < n_rep; // (attn_fused_bf: head = wave, wave + NW, ... gathers the denominators of hand-off A)
#ifndef MC_QKN_WO_EARLY
#define MC_QKN_WO_EARLY 0 // 1 (tuning): the gemma3 form requests its Wo pairs in front of the scores
#endif
        if ((QN != 0 && MC_QKV_WO_REQ_EARLY == 1) || (QKN == 1 && MC_QKN_WO_EARLY == 1)) {
            if (point != -1) return;
        } else if (QN != 0 && MC_QKV_WO_REQ_EARLY == 2) {
            // the waves that compute no scores ask in front of them (they are idle there), the others behind hand-off A
            if (point != (poller ? 1 : -1)) return;
        } else if (point == -1 || (MC_WO_REQ_SPLIT ? (point == 0) == poller : point != 0)) return;
        // (every load unconditional: one load behind a branch and hipcc waits vmcnt(0) wherever it waits afterwards -- the scores
        //  would wait for these weights.  A pair the wave does not have reads one broadcast line: masks, not selects)
        const bf16_t* rp = has_res ? res : y; // (a pointer that can be read either way; the value is masked below)
#pragma unroll
        for (int i = 0; i < PMAX; i++) {
            const uint32_t lm = 0u - (uint32_t)(pb + i < pe ? 1u : 0u);
            const uint64_t lm64 = ((uint64_t)lm << 32) | lm;
            const uint32_t pr = (pb + i) & lm;
            const char* wrow = static_cast<const char*>(wo_w) + (((uint64_t)pr * 2 * ROWB) & lm64) + ((lane * 16) & lm);
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int c = 0; c < LNCH; c++) {
                    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + (((size_t)r * ROWB + c * 1024) & lm64)));
                    ww[i][r][c] = make_uint4(v.x, v.y, v.z, v.w);
                }
            if constexpr (WB != 1) {
                // scales: row quads [ceil(out / 4)][ngroups][4] bf16 (gemv.h); the dword (rows 2 pr, 2 pr + 1) of the lane's group
                const char* srow = static_cast<const char*>(wo_s) + (((size_t)(pr >> 1) * ngroups) * 4 + (pr & 1u) * 2) * 2;
#pragma unroll
                for (int c = 0; c < LNCH; c++) {
                    const uint32_t g = group ? ((64u * WPK * c + WPK * lane) >> glog) : 0u;
                    ws[i][c] = *reinterpret_cast<const uint32_t*>(srow + ((g * 8u) & lm));
                }
            }
            wres[i] = reinterpret_cast<const uint32_t*>(rp)[pr] & (has_res ? 0xFFFFFFFFu : 0u);
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
    constexpr int TL_STRIDE = (QN || QKN >= 2) ? 16 : 8, TL_BASE = (QN || QKN >= 2) ? 3 : 0;
    if constexpr (QN != 0) {
        attn_fused_bf<HD, TT, 8>(q, kc, vt, psum_g, slab_g, st, n_rep, KV, max_seq, scale, nsplit, layer_tag, tl, publish, request_wo, fastpath, qx, kv_shift);
    } else if constexpr (QKN >= 2) {
        attn_fused_bf<HD, TT, 8>(nullptr, kc, vt, psum_g, slab_g, st, n_rep, KV, max_seq, scale, nsplit, layer_tag, tl, publish, request_wo, fastpath, qg);
    } else if constexpr (QKN != 0) {
        attn_fused_bf<HD, TT, 8>(nullptr, kc, vt, psum_g, slab_g, st, n_rep, KV, max_seq, scale, nsplit, layer_tag, tl, publish, request_wo, fastpath, qn);
    } else {
        static_assert(QN != 0 || TT == 1, "wide ranges: with the step's rows in LDS only (every tile of the step's slot is patched from there)");
        attn_fused_bf<HD, 1, 8>(q, kc, vt, psum_g, slab_g, st, n_rep, KV, max_seq, scale, nsplit, layer_tag, tl, publish, request_wo, fastpath);
    }
    // tl != null (tools/attn_wo_timeline.py only): thread 0 of every workgroup leaves s_memrealtime stamps of its phases
    auto stamp = [&](int i) {
        if (tl && threadIdx.x == 0) tl[(size_t)blockIdx.x * TL_STRIDE + TL_BASE + i] = __builtin_amdgcn_s_memrealtime();
    };

    // ---- while the attention row is on its way: everything of the wave's FIRST Wo pair that does not need it -- the dequantisation
    // (gemv.h m4b_dequant: ~ 60 % of the pair's instructions).  The weights were requested two hand-offs ago; the wait of hand-off C
    // (~ 2.5 us) covers this.  64 registers; a second pair (contexts of 1024 slots) is dequantised behind the hand-off as before.
#ifndef MC_WO_PREDEQUANT
#define MC_WO_PREDEQUANT 1
#endif
    const m4b_lane m4bk = m4b_lane_consts(lane);
    uint2 dq[2][WB ? 1 : LNCH][8];
    if (WB == 0 && MC_WO_PREDEQUANT && pb < pe) { // (wave-uniform)
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < LNCH; c++) {
                if constexpr (WB == 0) {
                    const uint32_t raw = ws[0][c];
                    m4b_dequant(dq[r][c], ww[0][r][c], m4b_prepare(r ? (raw & 0xFFFF0000u) : (raw << 16), m4bk));
                }
            }
    }
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
            if constexpr (WB == 0) *reinterpret_cast<uint32_t*>(xs + (p + (p >> 4)) * 16 + (g & 3u) * 4) = val[i];
            else *reinterpret_cast<uint32_t*>(xs + g * 4) = val[i]; // (natural order)
        }
    }
    __syncthreads();
    stamp(6);

    // ---- Wo: mc_gemv_i4_bfloat_lin{LNCH}_p0_e{0,1} for the wave's pairs
    const uint32_t lane_tr = (((lane >> 4) * 4 + (lane & 3)) * 17 + ((lane >> 2) & 3) * 4) * 16;
    typedef __attribute__((address_space(3))) mf_s4 lds_s4;
#pragma unroll
    for (int i = 0; i < PMAX; i++) {
        if (pb + i >= pe) break;
        float rsum[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if constexpr (WB == 2) {
                float a = 0.0f; // gemv.h mac<Q_EXACT> for int8: packets in order, one wave sum
#pragma unroll
                for (int c = 0; c < LNCH; c++) {
                    xregs<BF, 16> xr;
                    xr.load(xs + (size_t)(c * 64 + lane) * 32, 0);
                    const uint32_t raw = ws[i][c];
                    mac<Q_EXACT>(a, ww[i][r][c], asf(r ? (raw & 0xFFFF0000u) : (raw << 16)), xr, 0.0f, static_cast<fmt<WF_I8, BF>*>(nullptr));
                }
                rsum[r] = wave_sum_dpp(a);
            } else if constexpr (WB != 0) {
                float a = 0.0f; // gemv.h mac<WF_T>: four v_dot2_f32_bf16 per packet, packets in order, one wave sum
#pragma unroll
                for (int c = 0; c < LNCH; c++) {
                    const uint4 xv = *reinterpret_cast<const uint4*>(xs + (size_t)(c * 64 + lane) * 16);
                    const uint4 wv = ww[i][r][c];
                    a = dot2(wv.x, xv.x, a);
                    a = dot2(wv.y, xv.y, a);
                    a = dot2(wv.z, xv.z, a);
                    a = dot2(wv.w, xv.w, a);
                }
                rsum[r] = wave_sum_dpp(a);
            } else {
            mf_f4 acc[1] = {mf_f4{0, 0, 0, 0}};
#pragma unroll
            for (int c = 0; c < LNCH; c++) {
                lds_s4* xt = (lds_s4*)(xs + c * CHUNK_LDS + lane_tr);
                uint2 x[8];
#pragma unroll
                for (int e = 0; e < 8; e++) x[e] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(xt + e));
                const uint32_t raw = ws[i][c];
                if (MC_WO_PREDEQUANT && i == 0) m4b_dot(acc[0], dq[r][c], x);
                else mac4b_n<1>(acc, ww[i][r][c], m4b_prepare(r ? (raw & 0xFFFF0000u) : (raw << 16), m4bk), x);
            }
            const uint32_t e = lane & 3;
            const float mine = e == 0 ? acc[0][0] : (e == 1 ? acc[0][1] : (e == 2 ? acc[0][2] : acc[0][3]));
            rsum[r] = wave_sum_dpp(mine) * 0x1p37f; // 2^M4B_Q: the sum was formed at 2^-Q (mac4b_n)
            }
        }
        if (lane == 0) {
            float va = BF::rt(rsum[0]), vb = BF::rt(rsum[1]);
            if (has_res) { // add in T (kernel/arithmetic.metal:13-46)
                va = __uint_as_float(wres[i] << 16) + va;
                vb = __uint_as_float(wres[i] & 0xFFFF0000u) + vb;
            }
            reinterpret_cast<uint32_t*>(y)[pb + i] = pack_bf16x2(va, vb);
            if constexpr (CH != 0) granule_store(hid_g + pb + i, epoch_tag, pack_bf16x2(va, vb)); // hand-off D
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
// ... with wq|wk|wv in the launch too (qkv_in_launch above): one launch from the hidden row to the hidden row
#define MC_ATTN_QKV_WO(NAME, HD, LNCH, QN, TT)                                                                                            \
    extern "C" __global__ void __launch_bounds__(512)                                                                                    \
    NAME(const bf16_t* kc, const bf16_t* vt, bf16_t* attn_out, unsigned long long* psum_g, unsigned long long* slab_g,                  \
         unsigned long long* row_g, unsigned long long* qkv_g, step_state* st, uint32_t n_rep, uint32_t n_kv, uint32_t max_seq,         \
         float scale, uint32_t nsplit, uint32_t layer_tag, const void* wo_w, const void* wo_s, const bf16_t* x, bf16_t* y,              \
         uint32_t out_rows, uint32_t group, const void* qnorm_w, const void* qkv_w, const void* qkv_s, const float* fcos, const float* fsin, float eps,    \
         float mu, uint32_t fastpath, unsigned long long* tl, uint32_t kv_shift)                                                         \
    {                                                                                                                                    \
        attn_wo_body<HD, LNCH, QN, 0, TT>(nullptr, kc, vt, attn_out, psum_g, slab_g, row_g, st, n_rep, n_kv, max_seq, scale, nsplit,     \
                                          layer_tag, wo_w, wo_s, x, y, out_rows, group, 1u, fastpath, tl, qnorm_w, qkv_w, qkv_s, fcos,   \
                                          fsin, qkv_g, eps, mu, kv_shift);                                                               \
    }
// mc_attn_qkv_wo_i4_bfloat_hd{head_dim}_k{KiB per Wo row}_q{KiB per wq|wk|wv row}
MC_ATTN_QKV_WO(mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2, 128, 2, 2, 1)  // Llama-3-8B: dim 4096, 32 heads x 128
// ... with 128- and 256-slot ranges (round 5: contexts of 4096 and 8192 slots in one 512-thread workgroup per CU, as the int8 launch below): _t{tiles}
MC_ATTN_QKV_WO(mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2_t2, 128, 2, 2, 2)
MC_ATTN_QKV_WO(mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2_t4, 128, 2, 2, 4)
// ... WITHOUT Wo (round 5, Llama-3-70B: 64 heads x 128 make Wo rows of 4 KiB -- two pairs per wave are 64 registers requested behind the
// scores, measured slower than the GEMV twice, above): attention_norm + wq|wk|wv (rows of 4 KiB, K = 8192) + rope + cache write + the decode
// attention in one launch, the row left in HBM for the Wo GEMV: mc_attn_qkv_i4_bfloat_hd{head_dim}_q{KiB per wq|wk|wv row}
template <int HD, int QN>
__device__ __forceinline__ void
attn_qkv_body(const bf16_t* kc, const bf16_t* vt, bf16_t* attn_out, unsigned long long* psum_g, unsigned long long* slab_g, unsigned long long* qkv_g,
              step_state* st, uint32_t n_rep, uint32_t KV, uint32_t max_seq, float scale, uint32_t nsplit, uint32_t layer_tag, const bf16_t* x,
              uint32_t group, const void* norm_w, const void* qkv_w, const void* qkv_s, const float* fcos, const float* fsin, float eps, float mu,
              uint32_t fastpath, unsigned long long* tl)
{
    constexpr uint32_t CHUNK_LDS = 2048 * 2 / 16 * 17;
    __shared__ __attribute__((aligned(16))) char xs[QN * CHUNK_LDS];
    __shared__ float qred[16];
    __shared__ __attribute__((aligned(16))) bf16_t qkv_rows[18 * HD];
    typedef qkv_in_launch<HD, QN, 0> qx_t;
    typedef typename qx_t::lds_row lds_row;
    qx_t qx;
    qx.q_s = (lds_row)qkv_rows;
    qx.k_s = (lds_row)qkv_rows + n_rep * HD;
    qx.v_s = (lds_row)qkv_rows + (n_rep + 1u) * HD;
    qx.xp = x; qx.normp = norm_w; qx.qw = qkv_w; qx.qs = qkv_s; qx.fcos = fcos; qx.fsin = fsin;
    qx.kc = const_cast<bf16_t*>(kc); qx.vt = const_cast<bf16_t*>(vt); qx.qkv_g = qkv_g; qx.st = st; qx.xs = xs; qx.red = qred;
    qx.n_rep = n_rep; qx.KV = KV; qx.max_seq = max_seq; qx.nsplit = nsplit; qx.group = group; qx.layer_tag = layer_tag;
    qx.fastpath = fastpath; qx.eps = eps; qx.mu = mu; qx.tl = tl; qx.kv_shift = 0;
    qx.at_start();
    // one rounding of the fp32 sum (bmm.metal:80): the attention row the Wo GEMV reads
    auto store = [&](uint32_t head, uint32_t db, uint32_t col, float v) {
        if ((threadIdx.x & 63) < 16) attn_out[(size_t)head * HD + db * 16 + col] = f2bf(v);
    };
    attn_fused_bf<HD, 1, 8>(nullptr, kc, vt, psum_g, slab_g, st, n_rep, KV, max_seq, scale, nsplit, layer_tag, tl, store, [](int) {}, fastpath, qx, 0u);
}
extern "C" __global__ void __launch_bounds__(512)
mc_attn_qkv_i4_bfloat_hd128_q4(const bf16_t* kc, const bf16_t* vt, bf16_t* attn_out, unsigned long long* psum_g, unsigned long long* slab_g,
                               unsigned long long* qkv_g, step_state* st, uint32_t n_rep, uint32_t n_kv, uint32_t max_seq, float scale, uint32_t nsplit,
                               uint32_t layer_tag, const bf16_t* x, uint32_t group, const void* norm_w, const void* qkv_w, const void* qkv_s,
                               const float* fcos, const float* fsin, float eps, float mu, uint32_t fastpath, unsigned long long* tl)
{
    attn_qkv_body<128, 4>(kc, vt, attn_out, psum_g, slab_g, qkv_g, st, n_rep, n_kv, max_seq, scale, nsplit, layer_tag, x, group, norm_w, qkv_w, qkv_s,
                          fcos, fsin, eps, mu, fastpath, tl);
}
// ... with plain bfloat weights (nn::linear): mc_attn_qkv_wo_w_bfloat_hd{head_dim}_k{KiB per Wo row}_q{KiB per wq|wk|wv row}
#define MC_ATTN_QKV_WO_W(NAME, HD, LNCH, QN, TT)                                                                                          \
    extern "C" __global__ void __launch_bounds__(512)                                                                                    \
    NAME(const bf16_t* kc, const bf16_t* vt, bf16_t* attn_out, unsigned long long* psum_g, unsigned long long* slab_g,                  \
         unsigned long long* row_g, unsigned long long* qkv_g, step_state* st, uint32_t n_rep, uint32_t n_kv, uint32_t max_seq,         \
         float scale, uint32_t nsplit, uint32_t layer_tag, const void* wo_w, const void* wo_s, const bf16_t* x, bf16_t* y,              \
         uint32_t out_rows, uint32_t group, const void* qnorm_w, const void* qkv_w, const void* qkv_s, const float* fcos, const float* fsin, float eps,    \
         float mu, uint32_t fastpath, unsigned long long* tl, uint32_t kv_shift)                                                         \
    {                                                                                                                                    \
        attn_wo_body<HD, LNCH, QN, 1, TT>(nullptr, kc, vt, attn_out, psum_g, slab_g, row_g, st, n_rep, n_kv, max_seq, scale, nsplit,     \
                                          layer_tag, wo_w, wo_s, x, y, out_rows, group, 1u, fastpath, tl, qnorm_w, qkv_w, qkv_s, fcos,   \
                                          fsin, qkv_g, eps, mu, kv_shift);                                                               \
    }
MC_ATTN_QKV_WO_W(mc_attn_qkv_wo_w_bfloat_hd64_k4_q4, 64, 4, 4, 1)  // Llama-3.2-1B: dim 2048, 32 heads x 64, bf16 weights
MC_ATTN_QKV_WO_W(mc_attn_qkv_wo_w_bfloat_hd64_k4_q4_t2, 64, 4, 4, 2)  // ... S = 4096 (round 5: 128-slot ranges)
MC_ATTN_QKV_WO_W(mc_attn_qkv_wo_w_bfloat_hd64_k4_q4_t4, 64, 4, 4, 4)  // ... S = 8192 (256-slot ranges)
// ... with int8 weights: mc_attn_qkv_wo_i8_bfloat_hd{head_dim}_k{KiB per Wo row}_q{KiB per wq|wk|wv row}_t{64-slot tiles per scoring wave}
#define MC_ATTN_QKV_WO_I8(NAME, HD, LNCH, QN, TT)                                                                                         \
    extern "C" __global__ void __launch_bounds__(512)                                                                                    \
    NAME(const bf16_t* kc, const bf16_t* vt, bf16_t* attn_out, unsigned long long* psum_g, unsigned long long* slab_g,                  \
         unsigned long long* row_g, unsigned long long* qkv_g, step_state* st, uint32_t n_rep, uint32_t n_kv, uint32_t max_seq,         \
         float scale, uint32_t nsplit, uint32_t layer_tag, const void* wo_w, const void* wo_s, const bf16_t* x, bf16_t* y,              \
         uint32_t out_rows, uint32_t group, const void* qnorm_w, const void* qkv_w, const void* qkv_s, const float* fcos, const float* fsin, float eps,    \
         float mu, uint32_t fastpath, unsigned long long* tl, uint32_t kv_shift)                                                         \
    {                                                                                                                                    \
        attn_wo_body<HD, LNCH, QN, 2, TT>(nullptr, kc, vt, attn_out, psum_g, slab_g, row_g, st, n_rep, n_kv, max_seq, scale, nsplit,     \
                                          layer_tag, wo_w, wo_s, x, y, out_rows, group, 1u, fastpath, tl, qnorm_w, qkv_w, qkv_s, fcos,   \
                                          fsin, qkv_g, eps, mu, kv_shift);                                                               \
    }
MC_ATTN_QKV_WO_I8(mc_attn_qkv_wo_i8_bfloat_hd128_k4_q4_t1, 128, 4, 4, 1)  // Llama-3-8B int8, S <= 2048
MC_ATTN_QKV_WO_I8(mc_attn_qkv_wo_i8_bfloat_hd128_k4_q4_t2, 128, 4, 4, 2)  // ... S = 4096: 128-slot ranges
MC_ATTN_QKV_WO_I8(mc_attn_qkv_wo_i8_bfloat_hd128_k4_q4_t4, 128, 4, 4, 4)  // ... S = 8192: 256-slot ranges, one workgroup per CU

// ... gemma3 (round 5): q_norm / k_norm + rotation + cache write (mc_rope_kv_T) + attention + Wo in one launch, from the raw wq|wk|wv rows:
// mc_attn_wo_qkn_i4_bfloat_hd{head_dim}_k{KiB per Wo row}_t{64-slot tiles per scoring wave}
#define MC_ATTN_WO_QKN(NAME, HD, LNCH, TT)                                                                                               \
    extern "C" __global__ void __launch_bounds__(512)                                                                                    \
    NAME(const bf16_t* qkv, const bf16_t* kc, const bf16_t* vt, bf16_t* attn_out, unsigned long long* psum_g, unsigned long long* slab_g, \
         unsigned long long* row_g, step_state* st, uint32_t n_rep, uint32_t n_kv, uint32_t max_seq, float scale, uint32_t nsplit,      \
         uint32_t layer_tag, const void* wo_w, const void* wo_s, const bf16_t* res, bf16_t* y, uint32_t out_rows, uint32_t group,       \
         uint32_t has_res, uint32_t fastpath, unsigned long long* tl, const bf16_t* q_norm, const bf16_t* k_norm, const float* fcos,     \
         const float* fsin, float eps, float mu)                                                                                         \
    {                                                                                                                                    \
        attn_wo_body<HD, LNCH, 0, 0, TT, 1>(qkv, kc, vt, attn_out, psum_g, slab_g, row_g, st, n_rep, n_kv, max_seq, scale, nsplit,       \
                                            layer_tag, wo_w, wo_s, res, y, out_rows, group, has_res, fastpath, tl, q_norm, k_norm,       \
                                            nullptr, fcos, fsin, nullptr, eps, mu);                                                      \
    }
// ... and with wq|wk|wv in the launch (qkv_qkn_in_launch): mc_attn_qkv_wo_qkn_i4_bfloat_hd{head_dim}_k{KiB per Wo row}_p{1: pre-norm, 2: post-norm +
// residual + pre-norm}_t{tiles}; rows of 1.5 KiB (K = 3072)
#define MC_ATTN_QKV_WO_QKN(NAME, HD, LNCH, P, TT)                                                                                        \
    extern "C" __global__ void __launch_bounds__(512)                                                                                    \
    NAME(const bf16_t* kc, const bf16_t* vt, bf16_t* attn_out, unsigned long long* psum_g, unsigned long long* slab_g,                  \
         unsigned long long* row_g, unsigned long long* qkv_g, step_state* st, uint32_t n_rep, uint32_t n_kv, uint32_t max_seq,         \
         float scale, uint32_t nsplit, uint32_t layer_tag, const void* wo_w, const void* wo_s, const bf16_t* x, bf16_t* y,              \
         uint32_t out_rows, uint32_t group, const void* norm_w, const void* qkv_w, const void* qkv_s, const float* fcos,                \
         const float* fsin, float eps, float mu, uint32_t fastpath, unsigned long long* tl, const bf16_t* q_norm, const bf16_t* k_norm, \
         const void* post_w, const void* res_row, void* h_out)                                                                           \
    {                                                                                                                                    \
        attn_wo_body<HD, LNCH, 0, 0, TT, 1 + P>(nullptr, kc, vt, attn_out, psum_g, slab_g, row_g, st, n_rep, n_kv, max_seq, scale,       \
                                                nsplit, layer_tag, wo_w, wo_s, x, y, out_rows, group, 0u, fastpath, tl, norm_w, qkv_w,   \
                                                qkv_s, fcos, fsin, qkv_g, eps, mu, 0u,                                                   \
                                                gemma_extra{post_w, res_row, h_out, q_norm, k_norm});                                    \
    }
MC_ATTN_QKV_WO_QKN(mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p1_t1, 256, 2, 1, 1)  // Gemma-7B at S = 1024 (16 ranges of 64 slots): the first block (or parity taps)
MC_ATTN_QKV_WO_QKN(mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p2_t1, 256, 2, 2, 1)  // ... every other block
MC_ATTN_QKV_WO_QKN(mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p1_t2, 256, 2, 1, 2)  // Gemma-7B at S = 2048: the first block (or parity taps)
MC_ATTN_QKV_WO_QKN(mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p2_t2, 256, 2, 2, 2)  // ... every other block
MC_ATTN_QKV_WO_QKN(mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p1_t4, 256, 2, 1, 4)  // Gemma-7B at S = 4096 (256-slot ranges)
MC_ATTN_QKV_WO_QKN(mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p2_t4, 256, 2, 2, 4)
MC_ATTN_WO_QKN(mc_attn_wo_qkn_i4_bfloat_hd256_k2_t1, 256, 2, 1)  // Gemma-7B shapes up to S = 1024
MC_ATTN_WO_QKN(mc_attn_wo_qkn_i4_bfloat_hd256_k2_t2, 256, 2, 2)  // ... S = 2048: 16 ranges of 128 slots x 16 kv heads = one workgroup per CU
MC_ATTN_WO_QKN(mc_attn_wo_qkn_i4_bfloat_hd256_k2_t4, 256, 2, 4)  // ... S = 4096

// mc_attn_wo_i4_bfloat_hd{head_dim}_k{KiB per Wo row}
MC_ATTN_WO(mc_attn_wo_i4_bfloat_hd128_k2, 128, 2)  // Llama-3-8B: 32 heads x 128
// (hd128_k4 -- Llama-3-70B's 64 heads, 131 KB of Wo per CU -- was built and bit-identical too, and SLOWER than the two launches:
//  20.1 us against 9.3 + 9.7, profiles/r03_kernel_stats_70b*.csv: what this kernel gains is the Wo weights waiting in registers
//  when the row arrives, and two row pairs of 4 KiB rows per wave are 64 registers requested behind the scores, not 16)
#ifdef MC_ATTN_WO_K4
MC_ATTN_WO(mc_attn_wo_i4_bfloat_hd128_k4, 128, 4)  // tuning builds (-DMC_ATTN_WO_K4 + env MC_ATTN_WO_K4): Llama-3-70B.  Round 4, with the XCD-local hand-offs: 20.17 us against 9.15 + 9.80, 124.4 / 125.2 against 126.8 / 124.7 tokens/s
#endif
MC_ATTN_WO(mc_attn_wo_i4_bfloat_hd64_k1, 64, 1)    // 32 heads x 64
MC_ATTN_WO(mc_attn_wo_i4_bfloat_hd256_k2, 256, 2)  // Gemma-7B shapes: 16 heads x 256

// ------------------------------------------------------------------------------------------
// Round 6: the attention block AND ffn_norm + w1|w3 + SiLU * mul (include/metalchat/nn/transformer.h:130-137, 53-59) in one launch, for PLAIN
// bfloat weights (nn::linear; TinyLlama-1.1B, Llama-3.2-1B): mc_attn_qkv_wo_w13_w_bfloat_hd64_k4_q4_f{3p3,4p4}.
//
// The same chain was built three ways for the int4 headline first and lost every time (tools/experiments/README.md, round 6): with every weight
// on chip the exact int4 dequantisation alone takes the eight waves 6.6 - 8 us per workgroup, as long as the stand-alone launch needs to stream
// the matrix beside it.  Plain bfloat weights have no such floor -- four v_dot2_f32_bf16 per 16-byte packet -- so here a chained phase costs
// what it cannot have on chip when the row arrives, and the stand-alone launch's boundary + prologue + cold start (~ 3 us of its 9.6) go.
// What those builds' timelines taught is how to spend hand-off D's ~ 4 us (the hidden row from 1024 Wo pairs on 256 CUs to every CU):
//   * nothing is requested during the attention phases (a loader beside them sits in the polling CUs' own memory queues);
//   * behind the Wo phase waves 4-7 (the FETCHERS) request their F row pairs -- 8 KiB each, into registers -- and poll nothing, while waves 0-3
//     (the POLLERS) wait for the row; a wave's vector-memory results return in issue order, so the pollers request their pairs only
//     when the row is in, and multiply them as they arrive; the fetchers' F pairs are then long in;
//   * the loads of this phase are inline asm with hand-counted waits: one request site for both kinds of wave (reached at different
//     times), no load behind a branch.
//
// Numerics: bit for bit the two launches it replaces.  The attention block is the same code (attn_wo_body, CH = 1: its Wo epilogue also
// publishes the pair as a granule).  The w1|w3 phase is mc_gemv_w_bfloat_ling4_p1_e2's arithmetic (gemv.h LGEN): packet p's sum of squares
// formed by one thread (p < 256), the sums of packets 64 v .. 64 v + 63 added by wave_sum_dpp in lane order, the eight wave sums in order
// (the last four are zeros there too), the normalised row staged in natural order; a row = its four packets in order through four
// v_dot2_f32_bf16 each into ONE fp32 sum, one wave sum; the epilogue of gemv.h finish_pair (EPI_SILU_MUL).  Which wave multiplies a pair
// does not change a bit of it.
// ------------------------------------------------------------------------------------------
namespace {

// stamps (tl2 != null: tools/attn_w13_timeline.py), per workgroup, s_memrealtime:
//   40 Wo done (thread 0)   41 hand-off D: row gathered (thread 0)   42 row staged   44 wave 0 (poller) stored   45 wave 4 (fetcher) stored
constexpr int TL2_STRIDE = 48;

typedef uint32_t w13_v4 __attribute__((ext_vector_type(4)));
// Loads the compiler does not count (its own counted waits stay exact for ITS loads; these are waited for by hand, below)
__device__ __forceinline__ void w13_load16(w13_v4& dst, const void* p)
{
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
}
// everything but the N youngest vector-memory operations of this wave has completed; the values pass THROUGH the wait, so no use of them can be scheduled in front of it
template <int N> __device__ __forceinline__ void w13_wait(w13_v4& a, w13_v4& b, w13_v4& c, w13_v4& d)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}

// F: pairs of each fetcher wave; the pollers share what is left of the workgroup's pairs (the host: at most P each)
template <int HD, int F, int P>
__device__ __forceinline__ void
attn_qkv_wo_w13_w_body(const bf16_t* kc, const bf16_t* vt, bf16_t* attn_out, unsigned long long* psum_g, unsigned long long* slab_g, unsigned long long* row_g,
                       unsigned long long* qkv_g, step_state* st, uint32_t n_rep, uint32_t KV, uint32_t max_seq, float scale, uint32_t nsplit, uint32_t layer_tag,
                       const void* wo_w, const void* wo_s, const bf16_t* x, bf16_t* y, uint32_t out_rows, uint32_t group, const void* qnorm_w,
                       const void* qkv_w, const void* qkv_s, const float* fcos, const float* fsin, float eps, float mu, uint32_t fastpath,
                       unsigned long long* tl, uint32_t kv_shift, unsigned long long* hid_g, const void* __restrict__ w13_w, const void* __restrict__ ffn_norm,
                       bf16_t* __restrict__ gate, uint32_t ffn_rows, unsigned long long* tl2)
{
    constexpr uint32_t KF = 2048u, ROWB = 4096u; // dim 2048: rows of 4 KiB, 256 packets of the hidden row
    constexpr int PK = 4;                        // packets (KiB) per row
    constexpr int PM = F > P ? F : P;            // pairs a wave may hold
    __shared__ __attribute__((aligned(16))) char xs13[KF * 2];
    __shared__ float red13[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long* mytl = tl2 ? tl2 + (size_t)blockIdx.x * TL2_STRIDE : nullptr;

    // ================= the attention block ...
    attn_wo_body<HD, 4, 4, 1, 1, 0, 1>(nullptr, kc, vt, attn_out, psum_g, slab_g, row_g, st, n_rep, KV, max_seq, scale, nsplit, layer_tag, wo_w, wo_s, x, y,
                                       out_rows, group, 1u, fastpath, tl, qnorm_w, qkv_w, qkv_s, fcos, fsin, qkv_g, eps, mu, kv_shift, gemma_extra(), hid_g);
    auto stamp2 = [&](int i, uint32_t t) {
        if (mytl && tid == t) mytl[i] = __builtin_amdgcn_s_memrealtime();
    };
    stamp2(40, 0);
    // ================= ... and ffn_norm + w1|w3 + SiLU * mul
    const bool poller = wave < 4u;
    // the workgroup's pairs: equal contiguous shares of the launch (a remainder to the first workgroups); the fetchers F each, the pollers the rest
    const uint32_t NP = ffn_rows / 2, G = gridDim.x, wq_ = NP / G, wrem = NP - wq_ * G;
    const uint32_t nb = wq_ + (blockIdx.x < wrem ? 1u : 0u), sb0 = blockIdx.x * wq_ + min(blockIdx.x, wrem);
    const uint32_t nf = min(nb, 4u * (uint32_t)F), rest = nb - nf, w4 = wave & 3u;
    const uint32_t pb = poller ? sb0 + nf + w4 * (rest >> 2) + min(w4, rest & 3u) : sb0 + min(nf, w4 * (uint32_t)F);
    const uint32_t cnt = poller ? (rest >> 2) + (w4 < (rest & 3u) ? 1u : 0u) : min((uint32_t)F, nf - min(nf, w4 * (uint32_t)F));
    // ---- hand-off D (pollers): the hidden row the Wo phases of ALL workgroups finished; thread t < 256 gathers packet t = granules 4 t .. 4 t + 3
    // (one watched granule and ~ 0.4 us between looks while the row is not there, then one sweep: hand-off C)
    const uint32_t epoch_tag = st->epoch * 256u + layer_tag;
    w13_v4 xr = w13_v4{0, 0, 0, 0}, nr = w13_v4{0, 0, 0, 0};
    if (poller) { // (wave-uniform; every load inside is waited for inside)
        // the norm weights of the packet this thread stages: asked for in front of the wait (behind it they sat behind the workgroup's own
        // weight requests: 2.2 us from the first sight of the row to "row staged", profiles/r06_chainw_timeline_tiny_first_build.log)
        nr = reinterpret_cast<const w13_v4*>(ffn_norm)[tid];
        uint32_t val[4];
        handoff_wait w;
        for (;;) {
            const bool seen = (uint32_t)(granule_load(hid_g + 4u * tid + 3u) >> 32) == epoch_tag;
            if (__all(seen) || w.expired(st, 0xE0000000u | layer_tag)) break;
            __builtin_amdgcn_s_sleep(MC_HANDOFF_C_SLEEP);
        }
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned long long g = granule_load(hid_g + 4u * tid + i);
                ok = ok && (uint32_t)(g >> 32) == epoch_tag;
                val[i] = (uint32_t)g;
            }
            if (__all(ok) || w.expired(st, 0xE0000000u | layer_tag)) break;
        }
        xr = w13_v4{val[0], val[1], val[2], val[3]};
    }
    stamp2(41, 0);
    // ---- every wave's requests, in ONE place (the fetchers reach it ~ 4 us before the pollers): the wave's pairs (a pair the wave does not
    // have reads one broadcast line: masks, not selects)
    w13_v4 wreg[PM][2 * PK];
#pragma unroll
    for (int i = 0; i < PM; i++) {
        const uint32_t lm = 0u - (uint32_t)((uint32_t)i < cnt ? 1u : 0u);
        const uint64_t lm64 = ((uint64_t)lm << 32) | lm;
        const char* a = static_cast<const char*>(w13_w) + (((uint64_t)(pb + (uint32_t)i) * 2u * ROWB) & lm64) + ((lane * 16u) & lm);
#pragma unroll
        for (int t = 0; t < 2 * PK; t++) w13_load16(wreg[i][t], a + ((t * 1024u) & lm));
    }
    // (issued behind this point: nothing until the epilogue's store -- the counts below are exact)
    constexpr int NW = PM * 2 * PK; // weight loads, the youngest
    // ---- ffn_norm on the way into LDS (kernel/rmsnorm.metal:52-95; gemv.h LGEN, the build-time prologue: the same additions in the same order)
    {
        const uint32_t vv[4] = {xr.x, xr.y, xr.z, xr.w};
        float s1 = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float a = asf(vv[e] << 16), b = asf(vv[e] & 0xFFFF0000u);
            s1 += a * a;
            s1 += b * b;
        }
        float ss = 0.0f;
        ss += poller ? s1 : 0.0f; // (the stand-alone kernel's threads 256 .. 511 add nothing)
        const float wsum_ = wave_sum_dpp(ss);
        if (lane == 0) red13[wave] = wsum_;
        lds_barrier();
        float tot = 0.0f;
#pragma unroll
        for (uint32_t i = 0; i < 8u; i++) tot += red13[i];
        const float inv = 1.0f / sqrtf(tot / (float)KF + eps);
        if (poller) {
            const uint32_t wv[4] = {nr.x, nr.y, nr.z, nr.w};
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float a = (mu + asf(wv[e] << 16)) * asf(vv[e] << 16) * inv;
                const float b = (mu + asf(wv[e] & 0xFFFF0000u)) * asf(vv[e] & 0xFFFF0000u) * inv;
                o[e] = pack_bf16x2(a, b);
            }
            reinterpret_cast<w13_v4*>(xs13)[tid] = w13_v4{o[0], o[1], o[2], o[3]}; // (natural order)
        }
    }
    lds_barrier();
    stamp2(42, 0);
    // ---- the lane's slices of the row, once per wave: packet 64 c + lane of chunk c
    uint4 xq[PK];
#pragma unroll
    for (int c = 0; c < PK; c++) xq[c] = *reinterpret_cast<const uint4*>(xs13 + (size_t)(c * 64 + (int)lane) * 16);
    // ---- the wave's pairs as they arrive: a row = its four packets in order (gemv.h mac<WF_T>: four v_dot2_f32_bf16 per packet), one wave sum
    float my_a = 0.0f, my_b = 0.0f;
    auto pair_step = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        w13_wait<NW - 2 * PK * (i + 1)>(wreg[i][0], wreg[i][1], wreg[i][2], wreg[i][3]);
        w13_wait<NW - 2 * PK * (i + 1)>(wreg[i][4], wreg[i][5], wreg[i][6], wreg[i][7]);
        if ((uint32_t)i < cnt) { // (wave-uniform; no load inside)
            float rs[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                float a = 0.0f;
#pragma unroll
                for (int c = 0; c < PK; c++) {
                    const w13_v4 wv = wreg[i][PK * r + c];
                    a = dot2(wv.x, xq[c].x, a);
                    a = dot2(wv.y, xq[c].y, a);
                    a = dot2(wv.z, xq[c].z, a);
                    a = dot2(wv.w, xq[c].w, a);
                }
                rs[r] = wave_sum_dpp(a);
            }
            if (lane == (uint32_t)i) {
                my_a = rs[0];
                my_b = rs[1];
            }
        }
    };
    pair_step(std::integral_constant<int, 0>{});
    if constexpr (PM > 1) pair_step(std::integral_constant<int, 1>{});
    if constexpr (PM > 2) pair_step(std::integral_constant<int, 2>{});
    if constexpr (PM > 3) pair_step(std::integral_constant<int, 3>{});
    if constexpr (PM > 4) pair_step(std::integral_constant<int, 4>{});
    if constexpr (PM > 5) pair_step(std::integral_constant<int, 5>{});
    static_assert(F >= 1 && PM <= 6, "the steps above");
    // ---- the epilogue of gemv.h finish_pair (EPI_SILU_MUL), one lane per pair: out[j] = T(silu(T(w1 x)) * T(w3 x))
    if (lane < cnt) {
        const float ga = BF::rt(my_a), gb = BF::rt(my_b);
        const float g = mc::gemv::silu_T<BF>(ga);
        gate[pb + lane] = BF::st(g * gb);
    }
    stamp2(44, 0);
    stamp2(45, 256);
}

} // namespace

// mc_attn_qkv_wo_w13_w_bfloat_hd{head_dim}_k{KiB per Wo row}_q{KiB per wq|wk|wv row}_f{pairs of w1|w3 per fetcher wave}p{most pairs of a poller wave}
#define MC_ATTN_QKV_WO_W13_W(NAME, HD, F, P)                                                                                             \
    extern "C" __global__ void __launch_bounds__(512)                                                                                    \
    NAME(const bf16_t* kc, const bf16_t* vt, bf16_t* attn_out, unsigned long long* psum_g, unsigned long long* slab_g,                  \
         unsigned long long* row_g, unsigned long long* qkv_g, step_state* st, uint32_t n_rep, uint32_t n_kv, uint32_t max_seq,         \
         float scale, uint32_t nsplit, uint32_t layer_tag, const void* wo_w, const void* wo_s, const bf16_t* x, bf16_t* y,              \
         uint32_t out_rows, uint32_t group, const void* qnorm_w, const void* qkv_w, const void* qkv_s, const float* fcos, const float* fsin, float eps,    \
         float mu, uint32_t fastpath, unsigned long long* tl, uint32_t kv_shift, unsigned long long* hid_g, const void* w13_w,           \
         const void* ffn_norm, bf16_t* gate, uint32_t ffn_rows, unsigned long long* tl2)                                                 \
    {                                                                                                                                    \
        attn_qkv_wo_w13_w_body<HD, F, P>(kc, vt, attn_out, psum_g, slab_g, row_g, qkv_g, st, n_rep, n_kv, max_seq, scale, nsplit, layer_tag, wo_w, wo_s, x, y, \
                                         out_rows, group, qnorm_w, qkv_w, qkv_s, fcos, fsin, eps, mu, fastpath, tl, kv_shift, hid_g, w13_w, ffn_norm, gate, \
                                         ffn_rows, tl2);                                                                                 \
    }
// How many pairs the fetchers take decides how long hand-off D lasts (its polls drain behind them in the CU's memory queue) against how much is left to
// stream behind it.  Same box, alternating, tokens/s: TinyLlama (22 pairs per workgroup) F = 1 / 2 / 3 / 4 / 5: 1498 / 1534 / 1578-1603 / 1560-1567 / 1507-1520;
// Llama-3.2-1B (32 pairs) F = 2 / 3 / 4 / 5: 1620 / 1650 / 1661 / 1643 (profiles/r06_ab_chain_f*.log) -- the even deal wins: every wave the same number of pairs.
MC_ATTN_QKV_WO_W13_W(mc_attn_qkv_wo_w13_w_bfloat_hd64_k4_q4_f3p3, 64, 3, 3) // TinyLlama-1.1B: 22 pairs per workgroup = 4 x 3 + (3, 3, 2, 2)
MC_ATTN_QKV_WO_W13_W(mc_attn_qkv_wo_w13_w_bfloat_hd64_k4_q4_f4p4, 64, 4, 4) // Llama-3.2-1B:   32 pairs per workgroup = 4 x 4 + 4 x 4
