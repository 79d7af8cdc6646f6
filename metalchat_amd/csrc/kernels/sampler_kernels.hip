// Sampler chain: the step after the decode path.
//
// Part A -- the reference's kernels under their own names and argument order, launched through
// the encoder ABI exactly like the Metal ones:
//   sort_T          kernel/sort.metal:33-88         (launch: include/metalchat/kernel/sort.h:27-62)
//   cumsum_B_T      kernel/cumsum.metal:24-76       (launch: include/metalchat/kernel/sum.h:28-58)
//   multinomial_T   kernel/multinomial.metal:60-122
//   gt_T / le_T     kernel/logical.metal:13-68      (bool = one byte)
//   scatter_T       kernel/copy.metal:45-74
//   gather_T        kernel/copy.metal:77-113        (T = bfloat, float, int32_t)
//   sub_T / div_T   kernel/arithmetic.metal:88-157
//   sum_T           kernel/sum.metal:22-74          (launch: include/metalchat/kernel/sum.h:88-117)
//
// Part B -- mc_topk_candidates_T + mc_sample_T: make_default_sampler (nn/sampling.h:303-313,
// topk(50) -> nucleus(0.6, 0.9) -> multinomial(1)) fused into two launches on the decode stream,
// with no host round trip (the reference synchronises three times per token and partial_sorts
// the whole vocabulary on the CPU, nn/sampling.h:244-264).
#include "common.h"

using namespace mc;

#ifndef MC_IJ
#define MC_IJ                                                          \
    const uint32_t i = blockIdx.y * blockDim.y + threadIdx.y;          \
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
#endif

// ---------------------------------------------------------------------------------- sub / gt / le
template <typename T>
__device__ __forceinline__ void
sub_body(const layout2& ol, typename T::S* out, const layout2& al, const typename T::S* a,
         const layout2& bl, const typename T::S* b)
{
    MC_IJ;
    if (i < al.sizes[0] && k < al.sizes[1])
        out[at(ol, i, k)] = T::st(T::ld(a[at(al, i, k)]) - T::ld(b[at(bl, i, k)]));
}
extern "C" __global__ void
sub_bfloat(layout2 ol, bf16_t* out, layout2 al, const bf16_t* a, layout2 bl, const bf16_t* b)
{
    sub_body<BF>(ol, out, al, a, bl, b);
}
extern "C" __global__ void
sub_float(layout2 ol, float* out, layout2 al, const float* a, layout2 bl, const float* b)
{
    sub_body<F32>(ol, out, al, a, bl, b);
}

template <typename T>
__device__ __forceinline__ void
div_body(const layout2& ol, typename T::S* out, const layout2& al, const typename T::S* a,
         const layout2& bl, const typename T::S* b)
{
    MC_IJ;
    if (i < al.sizes[0] && k < al.sizes[1])
        out[at(ol, i, k)] = T::st(T::ld(a[at(al, i, k)]) / T::ld(b[at(bl, i, k)]));
}
extern "C" __global__ void
div_bfloat(layout2 ol, bf16_t* out, layout2 al, const bf16_t* a, layout2 bl, const bf16_t* b)
{
    div_body<BF>(ol, out, al, a, bl, b);
}
extern "C" __global__ void
div_float(layout2 ol, float* out, layout2 al, const float* a, layout2 bl, const float* b)
{
    div_body<F32>(ol, out, al, a, bl, b);
}

// sum of every row in fp32, one workgroup per row, thread t adds its contiguous slice of
// `block` elements, then the two-level wavefront reduction (kernel/sum.metal:37-70)
template <typename T>
__device__ __forceinline__ void
sum_body(const layout1& ol, typename T::S* out, const layout2& il, const typename T::S* in, uint32_t block)
{
    __shared__ float red[16];
    const uint32_t dim = il.sizes[1], i = blockIdx.x;
    const uint32_t begin = threadIdx.x * block, end = begin + block;
    float s = 0.0f;
    for (uint32_t j = begin; j < end && j < dim; j++) s += T::ld(in[at(il, i, j)]);
    const float tot = block_sum(s, red);
    if (threadIdx.x == 0) out[at(ol, i)] = T::st(tot);
}
extern "C" __global__ void
sum_bfloat(layout1 ol, bf16_t* out, layout2 il, const bf16_t* in, uint32_t block)
{
    sum_body<BF>(ol, out, il, in, block);
}
extern "C" __global__ void
sum_float(layout1 ol, float* out, layout2 il, const float* in, uint32_t block)
{
    sum_body<F32>(ol, out, il, in, block);
}

#define MC_LOGICAL(NAME, OP)                                                                     \
    extern "C" __global__ void NAME##_bfloat(layout2 ol, uint8_t* out, layout2 il, const bf16_t* in, \
                                             bf16_t value)                                       \
    {                                                                                            \
        MC_IJ;                                                                                   \
        if (i < il.sizes[0] && k < il.sizes[1]) out[at(ol, i, k)] = bf2f(in[at(il, i, k)]) OP bf2f(value); \
    }                                                                                            \
    extern "C" __global__ void NAME##_float(layout2 ol, uint8_t* out, layout2 il, const float* in, \
                                            float value)                                         \
    {                                                                                            \
        MC_IJ;                                                                                   \
        if (i < il.sizes[0] && k < il.sizes[1]) out[at(ol, i, k)] = in[at(il, i, k)] OP value;   \
    }
MC_LOGICAL(gt, >)
MC_LOGICAL(le, <=)

// ---------------------------------------------------------------------------------- scatter / gather
extern "C" __global__ void
scatter_bfloat(layout2 ol, bf16_t* out, layout2 ml, const uint8_t* mask, bf16_t value)
{
    MC_IJ;
    if (i < ol.sizes[0] && k < ol.sizes[1] && mask[at(ml, i, k)]) out[at(ol, i, k)] = value;
}
extern "C" __global__ void
scatter_float(layout2 ol, float* out, layout2 ml, const uint8_t* mask, float value)
{
    MC_IJ;
    if (i < ol.sizes[0] && k < ol.sizes[1] && mask[at(ml, i, k)]) out[at(ol, i, k)] = value;
}

template <typename S>
__device__ __forceinline__ void
gather_body(const layout2& ol, S* out, const layout2& il, const S* in, const layout2& xl, const int32_t* index)
{
    MC_IJ;
    if (i < xl.sizes[0] && k < xl.sizes[1]) out[at(ol, i, k)] = in[at(il, i, (uint32_t)index[at(xl, i, k)])];
}
extern "C" __global__ void
gather_bfloat(layout2 ol, bf16_t* out, layout2 il, const bf16_t* in, layout2 xl, const int32_t* index)
{
    gather_body(ol, out, il, in, xl, index);
}
extern "C" __global__ void
gather_float(layout2 ol, float* out, layout2 il, const float* in, layout2 xl, const int32_t* index)
{
    gather_body(ol, out, il, in, xl, index);
}
extern "C" __global__ void
gather_int32_t(layout2 ol, int32_t* out, layout2 il, const int32_t* in, layout2 xl, const int32_t* index)
{
    gather_body(ol, out, il, in, xl, index);
}

// ---------------------------------------------------------------------------------- sort
// One workgroup per row; thread t owns slots [t*block, (t+1)*block) of the power-of-two padded
// row, which lives in the OUTPUT buffers (global memory) like the reference's, so rows of any
// length sort (the 128k-entry vocabulary row is 131072 slots, 17*18/2 = 153 barrier stages).
template <typename T>
__device__ __forceinline__ void
sort_body(const layout2& vl, typename T::S* values, const layout2& xl, int32_t* indices,
          const layout2& il, const typename T::S* in, uint32_t block)
{
    const uint32_t dim = il.sizes[1], aligned = vl.sizes[1], b = blockIdx.x;
    const uint32_t begin = threadIdx.x * block, end = begin + block;
    for (uint32_t k = begin; k < end; k++) {
        values[at(vl, b, k)] = k < dim ? in[at(il, b, k)] : T::st(-INFINITY);
        indices[at(xl, b, k)] = (int32_t)k;
    }
    for (uint32_t k = 2; k <= aligned; k *= 2) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            __syncthreads(); // workgroup-scope release/acquire of the global stores above
            for (uint32_t i = begin; i < end; i++) {
                const uint32_t ij = i ^ j;
                if (i < ij) {
                    const size_t pi = at(vl, b, i), pj = at(vl, b, ij);
                    const typename T::S si = values[pi], sj = values[pj];
                    const float vi = T::ld(si), vj = T::ld(sj);
                    const bool up = (i & k) == 0;
                    if ((up && vi < vj) || (!up && vi > vj)) {
                        values[pi] = sj;
                        values[pj] = si;
                        const size_t qi = at(xl, b, i), qj = at(xl, b, ij);
                        const int32_t t = indices[qi];
                        indices[qi] = indices[qj];
                        indices[qj] = t;
                    }
                }
            }
        }
    }
}
extern "C" __global__ void
sort_bfloat(layout2 vl, bf16_t* values, layout2 xl, int32_t* indices, layout2 il, const bf16_t* in, uint32_t block)
{
    sort_body<BF>(vl, values, xl, indices, il, in, block);
}
extern "C" __global__ void
sort_float(layout2 vl, float* values, layout2 xl, int32_t* indices, layout2 il, const float* in, uint32_t block)
{
    sort_body<F32>(vl, values, xl, indices, il, in, block);
}

// ---------------------------------------------------------------------------------- cumsum
// Same arithmetic as the reference: per-thread prefix in T, then the totals of the threads
// before it added one at a time, nearest first, every add rounded to T.  (The reference keeps
// 256 group totals in threadgroup memory but launches up to 1024 threads; 1024 slots here.)
template <typename T, uint32_t B>
__device__ __forceinline__ void
cumsum_body(const layout2& ol, typename T::S* out, const layout2& il, const typename T::S* in)
{
    __shared__ float gs[1024];
    const uint32_t dim = il.sizes[1], i = blockIdx.x, tid = threadIdx.x;
    const uint32_t begin = tid * B, end = begin + B;
    const uint32_t bs = end > dim ? dim % B : B;
    float run = 0.0f;
    for (uint32_t k = begin, j = 0; k < end && k < dim; k++, j++) {
        const float x = T::ld(in[at(il, i, k)]);
        run = j > 0 ? T::rt(x + run) : x;
        out[at(ol, i, k)] = T::st(run); // parked; rewritten below
    }
    (void)bs;
    gs[tid] = run;
    __syncthreads();
    for (uint32_t k = begin; k < end && k < dim; k++) {
        float v = T::ld(out[at(ol, i, k)]);
        for (uint32_t a = 1; a <= tid; a++) v = T::rt(v + gs[tid - a]);
        out[at(ol, i, k)] = T::st(v);
    }
}
#define MC_CUMSUM(B)                                                                                      \
    extern "C" __global__ void cumsum_##B##_bfloat(layout2 ol, bf16_t* out, layout2 il, const bf16_t* in) \
    {                                                                                                     \
        cumsum_body<BF, B>(ol, out, il, in);                                                              \
    }                                                                                                     \
    extern "C" __global__ void cumsum_##B##_float(layout2 ol, float* out, layout2 il, const float* in)    \
    {                                                                                                     \
        cumsum_body<F32, B>(ol, out, il, in);                                                             \
    }
MC_CUMSUM(2)
MC_CUMSUM(4)
MC_CUMSUM(8)
MC_CUMSUM(16)
MC_CUMSUM(32)
MC_CUMSUM(64)
MC_CUMSUM(128)
MC_CUMSUM(256)
MC_CUMSUM(512)
MC_CUMSUM(1024)

// ---------------------------------------------------------------------------------- multinomial
struct pcg32 {
    uint64_t state, inc;
    __device__ __forceinline__ uint32_t
    next()
    {
        const uint64_t pre = state;
        state = pre * 6364136223846793005ULL + inc;
        const uint32_t xorshifted = (uint32_t)(((pre >> 18u) ^ pre) >> 27u);
        const uint32_t rot = (uint32_t)(pre >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    __device__ __forceinline__
    pcg32(uint64_t init_state, uint64_t init_seq) : state(0), inc((init_seq << 1u) | 1u)
    {
        next();
        state += init_state;
        next();
    }
    __device__ __forceinline__ float
    uniform()
    {
        return __uint_as_float((next() >> 9) | 0x3f800000u) - 1.0f;
    }
};

// position of the draw in a DESCENDING row (kernel/multinomial.metal:60-80)
template <typename T, typename F>
__device__ __forceinline__ int32_t
descending_search(F value_at, uint32_t n, float random)
{
    int low = 0, high = (int)n;
    while (low < high) {
        const uint32_t mid = (uint32_t)(low + high) / 2;
        if (value_at(mid) > random) low = (int)mid + 1;
        else high = (int)mid;
    }
    return (low > 1 ? low : 1) - 1;
}

// The lower end of the draw interval is read at column output.size(1) - 1 of the INPUT, as the
// reference does (multinomial.metal:112); past the end of the row it is 0 (the reference reads
// out of bounds there; its own test only passes when that read returns 0).
template <typename T>
__device__ __forceinline__ void
multinomial_body(const layout2& ol, int32_t* out, const layout2& il, const typename T::S* in,
                 uint64_t init_state, uint64_t init_seq)
{
    MC_IJ;
    const uint32_t rows = ol.sizes[0], ns = ol.sizes[1], dim = il.sizes[1];
    if (i < rows && k < ns) {
        const float a = ns - 1 < dim ? T::ld(in[at(il, i, ns - 1)]) : 0.0f;
        const float b = T::ld(in[at(il, i, 0)]);
        pcg32 g(init_state + i, init_seq + k);
        const float random = T::rt(g.uniform() * (b - a) + a);
        out[at(ol, i, k)] = descending_search<T>([&](uint32_t m) { return T::ld(in[at(il, i, m)]); }, dim, random);
    }
}
extern "C" __global__ void
multinomial_bfloat(layout2 ol, int32_t* out, layout2 il, const bf16_t* in, uint64_t init_state, uint64_t init_seq)
{
    multinomial_body<BF>(ol, out, il, in, init_state, init_seq);
}
extern "C" __global__ void
multinomial_float(layout2 ol, int32_t* out, layout2 il, const float* in, uint64_t init_state, uint64_t init_seq)
{
    multinomial_body<F32>(ol, out, il, in, init_state, init_seq);
}

// ==========================================================================================
// Part B: fused default sampler.
// Keys: (orderable value bits << 32) | (0xFFFFFFFF - index): one 64-bit descending order = value
// descending, then LOWER index first (std::partial_sort leaves equal values unordered,
// nn/sampling.h:249-251; this is the documented choice of both the oracle and this kernel).
// ==========================================================================================
__device__ __forceinline__ uint64_t
make_key(float v, uint32_t index)
{
    uint32_t u = __float_as_uint(v);
    if ((u << 1) == 0) u = 0; // -0 and +0 compare equal in the reference's comparator
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return ((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - index);
}
__device__ __forceinline__ float
key_value(uint64_t key)
{
    uint32_t u = (uint32_t)(key >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint32_t
key_index(uint64_t key)
{
    return 0xFFFFFFFFu - (uint32_t)key;
}

// descending bitonic sort of n (power of two) keys in LDS by the whole workgroup
__device__ __forceinline__ void
lds_sort_desc(uint64_t* keys, uint32_t n)
{
    for (uint32_t k = 2; k <= n; k *= 2)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
                const uint32_t ij = i ^ j;
                if (i < ij) {
                    const uint64_t a = keys[i], b = keys[ij];
                    const bool up = (i & k) == 0;
                    if ((up && a < b) || (!up && a > b)) { keys[i] = b; keys[ij] = a; }
                }
            }
        }
    __syncthreads();
}

// The same network on ONE wave with the slots in registers (round 4): slot e = lane * KPL + r.  A stage whose partner distance j
// is below KPL exchanges two registers of a lane; the others exchange with lane ^ (j / KPL) through the cross-lane unit -- no LDS
// round trip and no barrier per stage.  n (a power of two, 2 .. 64 KPL) is the size of the network: lanes past n / KPL sort
// padding among themselves.  exch(x, o, first): what a slot holds after its compare-exchange with `o`, `first` = the slot is
// the one that receives what sorts FIRST in this stage's direction (lds_sort_desc / kernel/sort.metal: slot i < ij with `up`,
// or slot ij without).
template <typename E>
__device__ __forceinline__ E
wave_xor(const E& v, uint32_t lx)
{
    if constexpr (sizeof(E) == 4) {
        return __builtin_bit_cast(E, __shfl_xor(__builtin_bit_cast(int, v), (int)lx, 64));
    } else {
        static_assert(sizeof(E) == 8, "one or two dwords");
        const uint2 u = __builtin_bit_cast(uint2, v);
        return __builtin_bit_cast(E, make_uint2((uint32_t)__shfl_xor((int)u.x, (int)lx, 64), (uint32_t)__shfl_xor((int)u.y, (int)lx, 64)));
    }
}
template <int KPL, typename E, typename Exch>
__device__ __forceinline__ void
wave_bitonic(E (&x)[KPL], uint32_t lane, uint32_t n, Exch exch)
{
    for (uint32_t k = 2; k <= n; k *= 2)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j < (uint32_t)KPL) {
#pragma unroll
                for (int jj = KPL / 2; jj >= 1; jj >>= 1) {
                    if (j != (uint32_t)jj) continue;
#pragma unroll
                    for (int r = 0; r < KPL; r++) {
                        if (r & jj) continue;
                        const bool up = ((lane * KPL + r) & k) == 0;
                        const E a = x[r], b = x[r | jj];
                        x[r] = exch(a, b, up);
                        x[r | jj] = exch(b, a, !up);
                    }
                }
            } else {
                const uint32_t lx = j / KPL;
                const bool lower = (lane & lx) == 0;
                E o[KPL]; // (every exchange of the stage in flight before the first is looked at)
#pragma unroll
                for (int r = 0; r < KPL; r++) o[r] = wave_xor(x[r], lx);
#pragma unroll
                for (int r = 0; r < KPL; r++) {
                    const bool up = ((lane * KPL + r) & k) == 0;
                    x[r] = exch(x[r], o[r], lower == up);
                }
            }
        }
}
// descending, keys distinct (or zero padding): the first slot keeps the larger
struct key_exch {
    __device__ __forceinline__ uint64_t operator()(uint64_t x, uint64_t o, bool first) const { return first == (x > o) ? x : o; }
    __device__ __forceinline__ uint32_t operator()(uint32_t x, uint32_t o, bool first) const { return first ? max(x, o) : min(x, o); }
};

// launch 1: workgroup w -- ONE wave -- sorts logits [w * chunk, (w + 1) * chunk) in registers and keeps its best kpad keys, best
// first.  chunk = 64 KPL: 512 (~ 250 workgroups for a 128k-entry row), 1024 or 2048.  (Round 3: 2048-key chunks through LDS,
// 66 barrier stages of sixteen waves, on 63 workgroups.)
template <typename T, int KPL>
__device__ __forceinline__ void
topk_candidates_body(const typename T::S* logits, uint32_t n, uint32_t kpad, uint64_t* cand)
{
    const uint32_t lane = threadIdx.x, base = blockIdx.x * 64u * KPL + lane * KPL;
    typename T::S raw[KPL]; // (unconditional loads from clamped addresses: a load behind a branch is waited for on the spot)
#pragma unroll
    for (int r = 0; r < KPL; r++) raw[r] = logits[base + r < n ? base + r : n - 1];
    if constexpr (sizeof(typename T::S) == 2) {
        // bfloat logits: the upper half of make_key's value word orders them completely (its lower half repeats the sign), and a
        // position inside the chunk needs 11 bits -- a 32-bit key (value << 16 | 0xFFFF - position) sorts exactly as the 64-bit
        // one does: half the cross-lane traffic, v_max / v_min instead of 64-bit compares.  Zero stays the padding (no real key
        // is zero: its lower half is at least 0xF800).
        uint32_t key[KPL];
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const uint32_t hi = (uint32_t)(make_key(T::ld(raw[r]), 0u) >> 48);
            key[r] = base + r < n ? (hi << 16) | (0xFFFFu - (lane * KPL + r)) : 0u;
        }
        wave_bitonic<KPL>(key, lane, 64u * KPL, key_exch{});
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            if (lane * KPL + r >= kpad) continue;
            // back to the row's 64-bit key: the bfloat behind the ordered upper half, the index behind the position
            const uint32_t hi = key[r] >> 16, bits = hi ^ ((hi & 0x8000u) ? 0x8000u : 0xFFFFu);
            const uint64_t k64 = make_key(__uint_as_float(bits << 16), blockIdx.x * 64u * KPL + (0xFFFFu - (key[r] & 0xFFFFu)));
            cand[(size_t)blockIdx.x * kpad + lane * KPL + r] = key[r] ? k64 : 0ull;
        }
    } else {
        uint64_t key[KPL];
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const uint64_t kk = make_key(T::ld(raw[r]), base + r);
            key[r] = base + r < n ? kk : 0ull; // 0 sorts last
        }
        wave_bitonic<KPL>(key, lane, 64u * KPL, key_exch{});
#pragma unroll
        for (int r = 0; r < KPL; r++)
            if (lane * KPL + r < kpad) cand[(size_t)blockIdx.x * kpad + lane * KPL + r] = key[r];
    }
}
#define MC_TOPK_CANDIDATES(NAME, T)                                                                          \
    extern "C" __global__ void __launch_bounds__(64)                                                         \
    NAME(const typename T::S* logits, uint32_t n, uint32_t kpad, uint64_t* cand, uint32_t chunk)             \
    {                                                                                                        \
        if (chunk == 512) topk_candidates_body<T, 8>(logits, n, kpad, cand);                                 \
        else if (chunk == 1024) topk_candidates_body<T, 16>(logits, n, kpad, cand);                          \
        else if (chunk == 2048) topk_candidates_body<T, 32>(logits, n, kpad, cand);                          \
    }
MC_TOPK_CANDIDATES(mc_topk_candidates_bfloat, BF)
MC_TOPK_CANDIDATES(mc_topk_candidates_float, F32)

struct sampler_params {
    uint32_t k;          // top-k (<= 128)
    uint32_t ncand;      // candidate keys written by launch 1 = nlists * kpad
    uint32_t cap;        // keys the dynamic LDS holds (a power of two >= 2 * kpad)
    float inv_temp;      // T(1 / T(temperature))
    float top_p;         // T(p)
    uint32_t nlists;     // sorted candidate lists (workgroups of launch 1), <= MC_SAMPLE_LISTS_MAX
    uint32_t kpad;       // keys per list
};
constexpr uint32_t MC_SAMPLE_LISTS_MAX = 1024;

struct step_state_s { // prefix of step_state (decode_kernels.hip)
    int32_t token, pos, kv_len, write_slot, ring_base, step_index, rope_row, rolled;
};

// launch 2: one workgroup.  Dynamic LDS: ncand_pad keys.  taps (optional): 7*k floats, the
// intermediates in the order of mco_sample_default.
template <typename T>
__device__ __forceinline__ void
sample_body(const uint64_t* cand, sampler_params p, const uint64_t* seeds, uint32_t n_seed_pairs,
            step_state_s* st, int32_t* tokens_out, float* taps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    __shared__ float val[128], srt[128], cum[128], gsum[64], part[32], grp[32];
    __shared__ int32_t ids[128], sidx[128];
    __shared__ float exp_sum;
    const uint32_t tid = threadIdx.x, k = p.k;
    // the draw's seed pair: requested NOW (two dependent round trips that thread 0 used to start at the very end); unconditional
    // loads from valid addresses, masked
    const uint32_t seed_pair = n_seed_pairs ? (uint32_t)st->step_index % n_seed_pairs : 0u;
    const uint64_t* seed_p = n_seed_pairs ? seeds + 2 * seed_pair : cand;
    const uint64_t seed0 = seed_p[0], seed1 = seed_p[1];
    // ---- the k best of the candidate lists WITHOUT sorting them all (round 4; it was one bitonic network over all <= 4096
    // candidates: 78 barrier stages of 1024 threads).  Every list is sorted, so the k-th largest list HEAD is a lower bound L
    // of the k-th largest key (k keys are >= it): only keys >= L can be among the k best -- typically k .. 4 k of them.  Keys
    // are distinct (the index is part of the key), so the k best of that subset are the k best of the row: the result is the
    // full sort's, key for key.
    __shared__ uint64_t heads[MC_SAMPLE_LISTS_MAX];
    __shared__ uint32_t cnt;
    uint32_t nl_pad = 2;
    while (nl_pad < p.nlists) nl_pad *= 2;
    if (tid == 0) cnt = 0;
    if (tid < 64) {
        // the list heads, sorted by wave 0 in registers
        auto sort_heads = [&](auto kpl_c) {
            constexpr int KPL = decltype(kpl_c)::value;
            uint64_t h[KPL];
#pragma unroll
            for (int r = 0; r < KPL; r++) {
                const uint32_t e = tid * KPL + r;
                h[r] = cand[(size_t)(e < p.nlists ? e : p.nlists - 1) * p.kpad];
            }
#pragma unroll
            for (int r = 0; r < KPL; r++) h[r] = tid * KPL + r < p.nlists ? h[r] : 0ull;
            wave_bitonic<KPL>(h, tid, nl_pad, key_exch{});
#pragma unroll
            for (int r = 0; r < KPL; r++)
                if (tid * KPL + r < nl_pad) heads[tid * KPL + r] = h[r];
        };
        if (nl_pad <= 64) sort_heads(std::integral_constant<int, 1>{});
        else if (nl_pad <= 256) sort_heads(std::integral_constant<int, 4>{});
        else sort_heads(std::integral_constant<int, 16>{});
    }
    __syncthreads();
    // (fewer lists than k: every candidate may be needed)
    uint64_t L = p.nlists >= k ? heads[k - 1] : 0ull;
    // keys >= t over all lists (each list is descending: stop at the first smaller one); with `collect` they go to keys[] while
    // there is room.  The first four keys of a list come in ONE round of loads (a list rarely contributes more).
    auto scan = [&](uint64_t t, bool collect) {
        uint32_t c = 0;
        for (uint32_t w = tid; w < p.nlists; w += blockDim.x) {
            const uint64_t* lp = cand + (size_t)w * p.kpad;
            uint64_t q[4];
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) q[i] = lp[i < p.kpad ? i : p.kpad - 1];
            uint32_t i = 0;
            for (; i < p.kpad; i++) {
                const uint64_t key = i < 4 ? (i == 0 ? q[0] : (i == 1 ? q[1] : (i == 2 ? q[2] : q[3]))) : lp[i];
                if (key < t) break;
                c++;
                if (collect) {
                    const uint32_t at = atomicAdd(&cnt, 1u);
                    if (at < p.cap) keys[at] = key;
                }
            }
        }
        return c;
    };
    scan(L, true);
    __syncthreads();
    if (cnt > p.cap) {
        // more keys >= L than the LDS holds (long runs of equal logits with a wide top-k): the exact k-th largest key by
        // bisection over the 64 key bits -- count(keys >= t) is monotone in t, keys are distinct, so the largest t with a
        // count >= k IS the k-th largest key and exactly k keys are >= it.  64 counting rounds: the slow path, never the usual one.
        uint64_t lo = 0;
        for (int bit = 63; bit >= 0; bit--) {
            const uint64_t t = lo | (1ull << bit);
            __syncthreads();
            if (tid == 0) cnt = 0;
            __syncthreads();
            const uint32_t c = scan(t, false);
            if (c) atomicAdd(&cnt, c);
            __syncthreads();
            if (cnt >= k) lo = t;
        }
        __syncthreads();
        if (tid == 0) cnt = 0;
        __syncthreads();
        scan(lo, true);
    }
    __syncthreads();
    const uint32_t have = cnt < p.cap ? cnt : p.cap;
    uint32_t npad = 2;
    while (npad < have || npad < k) npad *= 2;
    for (uint32_t i = have + tid; i < npad; i += blockDim.x) keys[i] = 0ull;
    __syncthreads();
    if (npad <= 512) {
        // the usual case: one wave, eight keys per lane
        if (tid < 64) {
            uint64_t x[8];
#pragma unroll
            for (int r = 0; r < 8; r++) x[r] = tid * 8 + r < npad ? keys[tid * 8 + r] : 0ull;
            wave_bitonic<8>(x, tid, npad, key_exch{});
#pragma unroll
            for (int r = 0; r < 8; r++)
                if (tid * 8 + r < npad) keys[tid * 8 + r] = x[r];
        }
        __syncthreads();
    } else {
        lds_sort_desc(keys, npad);
    }
    // topk_sampler: values + vocabulary ids of the k best, best first
    if (tid < k) {
        const float scaled = T::rt(key_value(keys[tid]) * p.inv_temp);   // mul(logits, T(1)/temperature)
        ids[tid] = (int32_t)key_index(keys[tid]);
        val[tid] = scaled;
        if (taps) taps[0 * k + tid] = scaled;
    }
    __syncthreads();
    // softmax without max shift (kernel/softmax.metal:24-88): one element per thread, then the
    // two-level 32-lane reduction -- summed here in the order the oracle fixes for it
    if (tid < 128) srt[tid] = tid < k ? exp_precise(val[tid]) : 0.0f; // srt = exp(x) scratch
    __syncthreads();
    // (lanes[i] += lanes[i + off] for off = 16 .. 1 over 32 lanes IS the shuffle-down tree: lane 0 of a group of 32 ends with
    //  exactly that sum -- the two levels without their serial loops over LDS)
    if (tid < 128) {
        float v = srt[tid];
        for (int off = 16; off >= 1; off >>= 1) v = v + __shfl_down(v, off, 32);
        if ((tid & 31u) == 0) grp[tid >> 5] = v;
    }
    __syncthreads();
    if (tid < 32) {
        float g = tid < 4 ? grp[tid] : 0.0f;
        for (int off = 16; off >= 1; off >>= 1) g = g + __shfl_down(g, off, 32);
        if (tid == 0) exp_sum = 1.0f / g;
    }
    __syncthreads();
    uint32_t aligned = 1;
    while (aligned < k) aligned *= 2;
    const float pr = tid < k ? T::rt(srt[tid < 128 ? tid : 0] * exp_sum) : -INFINITY;
    if (taps && tid < k) taps[1 * k + tid] = pr;
    __syncthreads();
    if (tid < 128) {
        srt[tid] = pr;
        sidx[tid] = (int32_t)tid;
    }
    // sort (kernel/sort.metal): the reference's bitonic network over `aligned` slots, -inf padded -- the same compare-exchanges
    // in the same order, by wave 0 in registers (two slots per lane; the probability carries its position as payload)
    __syncthreads();
    if (tid < 64 && aligned >= 2) {
        struct pv { float v; int32_t i; };
        pv x[2];
#pragma unroll
        for (int r = 0; r < 2; r++) x[r] = pv{srt[2 * tid + r], sidx[2 * tid + r]};
        // (kernel/sort.metal's strict compares: equal values stay where they are)
        wave_bitonic<2>(x, tid, aligned, [](const pv& a, const pv& o, bool first) { return (first ? a.v < o.v : a.v > o.v) ? o : a; });
#pragma unroll
        for (int r = 0; r < 2; r++) {
            srt[2 * tid + r] = x[r].v;
            sidx[2 * tid + r] = x[r].i;
        }
    }
    __syncthreads();
    // cumsum (kernel/cumsum.metal, BlockSize 2 for k <= 2048): thread t owns elements 2t, 2t+1
    const uint32_t nth = (k + 1) / 2;
    float l0 = 0.0f, l1 = 0.0f;
    if (tid < nth) {
        l0 = srt[2 * tid];
        const bool two = 2 * tid + 1 < k;
        l1 = two ? T::rt(srt[2 * tid + 1] + l0) : 0.0f;
        gsum[tid] = two ? l1 : l0;
    }
    __syncthreads();
    if (tid < nth) {
        for (uint32_t a = 1; a <= tid; a++) {
            const float acc = gsum[tid - a];
            l0 = T::rt(l0 + acc);
            l1 = T::rt(l1 + acc);
        }
        cum[2 * tid] = l0;
        if (2 * tid + 1 < k) cum[2 * tid + 1] = l1;
    }
    __syncthreads();
    // sub, gt(p), scatter(0), gather(indices)
    if (tid < k) {
        const float diff = T::rt(cum[tid] - srt[tid]);
        const float masked = diff > p.top_p ? 0.0f : srt[tid];
        if (taps) {
            taps[2 * k + tid] = srt[tid];
            taps[3 * k + tid] = cum[tid];
            taps[4 * k + tid] = diff;
            taps[5 * k + tid] = masked;
            taps[6 * k + tid] = (float)ids[sidx[tid]];
        }
        val[tid] = masked;
    }
    __syncthreads();
    // multinomial(sample_size 1) + gather: see multinomial_body for the interval quirk
    if (tid == 0) {
        const uint64_t s0 = n_seed_pairs ? seed0 : 0ull, s1 = n_seed_pairs ? seed1 : 0ull;
        const float a = val[0], b = val[0]; // column output.size(1) - 1 == 0
        pcg32 g(s0, s1);
        const float random = T::rt(g.uniform() * (b - a) + a);
        const int32_t pos = descending_search<T>([&](uint32_t m) { return val[m]; }, k, random);
        const int32_t token = ids[sidx[pos]];
        st->token = token;
        if (tokens_out) tokens_out[st->step_index] = token;
    }
    (void)part;
}
extern "C" __global__ void
mc_sample_bfloat(const uint64_t* cand, sampler_params p, const uint64_t* seeds, uint32_t n_seed_pairs,
                 step_state_s* st, int32_t* tokens_out, float* taps)
{
    sample_body<BF>(cand, p, seeds, n_seed_pairs, st, tokens_out, taps);
}
extern "C" __global__ void
mc_sample_float(const uint64_t* cand, sampler_params p, const uint64_t* seeds, uint32_t n_seed_pairs,
                step_state_s* st, int32_t* tokens_out, float* taps)
{
    sample_body<F32>(cand, p, seeds, n_seed_pairs, st, tokens_out, taps);
}
