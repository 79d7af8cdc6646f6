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

constexpr uint32_t MC_TOPK_CHUNK = 2048;

// launch 1: workgroup w sorts logits [w*2048, (w+1)*2048) and keeps its best kpad keys
template <typename T>
__device__ __forceinline__ void
topk_candidates_body(const typename T::S* logits, uint32_t n, uint32_t kpad, uint64_t* cand)
{
    __shared__ uint64_t keys[MC_TOPK_CHUNK];
    const uint32_t base = blockIdx.x * MC_TOPK_CHUNK;
    for (uint32_t i = threadIdx.x; i < MC_TOPK_CHUNK; i += blockDim.x)
        keys[i] = base + i < n ? make_key(T::ld(logits[base + i]), base + i) : 0ull; // 0 sorts last
    lds_sort_desc(keys, MC_TOPK_CHUNK);
    for (uint32_t i = threadIdx.x; i < kpad; i += blockDim.x) cand[(size_t)blockIdx.x * kpad + i] = keys[i];
}
extern "C" __global__ void
mc_topk_candidates_bfloat(const bf16_t* logits, uint32_t n, uint32_t kpad, uint64_t* cand)
{
    topk_candidates_body<BF>(logits, n, kpad, cand);
}
extern "C" __global__ void
mc_topk_candidates_float(const float* logits, uint32_t n, uint32_t kpad, uint64_t* cand)
{
    topk_candidates_body<F32>(logits, n, kpad, cand);
}

struct sampler_params {
    uint32_t k;          // top-k (<= 128)
    uint32_t ncand;      // candidate keys written by launch 1
    uint32_t ncand_pad;  // next power of two
    float inv_temp;      // T(1 / T(temperature))
    float top_p;         // T(p)
};

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
    for (uint32_t i = tid; i < p.ncand_pad; i += blockDim.x) keys[i] = i < p.ncand ? cand[i] : 0ull;
    lds_sort_desc(keys, p.ncand_pad);
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
    if (tid < 4) {
        float lanes[32];
        for (int l = 0; l < 32; l++) lanes[l] = srt[tid * 32 + l];
        for (int off = 16; off >= 1; off >>= 1)
            for (int i = 0; i < off; i++) lanes[i] = lanes[i] + lanes[i + off];
        grp[tid] = lanes[0];
    } else if (tid < 32) {
        grp[tid] = 0.0f;
    }
    __syncthreads();
    if (tid == 0) {
        float g[32];
        for (int l = 0; l < 32; l++) g[l] = grp[l];
        for (int off = 16; off >= 1; off >>= 1)
            for (int i = 0; i < off; i++) g[i] = g[i] + g[i + off];
        exp_sum = 1.0f / g[0];
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
    // sort (kernel/sort.metal): the reference's bitonic network over `aligned` slots, -inf padded
    for (uint32_t kk = 2; kk <= aligned; kk *= 2)
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            __syncthreads();
            const uint32_t i = tid, ij = i ^ j;
            if (i < aligned && i < ij) {
                const float vi = srt[i], vj = srt[ij];
                const bool up = (i & kk) == 0;
                if ((up && vi < vj) || (!up && vi > vj)) {
                    srt[i] = vj; srt[ij] = vi;
                    const int32_t t = sidx[i]; sidx[i] = sidx[ij]; sidx[ij] = t;
                }
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
        const uint32_t pair = n_seed_pairs ? (uint32_t)st->step_index % n_seed_pairs : 0u;
        const uint64_t s0 = n_seed_pairs ? seeds[2 * pair] : 0ull, s1 = n_seed_pairs ? seeds[2 * pair + 1] : 0ull;
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
