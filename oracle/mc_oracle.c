/*
 * mc_oracle.c -- CPU ORACLE (test infrastructure, see mc_oracle.h).  Scalar restatement of the
 * reference's kernels; every function cites the reference lines it follows.
 *
 * Build: see oracle/Makefile (gcc -O2 -fno-fast-math -ffp-contract=off -fopenmp).
 * -ffp-contract=off keeps a*b+c as two roundings so results do not depend on the host ISA.
 */
#include "mc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define AINLINE static inline __attribute__((always_inline))

/* ------------------------------------------------------------------------------------------
 * bf16  (include/metalchat/dtype.h:32-58: round-to-nearest-even on the float bits; NaN keeps a
 * quiet bit; the host class flushes subnormals to zero, the device `bfloat(float)` conversion is
 * plain RNE -- the two agree on every value the path produces, subnormal inputs are kept RNE here
 * because that is what kernel code does).
 * ------------------------------------------------------------------------------------------ */
mco_bf16
mco_f32_to_bf16(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u && (u & 0x007fffffu)) {
        return (mco_bf16)((u >> 16) | 0x40u); /* NaN stays NaN */
    }
    u += 0x7fffu + ((u >> 16) & 1u);
    return (mco_bf16)(u >> 16);
}

float
mco_bf16_to_f32(mco_bf16 b)
{
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

AINLINE float
ld(const int dt, const void* p, size_t i)
{
    if (dt == MCO_BF16) return mco_bf16_to_f32(((const mco_bf16*)p)[i]);
    return ((const float*)p)[i];
}

AINLINE void
st(const int dt, void* p, size_t i, float v)
{
    if (dt == MCO_BF16)
        ((mco_bf16*)p)[i] = mco_f32_to_bf16(v);
    else
        ((float*)p)[i] = v;
}

/* round a float to T and back: the value a T-typed temporary would hold */
AINLINE float
rt(const int dt, float v)
{
    return dt == MCO_BF16 ? mco_bf16_to_f32(mco_f32_to_bf16(v)) : v;
}

/* kernel/tensor.h:84-88,115-122,146-156: at(i...) = data[sum(stride_d*i_d + offset_d)] */
AINLINE size_t
at1(const uint32_t* l, uint32_t i0)
{
    return (size_t)l[1] * i0 + l[2];
}
AINLINE size_t
at2(const uint32_t* l, uint32_t i0, uint32_t i1)
{
    return ((size_t)l[2] * i0 + l[4]) + ((size_t)l[3] * i1 + l[5]);
}
AINLINE size_t
at3(const uint32_t* l, uint32_t i0, uint32_t i1, uint32_t i2)
{
    return ((size_t)l[3] * i0 + l[6]) + ((size_t)l[4] * i1 + l[7]) + ((size_t)l[5] * i2 + l[8]);
}

/* correctly rounded float transcendental = the value `precise::f` is an approximation of */
AINLINE float
exp_precise(float x)
{
    return (float)exp((double)x);
}

/* ------------------------------------------------------------------------------------------
 * bmm  (kernel/bmm.metal:25-82).  C[b,m,n] = T(sum_k float(A[b,m,k]) * float(B[b,k,n])), fp32
 * partial accumulated in increasing k (the 8x8 tiling only stages operands, the per-output
 * accumulation order is k = 0..K-1).
 * ------------------------------------------------------------------------------------------ */
AINLINE void
bmm_impl(const int dt, const uint32_t* ol, void* out, const uint32_t* al, const void* a,
         const uint32_t* bl, const void* b)
{
    const uint32_t nb = al[0], M = al[1], K = al[2], N = bl[2];
#pragma omp parallel for collapse(2) schedule(static)
    for (uint32_t bi = 0; bi < nb; bi++) {
        for (uint32_t n = 0; n < N; n++) {
            for (uint32_t m = 0; m < M; m++) {
                float partial = 0.0f;
                for (uint32_t k = 0; k < K; k++) {
                    partial += ld(dt, a, at3(al, bi, m, k)) * ld(dt, b, at3(bl, bi, k, n));
                }
                st(dt, out, at3(ol, bi, m, n), partial);
            }
        }
    }
}

void
mco_bmm(int dt, const uint32_t* ol, void* out, const uint32_t* al, const void* a,
        const uint32_t* bl, const void* b)
{
    if (dt == MCO_BF16)
        bmm_impl(MCO_BF16, ol, out, al, a, bl, b);
    else
        bmm_impl(MCO_F32, ol, out, al, a, bl, b);
}

/* ------------------------------------------------------------------------------------------
 * hadamard (kernel/mul.metal:13-48): out = in1 * in2 evaluated in T.
 * ------------------------------------------------------------------------------------------ */
void
mco_hadamard(int dt, const uint32_t* ol, void* out, const uint32_t* al, const void* a,
             const uint32_t* bl, const void* b)
{
    for (uint32_t i = 0; i < al[0]; i++)
        for (uint32_t k = 0; k < al[1]; k++)
            st(dt, out, at2(ol, i, k), ld(dt, a, at2(al, i, k)) * ld(dt, b, at2(bl, i, k)));
}

/* ------------------------------------------------------------------------------------------
 * hadamard_broadcast (kernel/mul.metal:51-85) -- THE DEQUANTIZER:
 *   out[i,j] = Output(in1[i,j]) * Output(in2[i % in2.size(0)])      (product evaluated in Output)
 * ------------------------------------------------------------------------------------------ */
void
mco_hadamard_broadcast(int odt, int sdt, const uint32_t* ol, void* out, const uint32_t* il,
                       const int8_t* in1, const uint32_t* sl, const void* in2)
{
    for (uint32_t i = 0; i < il[0]; i++) {
        const float s = rt(odt, ld(sdt, in2, at1(sl, i % sl[0])));
        for (uint32_t j = 0; j < il[1]; j++) {
            const float q = rt(odt, (float)in1[at2(il, i, j)]);
            st(odt, out, at2(ol, i, j), q * s);
        }
    }
}

/* scalar_mul (kernel/mul.metal:88-121): out = in * multiplier in T */
void
mco_scalar_mul(int dt, const uint32_t* ol, void* out, const uint32_t* il, const void* in,
               const void* multiplier)
{
    const float m = ld(dt, multiplier, 0);
    for (uint32_t i = 0; i < il[0]; i++)
        for (uint32_t k = 0; k < il[1]; k++)
            st(dt, out, at2(ol, i, k), ld(dt, in, at2(il, i, k)) * m);
}

/* simd_sum over a 32-lane group, restated as a butterfly (the hardware order is unspecified) */
static float
simd_sum32(float* v)
{
    for (int off = 16; off >= 1; off >>= 1)
        for (int i = 0; i < off; i++) v[i] = v[i] + v[i + off];
    return v[0];
}

/* two-level threadgroup reduction shared by rmsnorm and softmax
 * (kernel/rmsnorm.metal:58-84, kernel/softmax.metal:50-75): per-thread partials -> simd_sum ->
 * 32-slot threadgroup array -> simd_sum. */
static float
threadgroup_sum(const float* partial, uint32_t nthreads)
{
    float groups[32];
    for (int g = 0; g < 32; g++) groups[g] = 0.0f;
    for (uint32_t g = 0; g * 32 < nthreads; g++) {
        float lanes[32];
        for (uint32_t l = 0; l < 32; l++) {
            uint32_t t = g * 32 + l;
            lanes[l] = t < nthreads ? partial[t] : 0.0f;
        }
        groups[g] = simd_sum32(lanes);
    }
    return simd_sum32(groups);
}

static uint32_t
ceil_div_u32(uint32_t a, uint32_t b)
{
    return (a + b - 1) / b;
}

/* ------------------------------------------------------------------------------------------
 * rmsnorm (kernel/rmsnorm.metal:28-98; launch maths include/metalchat/kernel/rmsnorm.h:40-45)
 *   y = T((mu + w) * x * rsqrt(mean(x^2) + eps)), fp32 math.
 * ------------------------------------------------------------------------------------------ */
AINLINE void
rmsnorm_impl(const int dt, const uint32_t* ol, void* out, const uint32_t* il, const void* in,
             const uint32_t* wl, const void* w, float eps, float mu, uint32_t max_threads)
{
    const uint32_t rows = il[0], dim = il[1];
    const uint32_t block = ceil_div_u32(dim, max_threads);
    const uint32_t nthreads = ceil_div_u32(dim, block);
    float* partial = (float*)malloc(sizeof(float) * nthreads);
    for (uint32_t i = 0; i < rows; i++) {
        for (uint32_t t = 0; t < nthreads; t++) {
            float s = 0.0f;
            for (uint32_t j = t * block; j < (t + 1) * block && j < dim; j++) {
                float x = ld(dt, in, at2(il, i, j));
                s += x * x;
            }
            partial[t] = s;
        }
        const float acc = threadgroup_sum(partial, nthreads);
        const float mean_sq = acc / (float)dim;
        const float inv = 1.0f / sqrtf(mean_sq + eps);
        for (uint32_t j = 0; j < dim; j++) {
            const float x = ld(dt, in, at2(il, i, j));
            const float weight = mu + ld(dt, w, at1(wl, j));
            st(dt, out, at2(ol, i, j), weight * x * inv);
        }
    }
    free(partial);
}

void
mco_rmsnorm(int dt, const uint32_t* ol, void* out, const uint32_t* il, const void* in,
            const uint32_t* wl, const void* w, float eps, float mu, uint32_t max_threads)
{
    if (dt == MCO_BF16)
        rmsnorm_impl(MCO_BF16, ol, out, il, in, wl, w, eps, mu, max_threads);
    else
        rmsnorm_impl(MCO_F32, ol, out, il, in, wl, w, eps, mu, max_threads);
}

/* ------------------------------------------------------------------------------------------
 * rope (kernel/rope.metal:29-63).  NOTE `head_dim` in the kernel is f_cos.size(1) = dim/2.
 * PARITY UNPINNED: the reference has no test for the rotation kernel.
 * ------------------------------------------------------------------------------------------ */
void
mco_rope(int dt, const uint32_t* ol, void* out, const uint32_t* il, const void* in,
         const uint32_t* cl, const float* fcos, const uint32_t* sl, const float* fsin,
         uint32_t batch_size, uint32_t n_head, uint32_t start_pos)
{
    const uint32_t rows = il[0], half = cl[1];
    for (uint32_t i = 0; i < rows; i++) {
        const uint32_t pos = i / (batch_size * n_head);
        for (uint32_t k = 0; k < half; k++) {
            const float x1 = ld(dt, in, at2(il, i, k));
            const float x2 = ld(dt, in, at2(il, i, half + k));
            const float c = fcos[at2(cl, start_pos + pos, k)];
            const float s = fsin[at2(sl, start_pos + pos, k)];
            st(dt, out, at2(ol, i, k), c * x1 - s * x2);
            st(dt, out, at2(ol, i, half + k), s * x1 + c * x2);
        }
    }
}

/* one (cos, sin) pair exactly as rope_freqs computes it (kernel/rope.metal:94-99), with the
 * `precise::` functions restated as correctly rounded float results */
static void
rope_angle(uint32_t pos, uint32_t j, uint32_t dim, float theta, float* c, float* s)
{
    const float e = 2.0f * (float)j / (float)dim;
    const float freq = 1.0f / (float)pow((double)theta, (double)e);
    const float angle = (float)pos * freq;
    *c = (float)cos((double)angle);
    *s = (float)sin((double)angle);
}

/* rope_freqs (kernel/rope.metal:77-102) */
void
mco_rope_freqs(const uint32_t* cl, float* fcos, const uint32_t* sl, float* fsin, uint32_t dim,
               uint32_t start_pos, float theta)
{
    for (uint32_t i = 0; i < cl[0]; i++)
        for (uint32_t j = 0; j < dim / 2; j++)
            rope_angle(start_pos + i, j, dim, theta, &fcos[at2(cl, i, j)], &fsin[at2(sl, i, j)]);
}

/* ------------------------------------------------------------------------------------------
 * softmax (kernel/softmax.metal:24-88): exp(x) / sum(exp(x)) with NO max subtraction.
 * ------------------------------------------------------------------------------------------ */
AINLINE void
softmax_impl(const int dt, const uint32_t* ol, void* out, const uint32_t* il, const void* in,
             uint32_t max_threads)
{
    const uint32_t rows = il[0], dim = il[1];
    const uint32_t block = ceil_div_u32(dim, max_threads);
    const uint32_t nthreads = ceil_div_u32(dim, block);
    float* partial = (float*)malloc(sizeof(float) * nthreads);
    for (uint32_t i = 0; i < rows; i++) {
        for (uint32_t t = 0; t < nthreads; t++) {
            float s = 0.0f;
            for (uint32_t j = t * block; j < (t + 1) * block && j < dim; j++)
                s += exp_precise(ld(dt, in, at2(il, i, j)));
            partial[t] = s;
        }
        const float exp_sum = 1.0f / threadgroup_sum(partial, nthreads);
        for (uint32_t j = 0; j < dim; j++)
            st(dt, out, at2(ol, i, j), exp_precise(ld(dt, in, at2(il, i, j))) * exp_sum);
    }
    free(partial);
}

void
mco_softmax(int dt, const uint32_t* ol, void* out, const uint32_t* il, const void* in,
            uint32_t max_threads)
{
    if (dt == MCO_BF16)
        softmax_impl(MCO_BF16, ol, out, il, in, max_threads);
    else
        softmax_impl(MCO_F32, ol, out, il, in, max_threads);
}

/* embedding (kernel/embedding.metal:38-70): out[i,j,k] = w[in[i,j], k] */
void
mco_embedding(int dt, const uint32_t* ol, void* out, const uint32_t* il, const int32_t* in,
              const uint32_t* wl, const void* w)
{
    const size_t esz = dt == MCO_BF16 ? 2 : 4;
    for (uint32_t i = 0; i < il[0]; i++)
        for (uint32_t j = 0; j < il[1]; j++) {
            const uint32_t id = (uint32_t)in[at2(il, i, j)];
            for (uint32_t k = 0; k < wl[1]; k++)
                memcpy((char*)out + esz * at3(ol, i, j, k),
                       (const char*)w + esz * at2(wl, id, k), esz);
        }
}

/* copy (kernel/copy.metal:20-42) */
void
mco_copy(int dt, const uint32_t* ol, void* out, const uint32_t* il, const void* in)
{
    const size_t esz = dt == MCO_BF16 ? 2 : 4;
    for (uint32_t i = 0; i < il[0]; i++)
        for (uint32_t k = 0; k < il[1]; k++)
            memcpy((char*)out + esz * at2(ol, i, k), (const char*)in + esz * at2(il, i, k), esz);
}

/* roll (kernel/roll.metal:23-49): out[k] = in[base + ((k/stride + shift) % size)*stride + k%stride] */
void
mco_roll(int dt, const uint32_t* ol, void* out, const uint32_t* il, const void* in,
         uint32_t shift, uint32_t size, uint32_t stride)
{
    const size_t esz = dt == MCO_BF16 ? 2 : 4;
    const uint32_t stride_size = size * stride;
    for (uint32_t k = 0; k < il[0]; k++) {
        const uint32_t base = (k / stride_size) * stride_size;
        const uint32_t i = (k / stride + shift) % size;
        const uint32_t j = k % stride;
        const uint32_t m = base + i * stride + j;
        memcpy((char*)out + esz * at1(ol, k), (const char*)in + esz * at1(il, m), esz);
    }
}

/* add (kernel/arithmetic.metal:13-46) evaluated in T */
void
mco_add(int dt, const uint32_t* ol, void* out, const uint32_t* al, const void* a,
        const uint32_t* bl, const void* b)
{
    for (uint32_t i = 0; i < al[0]; i++)
        for (uint32_t k = 0; k < al[1]; k++)
            st(dt, out, at2(ol, i, k), ld(dt, a, at2(al, i, k)) + ld(dt, b, at2(bl, i, k)));
}

/* add_broadcast (kernel/arithmetic.metal:49-85): out[i,j] = in1[i,j] + in2[j % n] */
void
mco_add_broadcast(int dt, const uint32_t* ol, void* out, const uint32_t* al, const void* a,
                  const uint32_t* bl, const void* b)
{
    for (uint32_t i = 0; i < al[0]; i++)
        for (uint32_t j = 0; j < al[1]; j++)
            st(dt, out, at2(ol, i, j), ld(dt, a, at2(al, i, j)) + ld(dt, b, at1(bl, j % bl[0])));
}

/* silu (kernel/activation.metal:13-41): x / (T(1) + T(exp(-x))), every step a T value */
AINLINE float
silu_T(const int dt, float x)
{
    const float e = rt(dt, exp_precise(-x));
    const float d = rt(dt, 1.0f + e);
    return rt(dt, x / d);
}

void
mco_silu(int dt, const uint32_t* ol, void* out, const uint32_t* il, const void* in)
{
    for (uint32_t i = 0; i < il[0]; i++)
        for (uint32_t k = 0; k < il[1]; k++)
            st(dt, out, at2(ol, i, k), silu_T(dt, ld(dt, in, at2(il, i, k))));
}

/* gelu (kernel/activation.metal:44-78): tanh approximation in fp32 */
AINLINE float
gelu_f(float x)
{
    const float beta = 1.41421356237309504880f * 1.12837916709551257390f * 0.5f;
    const float kappa = 0.044715f;
    const float x3 = x * x * x;
    const float inner = beta * (x + kappa * x3);
    return 0.5f * x * (1.0f + (float)tanh((double)inner));
}

void
mco_gelu(int dt, const uint32_t* ol, void* out, const uint32_t* il, const void* in)
{
    for (uint32_t i = 0; i < il[0]; i++)
        for (uint32_t k = 0; k < il[1]; k++)
            st(dt, out, at2(ol, i, k), gelu_f(ld(dt, in, at2(il, i, k))));
}

/* ==========================================================================================
 * Sampler chain: the kernels nn::nucleus_sampler / topk_sampler /
 * multinomial_sampler are made of (include/metalchat/nn/sampling.h:152-315).
 * ========================================================================================== */

/* sub (kernel/arithmetic.metal:88-121) evaluated in T */
void
mco_sub(int dt, const uint32_t* ol, void* out, const uint32_t* al, const void* a,
        const uint32_t* bl, const void* b)
{
    for (uint32_t i = 0; i < al[0]; i++)
        for (uint32_t k = 0; k < al[1]; k++)
            st(dt, out, at2(ol, i, k), ld(dt, a, at2(al, i, k)) - ld(dt, b, at2(bl, i, k)));
}

/* div (kernel/arithmetic.metal:124-157) evaluated in T */
void
mco_div(int dt, const uint32_t* ol, void* out, const uint32_t* al, const void* a,
        const uint32_t* bl, const void* b)
{
    for (uint32_t i = 0; i < al[0]; i++)
        for (uint32_t k = 0; k < al[1]; k++)
            st(dt, out, at2(ol, i, k), ld(dt, a, at2(al, i, k)) / ld(dt, b, at2(bl, i, k)));
}

/* sum (kernel/sum.metal:22-74, launch include/metalchat/kernel/sum.h:88-117): per-thread slices of
 * block = ceil(dim / max_threads) elements, then the two-level 32-lane reduction, fp32, stored as T.
 * Pinned by test/test_kernel_sum.cc:43-64 (within 0.01 of the sequential sum). */
void
mco_sum(int dt, const uint32_t* ol, void* out, const uint32_t* il, const void* in, uint32_t max_threads)
{
    const uint32_t rows = il[0], dim = il[1];
    const uint32_t block = ceil_div_u32(dim, max_threads);
    const uint32_t nthreads = ceil_div_u32(dim, block);
    float* partial = (float*)malloc(sizeof(float) * nthreads);
    for (uint32_t i = 0; i < rows; i++) {
        for (uint32_t t = 0; t < nthreads; t++) {
            float s = 0.0f;
            for (uint32_t j = t * block; j < (t + 1) * block && j < dim; j++) s += ld(dt, in, at2(il, i, j));
            partial[t] = s;
        }
        st(dt, out, at1(ol, i), threadgroup_sum(partial, nthreads));
    }
    free(partial);
}

/* gt / le (kernel/logical.metal:13-68): bool = one byte, compare in T */
void
mco_gt(int dt, const uint32_t* ol, uint8_t* out, const uint32_t* il, const void* in, float value)
{
    const float v = rt(dt, value);
    for (uint32_t i = 0; i < il[0]; i++)
        for (uint32_t k = 0; k < il[1]; k++) out[at2(ol, i, k)] = ld(dt, in, at2(il, i, k)) > v;
}

void
mco_le(int dt, const uint32_t* ol, uint8_t* out, const uint32_t* il, const void* in, float value)
{
    const float v = rt(dt, value);
    for (uint32_t i = 0; i < il[0]; i++)
        for (uint32_t k = 0; k < il[1]; k++) out[at2(ol, i, k)] = ld(dt, in, at2(il, i, k)) <= v;
}

/* scatter (kernel/copy.metal:45-74): out[mask] = value, in place */
void
mco_scatter(int dt, const uint32_t* ol, void* out, const uint32_t* ml, const uint8_t* mask, float value)
{
    for (uint32_t i = 0; i < ol[0]; i++)
        for (uint32_t k = 0; k < ol[1]; k++)
            if (mask[at2(ml, i, k)]) st(dt, out, at2(ol, i, k), value);
}

/* gather (kernel/copy.metal:77-113): out[i,k] = in[i, index[i,k]]; dt 2 = int32 */
void
mco_gather(int dt, const uint32_t* ol, void* out, const uint32_t* il, const void* in,
           const uint32_t* xl, const int32_t* index)
{
    const size_t esz = dt == MCO_BF16 ? 2 : 4;
    for (uint32_t i = 0; i < xl[0]; i++)
        for (uint32_t k = 0; k < xl[1]; k++)
            memcpy((char*)out + esz * at2(ol, i, k),
                   (const char*)in + esz * at2(il, i, (uint32_t)index[at2(xl, i, k)]), esz);
}

/* sort (kernel/sort.metal:33-88, launch include/metalchat/kernel/sort.h:27-62): bitonic network
 * over ceil_pow2(dim) slots padded with -inf, DESCENDING, values and indices swapped together;
 * equal values are never swapped.  values / indices are [rows, aligned]; the caller slices
 * [:, :dim].  Pinned by test/test_kernel_sort.cc:17-50 (is_sorted + index consistency). */
void
mco_sort(int dt, const uint32_t* vl, void* values, const uint32_t* xl, int32_t* indices,
         const uint32_t* il, const void* in)
{
    const uint32_t rows = il[0], dim = il[1], aligned = vl[1];
    for (uint32_t b = 0; b < rows; b++) {
        for (uint32_t k = 0; k < aligned; k++) {
            st(dt, values, at2(vl, b, k), k < dim ? ld(dt, in, at2(il, b, k)) : -INFINITY);
            indices[at2(xl, b, k)] = (int32_t)k;
        }
        for (uint32_t k = 2; k <= aligned; k *= 2)
            for (uint32_t j = k >> 1; j > 0; j >>= 1)
                for (uint32_t i = 0; i < aligned; i++) {
                    const uint32_t ij = i ^ j;
                    if (i >= ij) continue;
                    const float vi = ld(dt, values, at2(vl, b, i)), vj = ld(dt, values, at2(vl, b, ij));
                    const int up = (i & k) == 0;
                    if ((up && vi < vj) || (!up && vi > vj)) {
                        st(dt, values, at2(vl, b, i), vj);
                        st(dt, values, at2(vl, b, ij), vi);
                        const int32_t t = indices[at2(xl, b, i)];
                        indices[at2(xl, b, i)] = indices[at2(xl, b, ij)];
                        indices[at2(xl, b, ij)] = t;
                    }
                }
    }
}

/* cumsum (kernel/cumsum.metal:24-76, launch include/metalchat/kernel/sum.h:28-58): thread t owns
 * BlockSize = max(2, ceil_pow2(ceil_div(dim, max_threads))) consecutive elements, prefix-sums
 * them IN T, then adds the totals of the threads before it one at a time (nearest first), each
 * add rounded to T.  Pinned by test/test_kernel_sum.cc:17-40 (float, margin 1e-4). */
void
mco_cumsum(int dt, const uint32_t* ol, void* out, const uint32_t* il, const void* in,
           uint32_t max_threads)
{
    const uint32_t rows = il[0], dim = il[1];
    uint32_t B = 1;
    while (B < ceil_div_u32(dim, max_threads)) B *= 2;
    if (B < 2) B = 2;
    const uint32_t nthreads = ceil_div_u32(dim, B);
    float* ls = (float*)malloc(sizeof(float) * (size_t)nthreads * B);
    float* gs = (float*)malloc(sizeof(float) * nthreads);
    for (uint32_t i = 0; i < rows; i++) {
        for (uint32_t t = 0; t < nthreads; t++) {
            const uint32_t begin = t * B, end = begin + B;
            const uint32_t bs = end > dim ? dim % B : B;
            for (uint32_t k = begin, j = 0; k < end && k < dim; k++, j++) {
                const float x = ld(dt, in, at2(il, i, k));
                ls[(size_t)t * B + j] = j > 0 ? rt(dt, x + ls[(size_t)t * B + j - 1]) : x;
            }
            gs[t] = ls[(size_t)t * B + bs - 1];
        }
        for (uint32_t t = 0; t < nthreads; t++) {
            const uint32_t begin = t * B, end = begin + B;
            const uint32_t bs = end > dim ? dim % B : B;
            for (uint32_t a = 1; a < nthreads; a++)
                if (t >= a)
                    for (uint32_t j = 0; j < bs; j++)
                        ls[(size_t)t * B + j] = rt(dt, ls[(size_t)t * B + j] + gs[t - a]);
            for (uint32_t k = begin; k < end && k < dim; k++)
                st(dt, out, at2(ol, i, k), ls[(size_t)t * B + (k - begin)]);
        }
    }
    free(ls);
    free(gs);
}

/* PCG32 (kernel/multinomial.metal:17-57) */
typedef struct {
    uint64_t state, inc;
} pcg32_t;

static uint32_t
pcg32_next(pcg32_t* g)
{
    const uint64_t pre = g->state;
    g->state = pre * 6364136223846793005ULL + g->inc;
    const uint32_t xorshifted = (uint32_t)(((pre >> 18u) ^ pre) >> 27u);
    const uint32_t rot = (uint32_t)(pre >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}

float
mco_pcg32_uniform(uint64_t init_state, uint64_t init_seq)
{
    pcg32_t g = {0, (init_seq << 1u) | 1u};
    pcg32_next(&g);
    g.state += init_state;
    pcg32_next(&g);
    const uint32_t u = (pcg32_next(&g) >> 9) | 0x3f800000u;
    float f;
    memcpy(&f, &u, 4);
    return f - 1.0f;
}

/* multinomial (kernel/multinomial.metal:60-122): output [rows, sample_size] of positions in a
 * DESCENDING row.  As in the reference, the lower end of the draw interval is read at column
 * output.size(1) - 1 of the INPUT (multinomial.metal:112): with sample_size == 1 that is column
 * 0, so a == b and every draw lands on position 0.  For sample_size > input dim the reference
 * reads out of bounds; its own test (test/test_kernel_multinomial.cc:17-55) only passes when
 * that read yields 0, which is what is used here. */
void
mco_multinomial(int dt, const uint32_t* ol, int32_t* out, const uint32_t* il, const void* in,
                uint64_t init_state, uint64_t init_seq)
{
    const uint32_t rows = ol[0], ns = ol[1], dim = il[1];
    for (uint32_t i = 0; i < rows; i++)
        for (uint32_t k = 0; k < ns; k++) {
            const float a = ns - 1 < dim ? ld(dt, in, at2(il, i, ns - 1)) : 0.0f;
            const float b = ld(dt, in, at2(il, i, 0));
            const float u = mco_pcg32_uniform(init_state + i, init_seq + k);
            const float random = rt(dt, u * (b - a) + a);
            int low = 0, high = (int)dim;
            while (low < high) {
                const uint32_t mid = (uint32_t)(low + high) / 2;
                if (ld(dt, in, at2(il, i, mid)) > random) low = (int)mid + 1;
                else high = (int)mid;
            }
            out[at2(ol, i, k)] = (low > 1 ? low : 1) - 1;
        }
}

/* topk_sampler (nn/sampling.h:216-258): std::partial_sort of the index row by value, descending.
 * The standard leaves the order of equal values unspecified; this restatement (and the HIP path)
 * break ties by the LOWER index first.  values_out / indices_out: [k]. */
void
mco_topk(int dt, const void* logits, const int32_t* indices, uint32_t n, uint32_t k,
         void* values_out, int32_t* indices_out)
{
    if (k > n) k = n;
    uint8_t* taken = (uint8_t*)calloc(n, 1);
    for (uint32_t r = 0; r < k; r++) {
        int64_t best = -1;
        float bv = 0.0f;
        for (uint32_t i = 0; i < n; i++) {
            if (taken[i]) continue;
            const float v = ld(dt, logits, i);
            if (best < 0 || v > bv) { best = i; bv = v; }
        }
        taken[best] = 1;
        st(dt, values_out, r, bv);
        indices_out[r] = indices ? indices[best] : (int32_t)best;
    }
    free(taken);
}

/* make_default_sampler (nn/sampling.h:303-313): topk(max(sample_size, 50)) -> nucleus(T(0.6),
 * T(0.9)) -> multinomial(sample_size), one row of logits; returns the sampled vocabulary id.
 * Intermediates (all T, length k) can be tapped through `taps` (7*k floats: scaled, probs, sorted,
 * cumsum, diff, masked, ids-as-float) when non-NULL. */
int32_t
mco_sample_default(int dt, const void* logits, uint32_t vocab, uint32_t top_k, float temperature,
                   float top_p, uint64_t init_state, uint64_t init_seq, float* taps)
{
    const uint32_t k = top_k < vocab ? top_k : vocab;
    uint32_t aligned = 1;
    while (aligned < k) aligned *= 2;
    const size_t esz = dt == MCO_BF16 ? 2 : 4;
    void* v0 = malloc(esz * aligned);
    void* v1 = malloc(esz * aligned);
    void* v2 = malloc(esz * aligned);
    void* v3 = malloc(esz * aligned);
    void* v4 = malloc(esz * aligned);
    int32_t* id0 = (int32_t*)malloc(4 * aligned);
    int32_t* sidx = (int32_t*)malloc(4 * aligned);
    int32_t* id1 = (int32_t*)malloc(4 * aligned);
    uint8_t* mask = (uint8_t*)malloc(aligned);
    const uint32_t lk[6] = {1, k, k, 1, 0, 0}, la[6] = {1, aligned, aligned, 1, 0, 0};
    mco_topk(dt, logits, NULL, vocab, k, v0, id0);
    /* nucleus_sampler::sample (nn/sampling.h:187-203) */
    const float temp_T = rt(dt, temperature);
    const float inv = rt(dt, 1.0f / temp_T);
    mco_bf16 mb = mco_f32_to_bf16(inv);
    mco_scalar_mul(dt, lk, v1, lk, v0, dt == MCO_BF16 ? (const void*)&mb : (const void*)&inv);
    mco_softmax(dt, lk, v2, lk, v1, 1024);
    mco_sort(dt, la, v3, la, sidx, lk, v2);
    mco_cumsum(dt, lk, v4, lk, v3, 1024);
    if (taps)
        for (uint32_t i = 0; i < k; i++) {
            taps[0 * k + i] = ld(dt, v1, i);
            taps[1 * k + i] = ld(dt, v2, i);
            taps[2 * k + i] = ld(dt, v3, i);
            taps[3 * k + i] = ld(dt, v4, i);
        }
    mco_sub(dt, lk, v1, lk, v4, lk, v3); /* probs_diff */
    mco_gt(dt, lk, mask, lk, v1, top_p);
    if (taps)
        for (uint32_t i = 0; i < k; i++) taps[4 * k + i] = ld(dt, v1, i);
    mco_scatter(dt, lk, v3, lk, mask, 0.0f);
    mco_gather(2, lk, id1, lk, id0, lk, sidx);
    if (taps)
        for (uint32_t i = 0; i < k; i++) {
            taps[5 * k + i] = ld(dt, v3, i);
            taps[6 * k + i] = (float)id1[i];
        }
    /* multinomial_sampler::sample (nn/sampling.h:284-292), sample_size 1 */
    int32_t pos = 0;
    const uint32_t l1[6] = {1, 1, 1, 1, 0, 0};
    mco_multinomial(dt, l1, &pos, lk, v3, init_state, init_seq);
    const int32_t token = id1[pos];
    free(v0); free(v1); free(v2); free(v3); free(v4); free(id0); free(sidx); free(id1); free(mask);
    return token;
}

/* ==========================================================================================
 * Model-level restatement: one transform(token, start_pos) with len == 1.
 *   nn::llama3::operator()        include/metalchat/nn/llama.h:113-134
 *   nn::gemma3::operator()        include/metalchat/nn/gemma.h:110-137
 *   nn::transformer::operator()   include/metalchat/nn/transformer.h:126-141
 *   nn::attention::operator()     include/metalchat/nn/attention.h:161-206
 *   nn::feed_forward::operator()  include/metalchat/nn/transformer.h:53-60
 *   nn::sink_cache::update/copy   include/metalchat/nn/cache.h:133-216
 *   nn::rope                      include/metalchat/nn/embedding.h:107-200
 *   quantization::lora_linear     include/metalchat/quantization/lora.h:94-122
 *   quantization::linear          include/metalchat/quantization/linear.h:45-55
 * Every intermediate the reference materialises as a T tensor is rounded to T here.
 * PARITY UNPINNED: the reference's integration tests assert nothing about logits.
 * ========================================================================================== */
struct mco_model {
    mco_model_options opt;
    mco_layer_weights* layers;
    int32_t emb_kind;
    const void* emb_weight;
    const float* emb_scales;
    const void* final_norm;
    mco_linear output;
    size_t esz;
    int32_t pre_len;
    /* per-layer KV caches [max_seq, n_kv, hd] of T, plus a scratch "new" buffer for the roll */
    void** k_cache;
    void** v_cache;
    void* roll_tmp;
    int32_t* end_pos;
    /* activations (T stored as float after rounding) */
    float* hidden_taps; /* (n_layers + 1) * dim */
};

static int g_threads = 0;
void
mco_set_num_threads(int n)
{
    g_threads = n;
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
}

static int32_t
bit_width_u32(uint32_t v)
{
    int32_t n = 0;
    while (v) {
        n++;
        v >>= 1;
    }
    return n;
}

/* y[o] = T(sum_k x[k] * W[o,k]) for the three linear kinds; x and y hold T values as floats */
static void
linear_apply(const int dt, const mco_linear* L, const float* x, float* y)
{
    const int32_t out = L->out_features, in = L->in_features;
    if (L->kind == 0) {
#pragma omp parallel for schedule(static)
        for (int32_t o = 0; o < out; o++) {
            float partial = 0.0f;
            if (dt == MCO_BF16) {
                const mco_bf16* w = (const mco_bf16*)L->weight + (size_t)o * in;
                for (int32_t k = 0; k < in; k++) partial += x[k] * mco_bf16_to_f32(w[k]);
            } else {
                const float* w = (const float*)L->weight + (size_t)o * in;
                for (int32_t k = 0; k < in; k++) partial += x[k] * w[k];
            }
            y[o] = rt(dt, partial);
        }
    } else {
        /* lora.h:105-117 / linear.h:50-54: Wd = T(q) * T(scale) evaluated in T (mul.metal:78-82),
         * then bmm in fp32 (bmm.metal:54-67). kind 1: scale per (row, group); kind 2: per row. */
        const int32_t G = L->kind == 1 ? L->group_size : in;
        const int32_t ng = in / G;
#pragma omp parallel for schedule(static)
        for (int32_t o = 0; o < out; o++) {
            const int8_t* q = (const int8_t*)L->weight + (size_t)o * in;
            float partial = 0.0f;
            for (int32_t g = 0; g < ng; g++) {
                const float s = rt(dt, L->scales[(size_t)o * ng + g]);
                for (int32_t k = g * G; k < (g + 1) * G; k++) {
                    const float wd = rt(dt, (float)q[k] * s);
                    partial += x[k] * wd;
                }
            }
            y[o] = rt(dt, partial);
        }
        if (L->kind == 1 && L->lora_rank > 0) {
            /* lora.h:119-121: result = output + mul(B(A(x)), scale) */
            const int32_t r = L->lora_rank;
            float* a = (float*)malloc(sizeof(float) * r);
            for (int32_t i = 0; i < r; i++) {
                float p = 0.0f;
                for (int32_t k = 0; k < in; k++)
                    p += x[k] * ld(dt, L->lora_a, (size_t)i * in + k);
                a[i] = rt(dt, p);
            }
            const float sc = rt(dt, L->lora_scale);
            for (int32_t o = 0; o < out; o++) {
                float p = 0.0f;
                for (int32_t i = 0; i < r; i++) p += a[i] * ld(dt, L->lora_b, (size_t)o * r + i);
                const float b = rt(dt, p);
                const float ad = rt(dt, b * sc);
                y[o] = rt(dt, y[o] + ad);
            }
            free(a);
        }
    }
}

/* rmsnorm over one row held as floats (values already T), weights in T memory */
static void
rmsnorm_row(const int dt, const float* x, const void* w, int32_t dim, float eps, float mu,
            float* y)
{
    const uint32_t max_threads = 1024;
    const uint32_t block = ceil_div_u32((uint32_t)dim, max_threads);
    const uint32_t nthreads = ceil_div_u32((uint32_t)dim, block);
    float partial[1024];
    for (uint32_t t = 0; t < nthreads; t++) {
        float s = 0.0f;
        for (uint32_t j = t * block; j < (t + 1) * block && j < (uint32_t)dim; j++)
            s += x[j] * x[j];
        partial[t] = s;
    }
    const float acc = threadgroup_sum(partial, nthreads);
    const float inv = 1.0f / sqrtf(acc / (float)dim + eps);
    for (int32_t j = 0; j < dim; j++) {
        const float weight = mu + ld(dt, w, (size_t)j);
        y[j] = rt(dt, weight * x[j] * inv);
    }
}

mco_model*
mco_model_create(const mco_model_options* opt, const mco_layer_weights* layers, int32_t emb_kind,
                 const void* emb_weight, const float* emb_scales, const void* final_norm,
                 const mco_linear* output)
{
    mco_model* m = (mco_model*)calloc(1, sizeof(mco_model));
    m->opt = *opt;
    m->layers = (mco_layer_weights*)malloc(sizeof(mco_layer_weights) * opt->n_layers);
    memcpy(m->layers, layers, sizeof(mco_layer_weights) * opt->n_layers);
    m->emb_kind = emb_kind;
    m->emb_weight = emb_weight;
    m->emb_scales = emb_scales;
    m->final_norm = final_norm;
    m->output = *output;
    m->esz = opt->dtype == MCO_BF16 ? 2 : 4;
    /* nn/cache.h:125-127: pre_len = bit_width(max_seq_len) - 1 */
    m->pre_len = opt->sink_pre_len >= 0 ? opt->sink_pre_len
                                        : bit_width_u32((uint32_t)opt->max_seq_len) - 1;
    const size_t cache_bytes =
        (size_t)opt->max_seq_len * opt->n_kv_heads * opt->head_dim * m->esz;
    m->k_cache = (void**)malloc(sizeof(void*) * opt->n_layers);
    m->v_cache = (void**)malloc(sizeof(void*) * opt->n_layers);
    for (int32_t i = 0; i < opt->n_layers; i++) {
        m->k_cache[i] = calloc(1, cache_bytes);
        m->v_cache[i] = calloc(1, cache_bytes);
    }
    m->roll_tmp = malloc(cache_bytes);
    m->end_pos = (int32_t*)calloc(opt->n_layers, sizeof(int32_t));
    m->hidden_taps = (float*)calloc((size_t)(opt->n_layers + 1) * opt->dim, sizeof(float));
    return m;
}

void
mco_model_destroy(mco_model* m)
{
    if (!m) return;
    for (int32_t i = 0; i < m->opt.n_layers; i++) {
        free(m->k_cache[i]);
        free(m->v_cache[i]);
    }
    free(m->k_cache);
    free(m->v_cache);
    free(m->roll_tmp);
    free(m->end_pos);
    free(m->hidden_taps);
    free(m->layers);
    free(m);
}

/* nn::sink_cache::copy (nn/cache.h:167-216) for len == 1.  `cache` is [max_seq, row] of T where
 * row = n_kv*hd.  Returns end_pos (the length of the returned view).
 * PARITY UNPINNED: no reference test reaches start_pos >= max_seq_len. */
static int32_t
sink_cache_update(mco_model* m, void* cache, const float* new_row, int32_t start_pos)
{
    const int dt = m->opt.dtype;
    const int32_t cache_size = m->opt.max_seq_len;
    const size_t row = (size_t)m->opt.n_kv_heads * m->opt.head_dim;
    const size_t rb = row * m->esz;
    const int32_t len = 1;
    const int32_t pre = m->pre_len, post = cache_size - pre;
    if (start_pos >= cache_size) {
        char* src = (char*)cache;
        char* dst = (char*)m->roll_tmp;
        /* prefix copy (cache.h:189-193) */
        memcpy(dst, src, (size_t)pre * rb);
        /* roll(cache_post -> cache_new_post, shift=len, dim=1) (cache.h:195-199, roll.metal) */
        for (int32_t p = 0; p < post; p++)
            memcpy(dst + (size_t)(pre + p) * rb, src + (size_t)(pre + (p + len) % post) * rb, rb);
        memcpy(src, dst, (size_t)cache_size * rb);
        start_pos = cache_size - len;
    }
    for (size_t j = 0; j < row; j++) st(dt, cache, (size_t)start_pos * row + j, new_row[j]);
    return start_pos + len;
}

/* Runs layers [layer_begin, layer_end) of one step.  The first stage embeds `token`; other stages
 * read the hidden row from hidden_in (T[dim]).  The last stage returns the greedy token (and the
 * logits); other stages write the hidden row to hidden_out and return -1.  This is how a layer
 * pipeline over several devices splits nn::llama3::operator() (include/metalchat/nn/llama.h:113-134):
 * the only state that crosses a stage boundary is the hidden row. */
int32_t
mco_model_step_range(mco_model* m, int32_t token, int32_t start_pos, int32_t layer_begin,
                     int32_t layer_end, const void* hidden_in, void* hidden_out, void* logits_out)
{
    const mco_model_options* o = &m->opt;
    const int dt = o->dtype;
    const int32_t dim = o->dim, H = o->n_heads, KV = o->n_kv_heads, hd = o->head_dim;
    const int32_t n_rep = H / KV, half = hd / 2;
    const float mu = o->family == 1 ? 1.0f : 0.0f;

    float* x = (float*)malloc(sizeof(float) * dim);
    float* hn = (float*)malloc(sizeof(float) * dim);
    float* q = (float*)malloc(sizeof(float) * H * hd);
    float* k = (float*)malloc(sizeof(float) * KV * hd);
    float* v = (float*)malloc(sizeof(float) * KV * hd);
    float* qr = (float*)malloc(sizeof(float) * H * hd);
    float* kr = (float*)malloc(sizeof(float) * KV * hd);
    float* att = (float*)malloc(sizeof(float) * H * hd);
    float* proj = (float*)malloc(sizeof(float) * dim);
    float* h1 = (float*)malloc(sizeof(float) * dim);
    float* g1 = (float*)malloc(sizeof(float) * o->ffn_dim);
    float* g3 = (float*)malloc(sizeof(float) * o->ffn_dim);
    float* ff = (float*)malloc(sizeof(float) * dim);
    float* scores = (float*)malloc(sizeof(float) * o->max_seq_len);
    float* partial = (float*)malloc(sizeof(float) * 1024);
    float* fcos = (float*)malloc(sizeof(float) * half * 2);
    float* fsin = (float*)malloc(sizeof(float) * half * 2);

    /* embedding (nn/embedding.h:82-86; quantization/lora.h:161-170 dequantises the table once) */
    if (layer_begin > 0) {
        for (int32_t j = 0; j < dim; j++) x[j] = ld(dt, hidden_in, (size_t)j);
    } else if (m->emb_kind == 0) {
        for (int32_t j = 0; j < dim; j++) x[j] = ld(dt, m->emb_weight, (size_t)token * dim + j);
    } else {
        const int8_t* qw = (const int8_t*)m->emb_weight + (size_t)token * dim;
        const float s = rt(dt, m->emb_scales[token]);
        for (int32_t j = 0; j < dim; j++) x[j] = rt(dt, (float)qw[j] * s);
    }
    if (o->family == 1 && layer_begin == 0) { /* nn/gemma.h:115 */
        const float sc = rt(dt, sqrtf((float)dim));
        for (int32_t j = 0; j < dim; j++) x[j] = rt(dt, x[j] * sc);
    }
    memcpy(m->hidden_taps, x, sizeof(float) * dim);

    /* rope tables for this position: table t = 0 global theta, 1 sliding theta */
    for (int t = 0; t < 2; t++) {
        const float theta = t == 0 ? o->rope_theta : o->rope_sliding_theta;
        if (theta <= 0.0f) continue;
        for (int32_t j = 0; j < half; j++)
            rope_angle((uint32_t)start_pos, (uint32_t)j, (uint32_t)hd, theta, &fcos[t * half + j],
                       &fsin[t * half + j]);
    }
    const float scale_T = rt(dt, o->attn_scale); /* attention.h:148 _M_scale(options.scale) is a T */

    for (int32_t li = layer_begin; li < layer_end; li++) {
        const mco_layer_weights* L = &m->layers[li];
        /* transformer.h:130 */
        rmsnorm_row(dt, x, L->attention_norm, dim, o->norm_eps, mu, hn);
        /* attention.h:170-172 */
        linear_apply(dt, &L->wq, hn, q);
        linear_apply(dt, &L->wk, hn, k);
        linear_apply(dt, &L->wv, hn, v);
        /* attention.h:174-175 (gemma q/k norm over head_dim) */
        if (L->q_norm) {
            for (int32_t h = 0; h < H; h++)
                rmsnorm_row(dt, q + h * hd, L->q_norm, hd, o->norm_eps, mu, q + h * hd);
            for (int32_t h = 0; h < KV; h++)
                rmsnorm_row(dt, k + h * hd, L->k_norm, hd, o->norm_eps, mu, k + h * hd);
        }
        /* rope (rope.metal:49-59) */
        const float* c = fcos + L->rope_table * half;
        const float* s = fsin + L->rope_table * half;
        for (int32_t h = 0; h < H; h++)
            for (int32_t j = 0; j < half; j++) {
                const float x1 = q[h * hd + j], x2 = q[h * hd + half + j];
                qr[h * hd + j] = rt(dt, c[j] * x1 - s[j] * x2);
                qr[h * hd + half + j] = rt(dt, s[j] * x1 + c[j] * x2);
            }
        for (int32_t h = 0; h < KV; h++)
            for (int32_t j = 0; j < half; j++) {
                const float x1 = k[h * hd + j], x2 = k[h * hd + half + j];
                kr[h * hd + j] = rt(dt, c[j] * x1 - s[j] * x2);
                kr[h * hd + half + j] = rt(dt, s[j] * x1 + c[j] * x2);
            }
        /* attention.h:177 */
        const int32_t S = sink_cache_update(m, m->k_cache[li], kr, start_pos);
        sink_cache_update(m, m->v_cache[li], v, start_pos);
        m->end_pos[li] = S;
        const size_t row = (size_t)KV * hd;
        /* attention.h:179-203: repeat_kv is a pure copy, indexed here as kv = h / n_reps
         * (functional/transform.h:20-90: repeat_interleave along the head dim) */
        for (int32_t h = 0; h < H; h++) {
            const int32_t kvh = h / n_rep;
            for (int32_t sp = 0; sp < S; sp++) {
                float p = 0.0f;
                for (int32_t d = 0; d < hd; d++)
                    p += qr[h * hd + d] * ld(dt, m->k_cache[li], (size_t)sp * row + kvh * hd + d);
                const float sc = rt(dt, p);          /* bmm -> T */
                scores[sp] = rt(dt, sc * scale_T);   /* scalar_mul in T */
            }
            /* softmax (softmax.metal) */
            const uint32_t block = ceil_div_u32((uint32_t)S, 1024);
            const uint32_t nthreads = ceil_div_u32((uint32_t)S, block);
            for (uint32_t t = 0; t < nthreads; t++) {
                float acc = 0.0f;
                for (uint32_t j = t * block; j < (t + 1) * block && j < (uint32_t)S; j++)
                    acc += exp_precise(scores[j]);
                partial[t] = acc;
            }
            const float exp_sum = 1.0f / threadgroup_sum(partial, nthreads);
            for (int32_t sp = 0; sp < S; sp++)
                scores[sp] = rt(dt, exp_precise(scores[sp]) * exp_sum);
            /* PV bmm */
            for (int32_t d = 0; d < hd; d++) {
                float p = 0.0f;
                for (int32_t sp = 0; sp < S; sp++)
                    p += scores[sp] * ld(dt, m->v_cache[li], (size_t)sp * row + kvh * hd + d);
                att[h * hd + d] = rt(dt, p);
            }
        }
        /* attention.h:205 */
        linear_apply(dt, &L->wo, att, proj);
        /* transformer.h:132-133 */
        if (L->attention_post_norm)
            rmsnorm_row(dt, proj, L->attention_post_norm, dim, o->norm_eps, mu, proj);
        for (int32_t j = 0; j < dim; j++) h1[j] = rt(dt, x[j] + proj[j]);
        /* transformer.h:135-139 */
        rmsnorm_row(dt, h1, L->ffn_norm, dim, o->norm_eps, mu, hn);
        linear_apply(dt, &L->w1, hn, g1);
        linear_apply(dt, &L->w3, hn, g3);
        for (int32_t j = 0; j < o->ffn_dim; j++) {
            const float a = o->family == 1 ? rt(dt, gelu_f(g1[j])) : silu_T(dt, g1[j]);
            g1[j] = rt(dt, a * g3[j]); /* hadamard in T */
        }
        linear_apply(dt, &L->w2, g1, ff);
        if (L->ffn_post_norm) rmsnorm_row(dt, ff, L->ffn_post_norm, dim, o->norm_eps, mu, ff);
        for (int32_t j = 0; j < dim; j++) x[j] = rt(dt, h1[j] + ff[j]);
        memcpy(m->hidden_taps + (size_t)(li + 1) * dim, x, sizeof(float) * dim);
    }

    int32_t best = -1;
    if (layer_end == o->n_layers) {
        /* llama.h:128-133 */
        rmsnorm_row(dt, x, m->final_norm, dim, o->norm_eps, mu, hn);
        float* logits = (float*)malloc(sizeof(float) * o->vocab);
        linear_apply(dt, &m->output, hn, logits);
        best = 0;
        for (int32_t i = 1; i < o->vocab; i++)
            if (logits[i] > logits[best]) best = i;
        if (logits_out)
            for (int32_t i = 0; i < o->vocab; i++) st(dt, logits_out, (size_t)i, logits[i]);
        free(logits);
    } else if (hidden_out) {
        for (int32_t j = 0; j < dim; j++) st(dt, hidden_out, (size_t)j, x[j]);
    }

    free(x); free(hn); free(q); free(k); free(v); free(qr); free(kr); free(att); free(proj);
    free(h1); free(g1); free(g3); free(ff); free(scores); free(partial); free(fcos); free(fsin);
    return best;
}

/* nn::sink_cache::copy (nn/cache.h:167-216) for any len <= cache size: rows [len][row] floats. */
static int32_t
sink_cache_update_n(mco_model* m, void* cache, const float* rows, int32_t len, int32_t start_pos)
{
    const int dt = m->opt.dtype;
    const int32_t cache_size = m->opt.max_seq_len;
    const size_t row = (size_t)m->opt.n_kv_heads * m->opt.head_dim;
    const size_t rb = row * m->esz;
    const int32_t pre = m->pre_len, post = cache_size - pre;
    if (start_pos >= cache_size) {
        char* src = (char*)cache;
        char* dst = (char*)m->roll_tmp;
        memcpy(dst, src, (size_t)pre * rb);
        for (int32_t p = 0; p < post; p++)
            memcpy(dst + (size_t)(pre + p) * rb, src + (size_t)(pre + (p + len) % post) * rb, rb);
        memcpy(src, dst, (size_t)cache_size * rb);
        start_pos = cache_size - len;
    }
    for (int32_t r = 0; r < len; r++)
        for (size_t j = 0; j < row; j++)
            st(dt, cache, (size_t)(start_pos + r) * row + j, rows[(size_t)r * row + j]);
    return start_pos + len;
}

/* nn::llama3::operator() / nn::gemma3::operator() for an input of `len` tokens
 * (include/metalchat/nn/llama.h:113-134, nn/gemma.h:110-137): the prompt pass.  Row-wise layers
 * are the same arithmetic as the one-token step applied to each row; what is new is
 *   - rope rows start_pos .. start_pos+len-1 (kernel/rope.metal:46-59, pos = row / n_head),
 *   - the cache write of len rows (nn/cache.h:205-213),
 *   - the mask: make_causal_mask(len, end_pos) (nn/attention.h:283-299) is -inf everywhere
 *     except the lower triangle of its LAST len columns, so with start_pos > 0 the earlier cache
 *     columns stay masked -- reproduced as is; sliding layers (gemma) add the second triangle
 *     (nn/attention.h:302-321): column c of the square is visible from row r iff r-window < c <= r;
 *     scores = T(T(T(q.k) * scale) + mask) (attention.h:197-201), softmax without max shift,
 *   - only the last row goes through the output head (llama.h:130-133).
 * len == 1 takes no mask at all (make_causal_mask returns nullopt) and equals mco_model_step.
 * PARITY UNPINNED: the reference asserts nothing about prompt logits.  Returns the greedy token. */
int32_t
mco_model_forward(mco_model* m, const int32_t* tokens, int32_t len, int32_t start_pos,
                  int32_t sliding_window, void* logits_out)
{
    const mco_model_options* o = &m->opt;
    const int dt = o->dtype;
    const int32_t dim = o->dim, H = o->n_heads, KV = o->n_kv_heads, hd = o->head_dim, ffn = o->ffn_dim;
    const int32_t n_rep = H / KV, half = hd / 2;
    const float mu = o->family == 1 ? 1.0f : 0.0f;
    const size_t row = (size_t)KV * hd;
    /* nn/cache.h:205-213 + kernel/copy.h:38-39: a chunk that starts inside the cache and ends outside has a
     * clamped target slice whose element count differs from the input's -- clone throws invalid_argument; so does
     * len > cache size (cache.h:178-183).  Nothing is touched. */
    if (len > o->max_seq_len || (start_pos < o->max_seq_len && start_pos + len > o->max_seq_len)) return -2;
#define NEW(n) (float*)malloc(sizeof(float) * (size_t)(n))
    float *x = NEW((size_t)len * dim), *hn = NEW((size_t)len * dim), *q = NEW((size_t)len * H * hd);
    float *k = NEW((size_t)len * row), *v = NEW((size_t)len * row), *att = NEW((size_t)len * H * hd);
    float *proj = NEW((size_t)len * dim), *h1 = NEW((size_t)len * dim);
    float *g1 = NEW((size_t)len * ffn), *g3 = NEW((size_t)len * ffn), *ff = NEW((size_t)len * dim);
    float *scores = NEW(o->max_seq_len), *partial = NEW(1024), *fcos = NEW((size_t)2 * len * half),
          *fsin = NEW((size_t)2 * len * half);
#undef NEW
    for (int32_t r = 0; r < len; r++) {
        float* xr = x + (size_t)r * dim;
        const int32_t token = tokens[r];
        if (m->emb_kind == 0) {
            for (int32_t j = 0; j < dim; j++) xr[j] = ld(dt, m->emb_weight, (size_t)token * dim + j);
        } else {
            const int8_t* qw = (const int8_t*)m->emb_weight + (size_t)token * dim;
            const float s = rt(dt, m->emb_scales[token]);
            for (int32_t j = 0; j < dim; j++) xr[j] = rt(dt, (float)qw[j] * s);
        }
        if (o->family == 1) {
            const float sc = rt(dt, sqrtf((float)dim));
            for (int32_t j = 0; j < dim; j++) xr[j] = rt(dt, xr[j] * sc);
        }
        for (int t = 0; t < 2; t++) {
            const float theta = t == 0 ? o->rope_theta : o->rope_sliding_theta;
            if (theta <= 0.0f) continue;
            for (int32_t j = 0; j < half; j++)
                rope_angle((uint32_t)(start_pos + r), (uint32_t)j, (uint32_t)hd, theta,
                           &fcos[((size_t)t * len + r) * half + j], &fsin[((size_t)t * len + r) * half + j]);
        }
    }
    const float scale_T = rt(dt, o->attn_scale);
    for (int32_t li = 0; li < o->n_layers; li++) {
        const mco_layer_weights* L = &m->layers[li];
#pragma omp parallel for schedule(dynamic)
        for (int32_t r = 0; r < len; r++) {
            float* hr = hn + (size_t)r * dim;
            rmsnorm_row(dt, x + (size_t)r * dim, L->attention_norm, dim, o->norm_eps, mu, hr);
            float* qq = q + (size_t)r * H * hd;
            float* kk = k + (size_t)r * row;
            linear_apply(dt, &L->wq, hr, qq);
            linear_apply(dt, &L->wk, hr, kk);
            linear_apply(dt, &L->wv, hr, v + (size_t)r * row);
            if (L->q_norm) {
                for (int32_t h = 0; h < H; h++) rmsnorm_row(dt, qq + h * hd, L->q_norm, hd, o->norm_eps, mu, qq + h * hd);
                for (int32_t h = 0; h < KV; h++) rmsnorm_row(dt, kk + h * hd, L->k_norm, hd, o->norm_eps, mu, kk + h * hd);
            }
            const float* c = fcos + ((size_t)L->rope_table * len + r) * half;
            const float* s = fsin + ((size_t)L->rope_table * len + r) * half;
            for (int32_t h = 0; h < H + KV; h++) {
                float* p = h < H ? qq + h * hd : kk + (h - H) * hd;
                for (int32_t j = 0; j < half; j++) {
                    const float x1 = p[j], x2 = p[half + j];
                    p[j] = rt(dt, c[j] * x1 - s[j] * x2);
                    p[half + j] = rt(dt, s[j] * x1 + c[j] * x2);
                }
            }
        }
        const int32_t S = sink_cache_update_n(m, m->k_cache[li], k, len, start_pos);
        sink_cache_update_n(m, m->v_cache[li], v, len, start_pos);
        m->end_pos[li] = S;
        const int sliding = L->rope_table == 1 && sliding_window > 0;
        for (int32_t r = 0; r < len; r++) {
            for (int32_t h = 0; h < H; h++) {
                const int32_t kvh = h / n_rep;
                const float* qq = q + ((size_t)r * H + h) * hd;
                for (int32_t sp = 0; sp < S; sp++) {
                    float p = 0.0f;
                    for (int32_t d = 0; d < hd; d++) p += qq[d] * ld(dt, m->k_cache[li], (size_t)sp * row + kvh * hd + d);
                    float sc = rt(dt, rt(dt, p) * scale_T);
                    if (len > 1) {
                        /* mask [len, S]: column sp belongs to the square iff sp >= S - len */
                        const int32_t cc = sp - (S - len);
                        float mk = -INFINITY;
                        if (cc >= 0 && cc <= r) mk = 0.0f;                                   /* upper: triu(.., 1) */
                        if (sliding) {
                            const float lower = (cc >= 0 && r < sliding_window + cc) ? 0.0f : -INFINITY;
                            mk = rt(dt, mk + lower);                                        /* add(upper, lower) */
                        }
                        sc = rt(dt, sc + mk);
                    }
                    scores[sp] = sc;
                }
                const uint32_t block = ceil_div_u32((uint32_t)S, 1024);
                const uint32_t nthreads = ceil_div_u32((uint32_t)S, block);
                for (uint32_t t = 0; t < nthreads; t++) {
                    float acc = 0.0f;
                    for (uint32_t j = t * block; j < (t + 1) * block && j < (uint32_t)S; j++) acc += exp_precise(scores[j]);
                    partial[t] = acc;
                }
                const float exp_sum = 1.0f / threadgroup_sum(partial, nthreads);
                for (int32_t sp = 0; sp < S; sp++) scores[sp] = rt(dt, exp_precise(scores[sp]) * exp_sum);
                for (int32_t d = 0; d < hd; d++) {
                    float p = 0.0f;
                    for (int32_t sp = 0; sp < S; sp++) p += scores[sp] * ld(dt, m->v_cache[li], (size_t)sp * row + kvh * hd + d);
                    att[((size_t)r * H + h) * hd + d] = rt(dt, p);
                }
            }
        }
#pragma omp parallel for schedule(dynamic)
        for (int32_t r = 0; r < len; r++) {
            float* pr = proj + (size_t)r * dim;
            float* hr = h1 + (size_t)r * dim;
            float* xr = x + (size_t)r * dim;
            linear_apply(dt, &L->wo, att + (size_t)r * H * hd, pr);
            if (L->attention_post_norm) rmsnorm_row(dt, pr, L->attention_post_norm, dim, o->norm_eps, mu, pr);
            for (int32_t j = 0; j < dim; j++) hr[j] = rt(dt, xr[j] + pr[j]);
            float* nr = hn + (size_t)r * dim;
            rmsnorm_row(dt, hr, L->ffn_norm, dim, o->norm_eps, mu, nr);
            float* a1 = g1 + (size_t)r * ffn;
            float* a3 = g3 + (size_t)r * ffn;
            linear_apply(dt, &L->w1, nr, a1);
            linear_apply(dt, &L->w3, nr, a3);
            for (int32_t j = 0; j < ffn; j++) {
                const float a = o->family == 1 ? rt(dt, gelu_f(a1[j])) : silu_T(dt, a1[j]);
                a1[j] = rt(dt, a * a3[j]);
            }
            float* fr = ff + (size_t)r * dim;
            linear_apply(dt, &L->w2, a1, fr);
            if (L->ffn_post_norm) rmsnorm_row(dt, fr, L->ffn_post_norm, dim, o->norm_eps, mu, fr);
            for (int32_t j = 0; j < dim; j++) xr[j] = rt(dt, hr[j] + fr[j]);
        }
        /* tap: the LAST row after this layer (what a following decode step continues from) */
        memcpy(m->hidden_taps + (size_t)(li + 1) * dim, x + (size_t)(len - 1) * dim, sizeof(float) * dim);
    }
    rmsnorm_row(dt, x + (size_t)(len - 1) * dim, m->final_norm, dim, o->norm_eps, mu, hn);
    float* logits = (float*)malloc(sizeof(float) * o->vocab);
    linear_apply(dt, &m->output, hn, logits);
    int32_t best = 0;
    for (int32_t i = 1; i < o->vocab; i++)
        if (logits[i] > logits[best]) best = i;
    if (logits_out)
        for (int32_t i = 0; i < o->vocab; i++) st(dt, logits_out, (size_t)i, logits[i]);
    free(logits);
    free(x); free(hn); free(q); free(k); free(v); free(att); free(proj); free(h1); free(g1); free(g3);
    free(ff); free(scores); free(partial); free(fcos); free(fsin);
    return best;
}

int32_t
mco_model_step(mco_model* m, int32_t token, int32_t start_pos, void* logits_out)
{
    return mco_model_step_range(m, token, start_pos, 0, m->opt.n_layers, NULL, NULL, logits_out);
}

void
mco_model_get_hidden(const mco_model* m, int32_t layer, void* out)
{
    const float* src = m->hidden_taps + (size_t)(layer + 1) * m->opt.dim;
    for (int32_t j = 0; j < m->opt.dim; j++) st(m->opt.dtype, out, (size_t)j, src[j]);
}

/* Test aid (no reference counterpart): fill layer `layer`'s cache with n logical rows [n, n_kv, hd] of T,
 * as if positions 0 .. n-1 had been decoded -- what sink_cache::copy's first branch leaves behind
 * (nn/cache.h:206-213).  Lets a parity test start at the benchmark's context length without
 * thousands of oracle steps; the GPU side takes the same rows through mc_decoder_import_kv. */
int32_t
mco_model_set_kv(mco_model* m, int32_t layer, const void* keys, const void* values, int32_t n)
{
    if (n < 0 || n > m->opt.max_seq_len || layer < 0 || layer >= m->opt.n_layers) return -1;
    const size_t rb = (size_t)m->opt.n_kv_heads * m->opt.head_dim * m->esz;
    memcpy(m->k_cache[layer], keys, (size_t)n * rb);
    memcpy(m->v_cache[layer], values, (size_t)n * rb);
    m->end_pos[layer] = n;
    return n;
}

int32_t
mco_model_get_kv(const mco_model* m, int32_t layer, void* keys_out, void* values_out)
{
    const size_t rb = (size_t)m->opt.n_kv_heads * m->opt.head_dim * m->esz;
    const int32_t n = m->end_pos[layer];
    memcpy(keys_out, m->k_cache[layer], (size_t)n * rb);
    memcpy(values_out, m->v_cache[layer], (size_t)n * rb);
    return n;
}
