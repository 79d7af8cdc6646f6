// Reference-named kernels: one gfx950 kernel per Metal kernel of the decode hot path, with the
// SAME host name, the SAME positional argument list (tensor_layout<N> by value, then the buffer)
// and the SAME dispatchThreads geometry, so the reference's kernel wrappers
// (include/metalchat/kernel/*.h) can drive them unchanged through the encoder seam.
//
// Written for CDNA4, not translated: 64-wide wavefront reductions, LDS only where a tile is
// reused, fp64-evaluated "precise" transcendentals (see common.h).  The hot decode path does not
// go through these one-op kernels -- it uses the fused kernels in decode_kernels.hip -- but every
// kernel here is parity-tested against the oracle on its own (tests/test_ref_kernels_gpu.py).
//
// Name mangling follows kernel/kernel.h:30-90: {kernel}[_{block}]_{type}[_{type}...].

#include "common.h"

using namespace mc;

// ------------------------------------------------------------------------------------------
// bmm_8_{bfloat,float}  (semantics of kernel/bmm.metal:25-82; launch include/metalchat/kernel/bmm.h:43-47:
// grid <ceil8(M), ceil8(N), batch> threads, group <8,8>).  One 64-lane wavefront owns an 8x8
// output tile: x -> row inside the tile, y -> column.  Operands are staged through LDS in
// 8 x 32 panels so each lane issues one global read per 4 MACs instead of two per MAC.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void
bmm8_body(const layout3& ol, typename T::S* out, const layout3& al, const typename T::S* a,
          const layout3& bl, const typename T::S* b)
{
    constexpr int BS = 8, KP = 32;
    __shared__ float a_s[BS][KP + 1];
    __shared__ float b_s[KP][BS + 1];

    const uint32_t M = al.sizes[1], K = al.sizes[2], N = bl.sizes[2];
    const uint32_t batch = blockIdx.z;
    const uint32_t tr = threadIdx.x, tc = threadIdx.y;
    const uint32_t row = blockIdx.x * BS + tr, col = blockIdx.y * BS + tc;
    const uint32_t lin = tc * BS + tr; // 0..63

    float partial = 0.0f;
    for (uint32_t k0 = 0; k0 < K; k0 += KP) {
        // A panel: 8 rows x 32 k  (4 elements per lane)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t e = lin + 64 * i, r = e / KP, kk = e % KP;
            const uint32_t gr = blockIdx.x * BS + r, gk = k0 + kk;
            a_s[r][kk] = (gr < M && gk < K) ? T::ld(a[at(al, batch, gr, gk)]) : 0.0f;
        }
        // B panel: 32 k x 8 cols
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t e = lin + 64 * i, kk = e / BS, c = e % BS;
            const uint32_t gk = k0 + kk, gc = blockIdx.y * BS + c;
            b_s[kk][c] = (gk < K && gc < N) ? T::ld(b[at(bl, batch, gk, gc)]) : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KP; kk++) partial += a_s[tr][kk] * b_s[kk][tc];
        __syncthreads();
    }
    if (row < M && col < N) out[at(ol, batch, row, col)] = T::st(partial);
}

extern "C" __global__ void
bmm_8_bfloat(layout3 ol, bf16_t* out, layout3 al, const bf16_t* a, layout3 bl, const bf16_t* b)
{
    bmm8_body<BF>(ol, out, al, a, bl, b);
}
extern "C" __global__ void
bmm_8_float(layout3 ol, float* out, layout3 al, const float* a, layout3 bl, const float* b)
{
    bmm8_body<F32>(ol, out, al, a, bl, b);
}

// ------------------------------------------------------------------------------------------
// 2-D elementwise family.  Geometry = make_kernel_grid_2d (src/kernel.cc:13-37): x walks the last
// dimension, y the rows.
// ------------------------------------------------------------------------------------------
#define MC_IJ                                                          \
    const uint32_t i = blockIdx.y * blockDim.y + threadIdx.y;          \
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;

// hadamard (kernel/mul.metal:13-48): product evaluated in T
template <typename T>
__device__ __forceinline__ void
hadamard_body(const layout2& ol, typename T::S* out, const layout2& al, const typename T::S* a,
              const layout2& bl, const typename T::S* b)
{
    MC_IJ;
    if (i < al.sizes[0] && k < al.sizes[1])
        out[at(ol, i, k)] = T::st(T::ld(a[at(al, i, k)]) * T::ld(b[at(bl, i, k)]));
}
extern "C" __global__ void
hadamard_bfloat(layout2 ol, bf16_t* out, layout2 al, const bf16_t* a, layout2 bl, const bf16_t* b)
{
    hadamard_body<BF>(ol, out, al, a, bl, b);
}
extern "C" __global__ void
hadamard_float(layout2 ol, float* out, layout2 al, const float* a, layout2 bl, const float* b)
{
    hadamard_body<F32>(ol, out, al, a, bl, b);
}

// hadamard_broadcast (kernel/mul.metal:51-85), the dequantizer:
//   out[i,j] = O(in1[i,j]) * O(in2[i % n])
template <typename O, typename S2>
__device__ __forceinline__ void
hadamard_broadcast_body(const layout2& ol, typename O::S* out, const layout2& il, const int8_t* in1,
                        const layout1& sl, const typename S2::S* in2)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i < il.sizes[0] && j < il.sizes[1]) {
        const float q = O::rt((float)in1[at(il, i, j)]);
        const float s = O::rt(S2::ld(in2[at(sl, i % sl.sizes[0])]));
        out[at(ol, i, j)] = O::st(q * s);
    }
}
extern "C" __global__ void
hadamard_broadcast_bfloat_int8_t_bfloat(layout2 ol, bf16_t* out, layout2 il, const int8_t* in1,
                                        layout1 sl, const bf16_t* in2)
{
    hadamard_broadcast_body<BF, BF>(ol, out, il, in1, sl, in2);
}
extern "C" __global__ void
hadamard_broadcast_bfloat_int8_t_float(layout2 ol, bf16_t* out, layout2 il, const int8_t* in1,
                                       layout1 sl, const float* in2)
{
    hadamard_broadcast_body<BF, F32>(ol, out, il, in1, sl, in2);
}
extern "C" __global__ void
hadamard_broadcast_float_int8_t_bfloat(layout2 ol, float* out, layout2 il, const int8_t* in1,
                                       layout1 sl, const bf16_t* in2)
{
    hadamard_broadcast_body<F32, BF>(ol, out, il, in1, sl, in2);
}
extern "C" __global__ void
hadamard_broadcast_float_int8_t_float(layout2 ol, float* out, layout2 il, const int8_t* in1,
                                      layout1 sl, const float* in2)
{
    hadamard_broadcast_body<F32, F32>(ol, out, il, in1, sl, in2);
}

// scalar_mul (kernel/mul.metal:88-121): in * multiplier evaluated in T
extern "C" __global__ void
scalar_mul_bfloat(layout2 ol, bf16_t* out, layout2 il, const bf16_t* in, bf16_t multiplier)
{
    MC_IJ;
    if (i < il.sizes[0] && k < il.sizes[1])
        out[at(ol, i, k)] = f2bf(bf2f(in[at(il, i, k)]) * bf2f(multiplier));
}
extern "C" __global__ void
scalar_mul_float(layout2 ol, float* out, layout2 il, const float* in, float multiplier)
{
    MC_IJ;
    if (i < il.sizes[0] && k < il.sizes[1]) out[at(ol, i, k)] = in[at(il, i, k)] * multiplier;
}

// add / add_broadcast (kernel/arithmetic.metal:13-85)
template <typename T>
__device__ __forceinline__ void
add_body(const layout2& ol, typename T::S* out, const layout2& al, const typename T::S* a,
         const layout2& bl, const typename T::S* b)
{
    MC_IJ;
    if (i < al.sizes[0] && k < al.sizes[1])
        out[at(ol, i, k)] = T::st(T::ld(a[at(al, i, k)]) + T::ld(b[at(bl, i, k)]));
}
extern "C" __global__ void
add_bfloat(layout2 ol, bf16_t* out, layout2 al, const bf16_t* a, layout2 bl, const bf16_t* b)
{
    add_body<BF>(ol, out, al, a, bl, b);
}
extern "C" __global__ void
add_float(layout2 ol, float* out, layout2 al, const float* a, layout2 bl, const float* b)
{
    add_body<F32>(ol, out, al, a, bl, b);
}

template <typename T>
__device__ __forceinline__ void
add_broadcast_body(const layout2& ol, typename T::S* out, const layout2& al,
                   const typename T::S* a, const layout1& bl, const typename T::S* b)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i < al.sizes[0] && j < al.sizes[1])
        out[at(ol, i, j)] = T::st(T::ld(a[at(al, i, j)]) + T::ld(b[at(bl, j % bl.sizes[0])]));
}
extern "C" __global__ void
add_broadcast_bfloat(layout2 ol, bf16_t* out, layout2 al, const bf16_t* a, layout1 bl,
                     const bf16_t* b)
{
    add_broadcast_body<BF>(ol, out, al, a, bl, b);
}
extern "C" __global__ void
add_broadcast_float(layout2 ol, float* out, layout2 al, const float* a, layout1 bl, const float* b)
{
    add_broadcast_body<F32>(ol, out, al, a, bl, b);
}

// copy (kernel/copy.metal:20-42)
template <typename S>
__device__ __forceinline__ void
copy_body(const layout2& ol, S* out, const layout2& il, const S* in)
{
    MC_IJ;
    if (i < il.sizes[0] && k < il.sizes[1]) out[at(ol, i, k)] = in[at(il, i, k)];
}
extern "C" __global__ void
copy_bfloat(layout2 ol, bf16_t* out, layout2 il, const bf16_t* in)
{
    copy_body(ol, out, il, in);
}
extern "C" __global__ void
copy_float(layout2 ol, float* out, layout2 il, const float* in)
{
    copy_body(ol, out, il, in);
}
extern "C" __global__ void
copy_int32_t(layout2 ol, int32_t* out, layout2 il, const int32_t* in)
{
    copy_body(ol, out, il, in);
}

// silu (kernel/activation.metal:13-41): x / (T(1) + T(exp(-x))), every step a T value
template <typename T>
__device__ __forceinline__ float
silu_T(float x)
{
    const float e = T::rt(exp_precise(-x));
    const float d = T::rt(1.0f + e);
    return T::rt(x / d);
}
template <typename T>
__device__ __forceinline__ void
silu_body(const layout2& ol, typename T::S* out, const layout2& il, const typename T::S* in)
{
    MC_IJ;
    if (i < il.sizes[0] && k < il.sizes[1])
        out[at(ol, i, k)] = T::st(silu_T<T>(T::ld(in[at(il, i, k)])));
}
extern "C" __global__ void
silu_bfloat(layout2 ol, bf16_t* out, layout2 il, const bf16_t* in)
{
    silu_body<BF>(ol, out, il, in);
}
extern "C" __global__ void
silu_float(layout2 ol, float* out, layout2 il, const float* in)
{
    silu_body<F32>(ol, out, il, in);
}

// gelu (kernel/activation.metal:44-78): tanh approximation in fp32
__device__ __forceinline__ float
gelu_f(float x)
{
    const float beta = 1.41421356237309504880f * 1.12837916709551257390f * 0.5f;
    const float kappa = 0.044715f;
    const float x3 = x * x * x;
    const float inner = beta * (x + kappa * x3);
    return 0.5f * x * (1.0f + (float)tanh((double)inner));
}
template <typename T>
__device__ __forceinline__ void
gelu_body(const layout2& ol, typename T::S* out, const layout2& il, const typename T::S* in)
{
    MC_IJ;
    if (i < il.sizes[0] && k < il.sizes[1])
        out[at(ol, i, k)] = T::st(gelu_f(T::ld(in[at(il, i, k)])));
}
extern "C" __global__ void
gelu_bfloat(layout2 ol, bf16_t* out, layout2 il, const bf16_t* in)
{
    gelu_body<BF>(ol, out, il, in);
}
extern "C" __global__ void
gelu_float(layout2 ol, float* out, layout2 il, const float* in)
{
    gelu_body<F32>(ol, out, il, in);
}

// ------------------------------------------------------------------------------------------
// rmsnorm (kernel/rmsnorm.metal:28-98).  One workgroup per row; thread t owns the contiguous
// slice [t*block, (t+1)*block).  Reduction: 64-lane shuffle tree, then up to 16 wave partials.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void
rmsnorm_body(const layout2& ol, typename T::S* out, const layout2& il, const typename T::S* in,
             const layout1& wl, const typename T::S* w, float eps, float mu, uint32_t block)
{
    __shared__ float red[16];
    const uint32_t dim = il.sizes[1], i = blockIdx.x;
    const uint32_t begin = threadIdx.x * block, end = begin + block;
    float s = 0.0f;
    for (uint32_t j = begin; j < end && j < dim; j++) {
        const float x = T::ld(in[at(il, i, j)]);
        s += x * x;
    }
    const float acc = block_sum(s, red);
    const float inv = 1.0f / sqrtf(acc / (float)dim + eps);
    for (uint32_t j = begin; j < end && j < dim; j++) {
        const float x = T::ld(in[at(il, i, j)]);
        const float weight = mu + T::ld(w[at(wl, j)]);
        out[at(ol, i, j)] = T::st(weight * x * inv);
    }
}
extern "C" __global__ void
rmsnorm_bfloat(layout2 ol, bf16_t* out, layout2 il, const bf16_t* in, layout1 wl, const bf16_t* w,
               float eps, float mu, uint32_t block)
{
    rmsnorm_body<BF>(ol, out, il, in, wl, w, eps, mu, block);
}
extern "C" __global__ void
rmsnorm_float(layout2 ol, float* out, layout2 il, const float* in, layout1 wl, const float* w,
              float eps, float mu, uint32_t block)
{
    rmsnorm_body<F32>(ol, out, il, in, wl, w, eps, mu, block);
}

// softmax (kernel/softmax.metal:24-88): exp(x) / sum(exp(x)), NO max subtraction
template <typename T>
__device__ __forceinline__ void
softmax_body(const layout2& ol, typename T::S* out, const layout2& il, const typename T::S* in,
             uint32_t block)
{
    __shared__ float red[16];
    const uint32_t dim = il.sizes[1], i = blockIdx.x;
    const uint32_t begin = threadIdx.x * block, end = begin + block;
    float s = 0.0f;
    for (uint32_t j = begin; j < end && j < dim; j++) s += exp_precise(T::ld(in[at(il, i, j)]));
    const float exp_sum = 1.0f / block_sum(s, red);
    for (uint32_t j = begin; j < end && j < dim; j++)
        out[at(ol, i, j)] = T::st(exp_precise(T::ld(in[at(il, i, j)])) * exp_sum);
}
extern "C" __global__ void
softmax_bfloat(layout2 ol, bf16_t* out, layout2 il, const bf16_t* in, uint32_t block)
{
    softmax_body<BF>(ol, out, il, in, block);
}
extern "C" __global__ void
softmax_float(layout2 ol, float* out, layout2 il, const float* in, uint32_t block)
{
    softmax_body<F32>(ol, out, il, in, block);
}

// ------------------------------------------------------------------------------------------
// rope (kernel/rope.metal:29-63).  `half` = f_cos.size(1) = head_dim / 2; the host launches
// head_dim threads per row (include/metalchat/kernel/embedding.h:115-116) and only k < half work.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void
rope_body(const layout2& ol, typename T::S* out, const layout2& il, const typename T::S* in,
          const layout2& cl, const float* fcos, const layout2& sl, const float* fsin,
          uint32_t batch_size, uint32_t n_head, uint32_t start_pos)
{
    MC_IJ;
    const uint32_t half = cl.sizes[1];
    if (i < il.sizes[0] && k < half) {
        const uint32_t pos = i / (batch_size * n_head);
        const float x1 = T::ld(in[at(il, i, k)]);
        const float x2 = T::ld(in[at(il, i, half + k)]);
        const float c = fcos[at(cl, start_pos + pos, k)];
        const float s = fsin[at(sl, start_pos + pos, k)];
        out[at(ol, i, k)] = T::st(c * x1 - s * x2);
        out[at(ol, i, half + k)] = T::st(s * x1 + c * x2);
    }
}
extern "C" __global__ void
rope_bfloat(layout2 ol, bf16_t* out, layout2 il, const bf16_t* in, layout2 cl, const float* fcos,
            layout2 sl, const float* fsin, uint32_t batch_size, uint32_t n_head, uint32_t start_pos)
{
    rope_body<BF>(ol, out, il, in, cl, fcos, sl, fsin, batch_size, n_head, start_pos);
}
extern "C" __global__ void
rope_float(layout2 ol, float* out, layout2 il, const float* in, layout2 cl, const float* fcos,
           layout2 sl, const float* fsin, uint32_t batch_size, uint32_t n_head, uint32_t start_pos)
{
    rope_body<F32>(ol, out, il, in, cl, fcos, sl, fsin, batch_size, n_head, start_pos);
}

// rope_freqs (kernel/rope.metal:77-102)
extern "C" __global__ void
rope_freqs_float(layout2 cl, float* fcos, layout2 sl, float* fsin, uint32_t dim,
                 uint32_t start_pos, float theta)
{
    const uint32_t i = blockIdx.y * blockDim.y + threadIdx.y;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cl.sizes[0] && j < dim / 2) {
        const float e = 2.0f * (float)j / (float)dim;
        const float freq = 1.0f / (float)pow((double)theta, (double)e);
        const float angle = (float)(start_pos + i) * freq;
        fcos[at(cl, i, j)] = (float)cos((double)angle);
        fsin[at(sl, i, j)] = (float)sin((double)angle);
    }
}

// ------------------------------------------------------------------------------------------
// embedding (kernel/embedding.metal:38-70).  grid x: positions (blocked), y: embedding dim, z: batch
// ------------------------------------------------------------------------------------------
template <typename S>
__device__ __forceinline__ void
embedding_body(const layout3& ol, S* out, const layout2& il, const int32_t* in, const layout2& wl,
               const S* w, uint32_t block)
{
    const uint32_t dim_size = il.sizes[1], emb = wl.sizes[1], i = blockIdx.z;
    const uint32_t begin = blockIdx.x * blockDim.x + threadIdx.x * block, end = begin + block;
    const uint32_t k = blockIdx.y * blockDim.y + threadIdx.y;
    if (k < emb)
        for (uint32_t j = begin; j < end && j < dim_size; j++)
            out[at(ol, i, j, k)] = w[at(wl, (uint32_t)in[at(il, i, j)], k)];
}
extern "C" __global__ void
embedding_bfloat(layout3 ol, bf16_t* out, layout2 il, const int32_t* in, layout2 wl,
                 const bf16_t* w, uint32_t block)
{
    embedding_body(ol, out, il, in, wl, w, block);
}
extern "C" __global__ void
embedding_float(layout3 ol, float* out, layout2 il, const int32_t* in, layout2 wl, const float* w,
                uint32_t block)
{
    embedding_body(ol, out, il, in, wl, w, block);
}

// roll (kernel/roll.metal:23-49): out[k] = in[base + ((k/stride + shift) % size)*stride + k%stride]
template <typename S>
__device__ __forceinline__ void
roll_body(const layout1& ol, S* out, const layout1& il, const S* in, uint32_t shift, uint32_t size,
          uint32_t stride)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride_size = size * stride;
    const uint32_t base = (k / stride_size) * stride_size;
    const uint32_t i = (k / stride + shift) % size;
    const uint32_t j = k % stride;
    const uint32_t m = base + i * stride + j;
    if (k < il.sizes[0]) out[at(ol, k)] = in[at(il, m)];
}
extern "C" __global__ void
roll_bfloat(layout1 ol, bf16_t* out, layout1 il, const bf16_t* in, uint32_t shift, uint32_t size,
            uint32_t stride)
{
    roll_body(ol, out, il, in, shift, size, stride);
}
extern "C" __global__ void
roll_float(layout1 ol, float* out, layout1 il, const float* in, uint32_t shift, uint32_t size,
           uint32_t stride)
{
    roll_body(ol, out, il, in, shift, size, stride);
}
