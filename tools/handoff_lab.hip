// Tuning aid (not the product path): what does a hand-off between the workgroups of ONE XCD cost when the granule stays in that
// XCD's L2 (plain store, agent-scope load) instead of going through the fabric (agent-scope store, agent-scope load:
// cdna_hip_programming.md Guideline 16 form R2, what decode_kernels.hip does today)?
//
// 256 workgroups, one per CU; workgroup b belongs to group b % 8 (blocks b and b + 8 share an XCD in practice -- for speed only)
// and is member b / 8 of it.  A round = every member publishes one {value, tag} granule and gathers the 32 granules of its group;
// the value it publishes next depends on what it gathered, so R rounds are R dependent all-to-all hand-offs.
//   mode 0: sc1 store, sc1 loads                       (today's granule_store / granule_load)
//   mode 1: plain store, sc1 loads                     (L2-local only: wrong placement = the bounded wait gives up)
//   mode 2: plain store to FAST + sc1 store to SLOW; the consumer polls FAST and looks at SLOW every 8th look
//           (placement-independent: a member on another XCD is found through SLOW)
//   mode 3: as mode 2, but FAST is polled with SCALAR loads (s_load_dwordx16 glc: the scalar cache bypassed, served by the XCD's L2) --
//           they do not queue in the CU's vector-memory pipe behind the tiles other waves stream
//   load != 0: waves 1-3 of every workgroup stream a private buffer with non-temporal loads meanwhile (a busy CU)
//   hipcc --offload-arch=gfx950 -O3 tools/handoff_lab.hip -o tools/handoff_lab && tools/handoff_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned long long gu64_t;

__device__ __forceinline__ void
store_sc1(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store((gu64_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void
store_plain(unsigned long long* p, unsigned long long v)
{
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned long long
load_sc1(const unsigned long long* p)
{
    return __hip_atomic_load((gu64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
// 32 granules = 256 bytes at `p` (wave-uniform) into 64 SGPRs
__device__ __forceinline__ void
sload_row(const unsigned long long* p, u32x16& a, u32x16& b, u32x16& c, u32x16& d)
{
    asm volatile("s_load_dwordx16 %0, %4, 0x0 glc\n\ts_load_dwordx16 %1, %4, 0x40 glc\n\ts_load_dwordx16 %2, %4, 0x80 glc\n\t"
                 "s_load_dwordx16 %3, %4, 0xc0 glc\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d)
                 : "s"(p)
                 : "memory");
}

// fast / slow: [4 buffers][8 groups][32 members] granules; out: [256] final values; xcc: [256]; stamps: [256][rounds + 1]
extern "C" __global__ void __launch_bounds__(256)
k_rounds(unsigned long long* fast, unsigned long long* slow, uint32_t* out, uint32_t* xcc, unsigned long long* stamps, uint32_t* gaveup,
         const uint4* stream, uint32_t stream_pk, uint32_t rounds, uint32_t mode, uint32_t load, uint32_t epoch, uint32_t sleep)
{
    __shared__ uint32_t done;
    const uint32_t b = blockIdx.x, group = b % 8, member = b / 8, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        done = 0;
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[b] = id & 0xf;
    }
    __syncthreads();
    if (wave != 0) {
        if (!load) return;
        // a busy CU: stream until wave 0 is done
        const uint4* base = stream + (size_t)b * stream_pk;
        uint32_t acc = 0, p = threadIdx.x - 64;
        while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (p % stream_pk)));
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
                p += 192;
            }
        }
        if (acc == 0x12345678u) out[b] = acc;
        return;
    }
    uint32_t value = b + 1;
    if (lane == 0) stamps[(size_t)b * (rounds + 1)] = __builtin_amdgcn_s_memrealtime();
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t tag = epoch * 65536u + r + 1;
        const size_t slot = ((size_t)(r & 3) * 8 + group) * 32;
        const unsigned long long g = ((unsigned long long)tag << 32) | value;
        if (lane == 0) {
            if (mode == 0) store_sc1(fast + slot + member, g);
            if (mode == 1) store_plain(fast + slot + member, g);
            if (mode >= 2) {
                store_plain(fast + slot + member, g);
                store_sc1(slow + slot + member, g);
            }
        }
        uint32_t sum = 0;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (uint32_t look = 0; mode == 3; look++) {
            u32x16 q[4];
            sload_row(fast + slot, q[0], q[1], q[2], q[3]);
            uint32_t bad = 0, tot = 0;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    bad |= q[i][2 * j + 1] ^ tag;
                    tot += q[i][2 * j];
                }
            if (bad == 0) {
                sum = lane == 0 ? tot : 0u;
                break;
            }
            if ((look & 7u) == 7u) { // the fabric copy, with vector loads as mode 2
                const unsigned long long v = lane < 32 ? load_sc1(slow + slot + lane) : ((unsigned long long)tag << 32);
                if (__all((uint32_t)(v >> 32) == tag)) {
                    sum = lane < 32 ? (uint32_t)v : 0u;
                    break;
                }
            }
            if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) {
                if (lane == 0) atomicAdd(gaveup, 1u);
                sum = 0;
                break;
            }
            if (sleep) __builtin_amdgcn_s_sleep(4);
        }
        for (uint32_t look = 0; mode != 3; look++) {
            unsigned long long v = lane < 32 ? load_sc1(fast + slot + lane) : ((unsigned long long)tag << 32);
            bool ok = (uint32_t)(v >> 32) == tag;
            if (mode == 2 && (look & 7u) == 7u && !__all(ok)) {
                if (!ok) v = load_sc1(slow + slot + lane);
                ok = (uint32_t)(v >> 32) == tag;
            }
            if (__all(ok)) {
                sum = lane < 32 ? (uint32_t)v : 0u;
                break;
            }
            if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) { // 50 ms: give up
                if (lane == 0) atomicAdd(gaveup, 1u);
                sum = 0;
                break;
            }
            if (sleep) __builtin_amdgcn_s_sleep(4);
        }
        for (int off = 16; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
        value = sum * 2654435761u + b;
        if (lane == 0) stamps[(size_t)b * (rounds + 1) + r + 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (lane == 0) {
        out[b] = value;
        __hip_atomic_store(&done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

int
main()
{
    const uint32_t rounds = 64, stream_pk = 65536; // 1 MiB per workgroup
    unsigned long long *fast, *slow, *stamps;
    uint32_t *out, *xcc, *gaveup;
    uint4* stream;
    CK(hipMalloc(&fast, 4 * 8 * 32 * 8));
    CK(hipMalloc(&slow, 4 * 8 * 32 * 8));
    CK(hipMalloc(&stamps, (size_t)256 * (rounds + 1) * 8));
    CK(hipMalloc(&out, 256 * 4));
    CK(hipMalloc(&xcc, 256 * 4));
    CK(hipMalloc(&gaveup, 4));
    CK(hipMalloc(&stream, (size_t)256 * stream_pk * 16));
    CK(hipMemset(fast, 0, 4 * 8 * 32 * 8));
    CK(hipMemset(slow, 0, 4 * 8 * 32 * 8));
    CK(hipMemset(stream, 1, (size_t)256 * stream_pk * 16));
    CK(hipMemset(gaveup, 0, 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    uint32_t epoch = 1;
    std::vector<uint32_t> ref_out;
    for (uint32_t load = 0; load < 2; load++)
        for (uint32_t sleep = 0; sleep < 2; sleep++)
            for (uint32_t mode = 0; mode < 4; mode++) {
                std::vector<double> per_round;
                uint32_t gave = 0;
                std::vector<uint32_t> h_out(256), h_xcc(256);
                for (int rep = 0; rep < 5; rep++, epoch++) {
                    hipLaunchKernelGGL(k_rounds, dim3(256), dim3(256), 0, s, fast, slow, out, xcc, stamps, gaveup, stream, stream_pk, rounds, mode, load,
                                       epoch, sleep);
                    CK(hipStreamSynchronize(s));
                    std::vector<unsigned long long> h((size_t)256 * (rounds + 1));
                    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(h_out.data(), out, 256 * 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(h_xcc.data(), xcc, 256 * 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(&gave, gaveup, 4, hipMemcpyDeviceToHost));
                    if (rep == 0) continue; // warm-up
                    // per-round time: the median over workgroups of (stamp[r + 1] - stamp[r]) over rounds 8 ..
                    for (uint32_t r = 8; r < rounds; r++) {
                        std::vector<double> d(256);
                        for (int b = 0; b < 256; b++) d[b] = (double)(h[(size_t)b * (rounds + 1) + r + 1] - h[(size_t)b * (rounds + 1) + r]) * 0.01;
                        std::sort(d.begin(), d.end());
                        per_round.push_back(d[128]);
                    }
                }
                std::sort(per_round.begin(), per_round.end());
                // do the groups sit on one XCD each?
                int mixed = 0;
                for (int g = 0; g < 8; g++)
                    for (int m = 1; m < 32; m++) mixed += h_xcc[g + 8 * m] != h_xcc[g];
                if (mode == 0 && load == 0 && sleep == 0) ref_out = h_out;
                const bool same = ref_out == h_out;
                printf("load %u sleep %u mode %u: per round median %.2f us  p10 %.2f  p90 %.2f   gave up %u  members off their group's XCD %d  values %s\n", load,
                       sleep, mode, per_round[per_round.size() / 2], per_round[per_round.size() / 10], per_round[per_round.size() * 9 / 10], gave, mixed,
                       same ? "== mode 0" : "DIFFER");
            }
    return 0;
}
