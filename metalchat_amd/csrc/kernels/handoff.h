// In-launch hand-offs (MI355X_MICROARCH.md / cdna_hip_programming.md Guideline 16, form R2) and the step state they are tagged by.
// Every shared word is ONE naturally aligned 8-byte {value, tag} granule written by one agent-scope (sc1) store and read by
// agent-scope loads until its tag is this launch's: the data is the flag, no fence, nothing depends on dispatch order or on which
// XCD a workgroup lands.  tag = step epoch * 256 + a number that is unique inside the token (layer + 1, or 128 + layer for the
// second hand-off buffer of a layer): the epoch counts steps since the decoder was created and is never reset, so a granule of
// an earlier launch is never mistaken for this one's and nothing has to be cleared between launches.  Every wait is bounded
// (50 ms of s_memrealtime -- a launch is ~ 15 us, so 3000 x its length; round 4: 2 s): on expiry the launch sets state.err and carries on with what it has; later waits see the flag and do
// not wait at all, and the host reports it (mc_decoder_generate / _step return MC_ERR_RUNTIME).  A launch whose workgroups wait
// for one another must be co-resident: the host guarantees it (decoder.cc attn_fused(), attn_wo_fused(), chain_ok()).
#pragma once

#include "common.h"

struct step_state {
    int32_t token;      // input token of the current step
    int32_t pos;        // start_pos of the current step
    int32_t kv_len;     // valid cache slots after this step's write  = min(pos + 1, max_seq)
    int32_t write_slot; // physical slot of this step's K/V row
    int32_t ring_base;  // rotation of the post-sink ring
    int32_t step_index; // index into tokens_out for chained generation
    int32_t rope_row;   // pos - rope_table_start
    int32_t rolled;     // number of rolls so far (debug)
    int32_t rope_start; // first position of the rope table window (nn/embedding.h:190-198); moved by mc_step_rope
    uint32_t epoch;     // counts the steps since the decoder was created (never reset): the tag of in-launch hand-offs
    uint32_t err;       // set by a kernel whose in-launch hand-off gave up (mc_attn_fused_T); 0 = none
    int32_t pad[1];
};

typedef __attribute__((address_space(1))) unsigned long long gu64_t;
typedef __attribute__((address_space(1))) uint32_t gu32_t;

__device__ __forceinline__ void
granule_store(unsigned long long* g, uint32_t tag, uint32_t value)
{
    __hip_atomic_store((gu64_t*)g, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long
granule_load(const unsigned long long* g)
{
    return __hip_atomic_load((gu64_t*)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ---- the XCD-local fast path of a hand-off (round 4; tools/handoff_lab.hip: 0.72 us per all-to-all round of 32 workgroups on one
// XCD against 1.24 through the fabric on an idle chip, 1.8 against 3.8 beside streaming waves).  A granule published TWICE: a
// PLAIN 8-byte store to its `fast` word -- it stays in the producer XCD's L2, where an agent-scope (sc1: L1-bypassing, L2-served)
// load of a consumer ON THE SAME XCD finds it one L2 round trip later -- and the guide's agent-scope store to its `slow` word,
// which every XCD sees.  A consumer looks at `fast` and, every fourth look, at `slow` for the granules still missing: whichever
// carries this launch's tag first is the value (one aligned 8-byte store each: untorn, and a tag is unique per launch, so a stale
// or foreign line can only read as "not yet").  Placement decides the SPEED only -- workgroups with equal blockIdx.x % 8 share an
// XCD in practice, and the hand-offs A and B of the decode attention stay inside one kv head = one such class -- never the result:
// a consumer on another XCD is served by `slow` exactly as before.  `fast` sits `fast_off` granules behind `slow`.
#ifndef MC_HANDOFF_SLOW_EVERY
#define MC_HANDOFF_SLOW_EVERY 16 // a power of two: every so many looks go to the fabric copy
#endif
__device__ __forceinline__ bool
handoff_slow_look(uint32_t look)
{
    return (look & (MC_HANDOFF_SLOW_EVERY - 1u)) == MC_HANDOFF_SLOW_EVERY - 1u;
}
__device__ __forceinline__ void
granule_store_plain(unsigned long long* g, uint32_t tag, uint32_t value)
{
    const unsigned long long v = ((unsigned long long)tag << 32) | value;
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"((gu64_t*)g), "v"(v) : "memory");
}
__device__ __forceinline__ void
granule_store_dual(unsigned long long* slow, size_t fast_off, uint32_t tag, uint32_t value)
{
    granule_store_plain(slow + fast_off, tag, value);
    granule_store(slow, tag, value);
}
// one look at a granule: `look` counts this wait's looks (wave-uniform); have = a value with this launch's tag is already in `g`
__device__ __forceinline__ unsigned long long
granule_look_dual(const unsigned long long* slow, size_t fast_off, uint32_t look)
{
    return granule_load(handoff_slow_look(look) ? slow : slow + fast_off);
}

// one round of a bounded wait: false = keep waiting.  `ok` is wave-uniform.
#ifndef MC_HANDOFF_BOUND_TICKS
#define MC_HANDOFF_BOUND_TICKS 5000000ull // 50 ms at 100 MHz
#endif
struct handoff_wait {
    unsigned long long t0;
    uint32_t spins;
    __device__ __forceinline__ handoff_wait() : t0(__builtin_amdgcn_s_memrealtime()), spins(0) {}
    // true: give up (this launch or an earlier one of the token ran out of time)
    __device__ __forceinline__ bool
    expired(step_state* st, uint32_t code)
    {
#ifndef MC_HANDOFF_SLEEP
#define MC_HANDOFF_SLEEP 4 // x 64 cycles between two looks of a waiting wave (tuning builds: tools/experiments/README.md)
#endif
        __builtin_amdgcn_s_sleep(MC_HANDOFF_SLEEP);
        if ((++spins & 63u) != 0) return false;
        if (__hip_atomic_load((gu32_t*)&st->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > MC_HANDOFF_BOUND_TICKS) {
            __hip_atomic_store((gu32_t*)&st->err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        return false;
    }
};
