#!/usr/bin/env python3
"""Where a launch of mc_attn_qkv_wo_w13_w_bfloat_hd64_k4_q4_f{3p3,4p4} (round 6: the plain-weight attention block + ffn_norm + w1|w3 + SiLU * mul in one
launch) spends its time, next to the two launches it replaces on the same buffers.  TinyLlama-1.1B (MODEL=tinyllama: 4 kv heads as 8 virtual ones,
ffn 5632) or Llama-3.2-1B (MODEL=llama32: 8 kv heads, ffn 8192) shapes, a full cache of 2048 slots, the matrices of a synthetic decoder of L layers,
launches back to back with consecutive layer tags; s_memrealtime stamps (100 MHz) of every workgroup:
  block (tl, thread 0):   0 start  1 row staged  2 wq|wk|wv pairs published  11 hand-off Q  4 scores  5 hand-off A  6 P.V  7 partials published
                          8 hand-off B + reduce  9 hand-off C, attention row staged  10 Wo pairs stored
  chain (tl2):            40 Wo done (thread 0)  41 hand-off D: hidden row gathered (thread 0)  42 row staged  44 wave 0 (poller) stored  45 wave 4 (fetcher) stored
usage: [MODEL=tinyllama|llama32] attn_w13_timeline.py [launches=32] [layers=8]     (MC_HSACO: a tuning build of the code object)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import metalchat_amd as mc
import modelgen as mg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8
TINY = os.environ.get("MODEL", "tinyllama") == "tinyllama"
H, KVR, hd, S, dim, ffn, F = (32, 4, 64, 2048, 2048, 5632, "3p3") if TINY else (32, 8, 64, 2048, 2048, 8192, "4p4")
VSH = 1 if TINY else 0                      # virtual kv heads (decoder.cc kv_virtual_shift)
KV, n_rep, nsplit = KVR << VSH, (H // KVR) >> VSH, S // 64
WGS = nsplit * KV
acc = mc.HardwareAccelerator(path=os.environ.get("MC_HSACO"))
cfg = dict(dtype=0, n_layers=L, vocab=2048, norm_eps=1e-5, max_seq_len=S, family=0, dim=dim, n_heads=H, n_kv_heads=KVR, head_dim=hd, ffn_dim=ffn,
           rope_theta=10000.0, attn_scale=hd ** -0.5)
dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_T, group_size=0))
dec.init_synthetic(7)
wo = [dec.weight_ptrs(l, "wo") for l in range(L)]
qkv = [dec.weight_ptrs(l, "qkv") for l in range(L)]
w13 = [dec.weight_ptrs(l, "w13") for l in range(L)]
rng = np.random.default_rng(0)
bf = lambda a: (np.asarray(a, np.float32).view(np.uint32) >> 16).astype(np.uint16)
caches = [(acc.to_device(bf(rng.normal(0, 0.4, KVR * S * hd))), acc.to_device(bf(rng.normal(0, 0.5, KVR * hd * S)))) for _ in range(N)]
attn_out = acc.alloc(H * hd * 2)
hidden0 = bf(rng.normal(0, 1, dim))
hidden = acc.to_device(hidden0)
norm_w = acc.to_device(bf(rng.uniform(0.5, 1.5, dim)))
fcos = acc.to_device(np.cos(rng.uniform(0, 6, (4, hd // 2))).astype(np.float32))
fsin = acc.to_device(np.sin(rng.uniform(0, 6, (4, hd // 2))).astype(np.float32))
psum = acc.to_device(np.zeros(2 * H * nsplit, np.uint64))
slab = acc.to_device(np.zeros(2 * H * hd * nsplit, np.uint64))
row_g = acc.to_device(np.zeros(H * hd // 2, np.uint64))
hid_g = acc.to_device(np.zeros(dim // 2, np.uint64))
qkv_g = acc.to_device(np.zeros(2 * KV * (n_rep + 2) * hd // 2, np.uint64))
gate = acc.alloc(ffn * 2)
TL2 = 48
tl = acc.alloc(N * WGS * 16 * 8)
tl2 = acc.alloc(N * WGS * TL2 * 8)
k_chain = acc.load(f"mc_attn_qkv_wo_w13_w_bfloat_hd64_k4_q4_f{F}")
k_block = acc.load("mc_attn_qkv_wo_w_bfloat_hd64_k4_q4")
k_w13 = acc.load("mc_gemv_w_bfloat_ling4_p1_e2")
W = lambda p: acc.wrap(p, 1 << 40)
LDS13 = dim * 2 + 128 + 8 * 512
FAST = int(os.environ.get("MC_HANDOFF_FAST", "1"))


def block_args(i, tlp):
    kc, vt = caches[i]
    l = i % L
    return [kc, vt, attn_out, psum, slab, row_g, qkv_g, state, np.uint32(n_rep), np.uint32(KV), np.uint32(S), np.float32(hd ** -0.5),
            np.uint32(nsplit), np.uint32(i + 1), W(wo[l][0]), None, hidden, hidden, np.uint32(dim), np.uint32(0),
            norm_w, W(qkv[l][0]), None, fcos, fsin, np.float32(1e-5), np.float32(0.0), np.uint32(FAST), tlp, np.uint32(VSH)]


def run(form, stamps):
    hidden.upload(hidden0)
    acc.timer_begin()
    for i in range(N):
        l = i % L
        tlp = acc.wrap(tl.device_ptr + i * WGS * 128, WGS * 128) if stamps else None
        if form == "chain":
            t2 = acc.wrap(tl2.device_ptr + i * WGS * TL2 * 8, WGS * TL2 * 8) if stamps else None
            mc.KernelTask(k_chain, (WGS * 512, 1, 1), (512, 1, 1), block_args(i, tlp) + [hid_g, W(w13[l][0]), norm_w, gate, np.uint32(2 * ffn), t2])()
        else:
            mc.KernelTask(k_block, (WGS * 512, 1, 1), (512, 1, 1), block_args(i, tlp))()
            mc.KernelTask(k_w13, (256 * 512, 1, 1), (512, 1, 1),
                          [W(w13[l][0]), None, hidden, gate, None, norm_w, np.uint32(2 * ffn), np.uint32(dim), np.uint32(0), np.float32(1e-5),
                           np.float32(0.0), None, None, np.uint32(0), np.float32(0)], lds_bytes=LDS13)()
    ms = acc.timer_end_ms()
    acc.wait()
    return ms * 1e3 / N


results = {}
for epoch in range(1, 5):
    st = np.zeros(12, np.int32)
    st[2], st[3], st[6], st[9] = S, S - 1, 1, epoch   # kv_len, write_slot, rope_row, epoch
    state = acc.to_device(st)
    form = "chain" if epoch % 2 else "two"
    stamps = epoch >= 3
    if stamps:
        tl.upload(np.zeros(N * WGS * 16, np.uint64))
        if form == "chain":
            tl2.upload(np.zeros(N * WGS * TL2, np.uint64))
    us = run(form, stamps)
    if stamps and form == "chain":
        t = tl.download(np.uint64, N * WGS * 16).reshape(N, WGS, 16).astype(np.int64)
        t2 = tl2.download(np.uint64, N * WGS * TL2).reshape(N, WGS, TL2).astype(np.int64)
    if stamps and form == "two":
        tb = tl.download(np.uint64, N * WGS * 16).reshape(N, WGS, 16).astype(np.int64)
    g = gate.download(np.uint16, ffn)
    h = hidden.download(np.uint16, dim)
    err = int(state.download(np.int32, 12)[10])
    results.setdefault(form, []).append((us, g.copy(), h.copy()))
    print(f"epoch {epoch}: {form:5s} {N} launches (eager, host-bound: not a timing{', stamped' if stamps else ''}), err word {err:#x}", flush=True)
ga, ha = results["chain"][0][1:]
gb, hb = results["two"][0][1:]
print(f"identity after {N} chained steps on one hidden row: gate row {'EQUAL' if np.array_equal(ga, gb) else 'DIFFERENT'}, hidden row {'EQUAL' if np.array_equal(ha, hb) else 'DIFFERENT'}")

order = [0, 1, 2, 11, 4, 5, 6, 7, 8, 9, 10]
names = ["start -> row staged", "wq|wk|wv pairs + rope + publish", "hand-off Q", "scores + exp", "hand-off A (denominators)", "P.V",
         "publish partial rows", "hand-off B + reduce", "hand-off C + staging", "Wo pairs + store"]
print("\n-- the attention block's phases, medians over the workgroups (us): inside the chained launch | as a launch of its own")
for i in (1, N // 2, N - 1):
    row = []
    for p in range(len(names)):
        da = (t[i, :, order[p + 1]] - t[i, :, order[p]]) / 100.0
        db = (tb[i, :, order[p + 1]] - tb[i, :, order[p]]) / 100.0
        row.append(f"{names[p]} {np.median(da):.2f}|{np.median(db):.2f}")
    ea = (t[i, :, 10].max() - t[i, :, 0].min()) / 100.0
    eb = (tb[i, :, 10].max() - tb[i, :, 0].min()) / 100.0
    print(f" launch {i}: " + "; ".join(row) + f"; last Wo store after the first start {ea:.2f}|{eb:.2f}")
print("\n-- the chained launch: microseconds after the first workgroup's start")
for i in (1, N // 2, N - 1):
    s0 = t[i, :, 0].min()
    def col(a, j):
        d = (a[i, :, j] - s0) / 100.0
        return f"min {d.min():6.2f} median {np.median(d):6.2f} max {d.max():6.2f}"
    print(f" launch {i}:")
    print(f"   hand-off C done, attention row staged  {col(t, 9)}")
    print(f"   Wo pairs stored (thread 0)             {col(t2, 40)}")
    print(f"   hand-off D: row gathered (thread 0)    {col(t2, 41)}")
    print(f"   row normalised and staged              {col(t2, 42)}")
    print(f"   wave 0 (poller) stored                 {col(t2, 44)}")
    print(f"   wave 4 (fetcher) stored                {col(t2, 45)}")
dec.release()
