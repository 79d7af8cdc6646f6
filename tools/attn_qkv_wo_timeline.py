#!/usr/bin/env python3
"""Where a launch of mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2 spends its time (tuning aid; the kernel takes a stamp buffer as its last
argument, null in the product): Llama-3-8B shapes, a full cache of 2048 slots, the wq|wk|wv and Wo matrices of a synthetic decoder,
`n` launches back to back with consecutive layer tags, s_memrealtime stamps of every workgroup (thread 0; 100 MHz):
  0 start  1 hidden row normalised and staged  2 the wave's wq|wk|wv pairs published  11 hand-off Q done (queries, K / V row in LDS)
  4 scores + exp done  5 hand-off A done  6 P.V done  7 partial rows published  8 hand-off B + reduce done
  9 hand-off C done, attention row staged  10 Wo pairs stored
usage: [MC_HANDOFF_FAST=0] [FMT=i8 [TILES=4]] attn_qkv_wo_timeline.py [launches=32]
FMT=i8: mc_attn_qkv_wo_i8_bfloat_hd128_k4_q4_t{TILES} (round 5): TILES = 1 at S = 2048, 4 at S = 8192 (256-slot ranges)
FMT=gemma [POST=1]: mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p{1 + POST}_t2 (round 5) on Gemma-7B's shapes, S = 2048"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import metalchat_amd as mc
import modelgen as mg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
FAST = int(os.environ.get("MC_HANDOFF_FAST", "1"))
I8 = os.environ.get("FMT") == "i8"
GEMMA = os.environ.get("FMT") == "gemma"
POST = int(os.environ.get("POST", "1"))
TILES = int(os.environ.get("TILES", "4")) if I8 else (2 if GEMMA else 1)
H, KV, hd, S, dim = (16, 16, 256, 2048, 3072) if GEMMA else (32, 8, 128, 2048 * TILES, 4096)
n_rep, nsplit = H // KV, S // (64 * TILES)
acc = mc.HardwareAccelerator(path=os.environ.get("MC_HSACO"))
cfg = dict(dtype=0, n_layers=2, vocab=2048, norm_eps=1e-5, max_seq_len=S, family=1 if GEMMA else 0, dim=dim, n_heads=H, n_kv_heads=KV, head_dim=hd, ffn_dim=1024,
           rope_theta=500000.0, attn_scale=hd ** -0.5)
dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I8 if I8 else mc.WFMT_I4, group_size=128))
dec.init_synthetic(7)
wo = [dec.weight_ptrs(l, "wo") for l in (0, 1)]
qkv = [dec.weight_ptrs(l, "qkv") for l in (0, 1)]
rng = np.random.default_rng(0)
bf = lambda a: (np.asarray(a, np.float32).view(np.uint32) >> 16).astype(np.uint16)
caches = [(acc.to_device(bf(rng.normal(0, 0.4, KV * S * hd))), acc.to_device(bf(rng.normal(0, 0.5, KV * hd * S)))) for _ in range(N)]
attn_out = acc.alloc(H * hd * 2)
hidden = acc.to_device(bf(rng.normal(0, 1, dim)))
norm_w = acc.to_device(bf(rng.uniform(0.5, 1.5, dim)))
fcos = acc.to_device(np.cos(rng.uniform(0, 6, (4, hd // 2))).astype(np.float32))
fsin = acc.to_device(np.sin(rng.uniform(0, 6, (4, hd // 2))).astype(np.float32))
psum = acc.to_device(np.zeros(2 * H * nsplit, np.uint64))
slab = acc.to_device(np.zeros(2 * H * hd * nsplit, np.uint64))
row_g = acc.to_device(np.zeros(H * hd // 2, np.uint64))
qkv_g = acc.to_device(np.zeros(2 * (H + 2 * KV) * hd // 2, np.uint64))
WGS = nsplit * KV
tl = acc.alloc(N * WGS * 16 * 8)
k = acc.load(f"mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p{1 + POST}_t2" if GEMMA else
             (f"mc_attn_qkv_wo_i8_bfloat_hd128_k4_q4_t{TILES}" if I8 else "mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2"))
qkn = acc.to_device(bf(rng.uniform(-0.2, 0.2, hd)))
hout = acc.alloc(dim * 2)
yout = acc.alloc(dim * 2)
W = lambda p: acc.wrap(p, 1 << 40)
order = [0, 1, 2, 11, 4, 5, 6, 7, 8, 9, 10]
names = ["start -> row staged", "wq|wk|wv pairs + rope + publish", "hand-off Q (gemma: + q/k-norm + rope)", "scores + exp", "hand-off A (denominators)", "P.V",
         "publish partial rows", "hand-off B + reduce", "hand-off C + staging", "Wo pairs + store"]
for epoch in (1, 2, 3):
    st = np.zeros(12, np.int32)
    st[2], st[3], st[6], st[9] = S, S - 1, 1, epoch   # kv_len, write_slot, rope_row, epoch
    state = acc.to_device(st)
    tl.upload(np.zeros(N * WGS * 16, np.uint64))
    acc.timer_begin()
    for i in range(N):
        kc, vt = caches[i]
        mc.KernelTask(k, (WGS * 512, 1, 1), (512, 1, 1),
                      [kc, vt, attn_out, psum, slab, row_g, qkv_g, state, np.uint32(n_rep), np.uint32(KV), np.uint32(S), np.float32(hd ** -0.5),
                       np.uint32(nsplit), np.uint32(i + 1), W(wo[i & 1][0]), W(wo[i & 1][1]), hidden, hidden, np.uint32(dim), np.uint32(128),
                       norm_w, W(qkv[i & 1][0]), W(qkv[i & 1][1]), fcos, fsin, np.float32(1e-5), np.float32(0.0), np.uint32(FAST),
                       acc.wrap(tl.device_ptr + i * WGS * 128, WGS * 128), np.uint32(0)] if not GEMMA else
                      [kc, vt, attn_out, psum, slab, row_g, qkv_g, state, np.uint32(n_rep), np.uint32(KV), np.uint32(S), np.float32(hd ** -0.5),
                       np.uint32(nsplit), np.uint32(i + 1), W(wo[i & 1][0]), W(wo[i & 1][1]), hidden, yout, np.uint32(dim), np.uint32(128),
                       norm_w, W(qkv[i & 1][0]), W(qkv[i & 1][1]), fcos, fsin, np.float32(1e-6), np.float32(1.0), np.uint32(FAST),
                       acc.wrap(tl.device_ptr + i * WGS * 128, WGS * 128), qkn, qkn, norm_w if POST else None, hidden if POST else None,
                       hout if POST else None])()
    ms = acc.timer_end_ms()
    acc.wait()
    t = tl.download(np.uint64, N * WGS * 16).reshape(N, WGS, 16).astype(np.int64)
    print(f"epoch {epoch}: {N} launches {ms * 1e3 / N:.2f} us per launch (eager), err word {int(state.download(np.int32, 12)[10]):#x}")
    if epoch < 3:
        continue
    for i in (1, N // 2, N - 1):
        s0 = t[i, :, 0].min()
        print(f" launch {i}: workgroup starts spread {(t[i, :, 0].max() - s0) / 100:.2f} us; end of the last workgroup {(t[i, :, 10].max() - s0) / 100:.2f} us after the first start")
        for p, nm in enumerate(names):
            d = (t[i, :, order[p + 1]] - t[i, :, order[p]]) / 100.0
            print(f"   {nm:32s} median {np.median(d):5.2f}  p10 {np.percentile(d, 10):5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():5.2f} us")
        for p in order[1:] + ([12, 13, 14] if os.environ.get("STAMPS") else []):
            d = (t[i, :, p] - s0) / 100.0
            print(f"   stamp {p:2d} after first start: min {d.min():5.2f} median {np.median(d):5.2f} max {d.max():5.2f} us")
dec.release()
