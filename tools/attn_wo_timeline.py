#!/usr/bin/env python3
"""Where a launch of mc_attn_wo_i4_bfloat_hd128_k2 spends its time (tuning aid; the kernel takes a stamp buffer as its last argument, null in the product): Llama-3-8B shapes, a full
cache of 2048 slots, the Wo matrix of a synthetic decoder, `n` launches back to back with consecutive layer tags, s_memrealtime
stamps of every workgroup (thread 0; 100 MHz):
  0 start  1 scores + exp done  2 hand-off A done  3 P.V done  4 partial rows published  5 hand-off B + reduce done
  6 hand-off C done, attention row staged  7 Wo pairs stored
usage: [MC_HANDOFF_FAST=0] [FMT=gemma [TILES=2]] attn_wo_timeline.py [launches=32]
FMT=gemma: mc_attn_wo_qkn_i4_bfloat_hd256_k2_t{TILES} (round 5) on Gemma-7B's shapes, S = 1024 TILES"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import metalchat_amd as mc
import modelgen as mg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
FAST = int(os.environ.get("MC_HANDOFF_FAST", "1"))
GEMMA = os.environ.get("FMT") == "gemma"
TILES = int(os.environ.get("TILES", "2")) if GEMMA else 1
H, KV, hd, S, dim = (16, 16, 256, 1024 * TILES, 3072) if GEMMA else (32, 8, 128, 2048, 4096)
n_rep, nsplit = H // KV, S // (64 * TILES)
acc = mc.HardwareAccelerator(path=os.environ.get("MC_HSACO"))
cfg = dict(dtype=0, n_layers=2, vocab=2048, norm_eps=1e-5, max_seq_len=S, family=0, dim=dim, n_heads=H, n_kv_heads=KV, head_dim=hd, ffn_dim=1024,
           rope_theta=500000.0, attn_scale=hd ** -0.5)
dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=128))
dec.init_synthetic(7)
ptrs = [dec.weight_ptrs(l, "wo") for l in (0, 1)]
rng = np.random.default_rng(0)
bf = lambda a: (np.asarray(a, np.float32).view(np.uint32) >> 16).astype(np.uint16)
q = acc.to_device(bf(rng.normal(0, 1, (H + 2 * KV) * hd)))   # (FMT=gemma: the raw wq|wk|wv rows)
nw = acc.to_device(bf(rng.uniform(-0.2, 0.2, hd)))
fcos = acc.to_device(np.cos(rng.uniform(0, 6, (4, hd // 2))).astype(np.float32))
fsin = acc.to_device(np.sin(rng.uniform(0, 6, (4, hd // 2))).astype(np.float32))
caches = [(acc.to_device(bf(rng.normal(0, 0.4, KV * S * hd))), acc.to_device(bf(rng.normal(0, 0.5, KV * hd * S)))) for _ in range(N)]
attn_out = acc.alloc(H * hd * 2)
hidden = acc.to_device(bf(rng.normal(0, 1, dim)))
psum = acc.to_device(np.zeros(2 * H * nsplit, np.uint64))
slab = acc.to_device(np.zeros(2 * H * hd * nsplit, np.uint64))
row_g = acc.to_device(np.zeros(H * hd // 2, np.uint64))
WGS = nsplit * KV
tl = acc.alloc(N * WGS * 8 * 8)
k = acc.load(f"mc_attn_wo_qkn_i4_bfloat_hd256_k2_t{TILES}" if GEMMA else "mc_attn_wo_i4_bfloat_hd128_k2")
W = lambda p: acc.wrap(p, 1 << 40)
names = ["start -> scores + exp", "hand-off A (denominators)", "P.V", "publish partial rows", "hand-off B + reduce", "hand-off C + staging", "Wo pairs + store"]
for epoch in (1, 2, 3):
    st = np.zeros(12, np.int32)
    st[2], st[3], st[6], st[9] = S, S - 1, 1, epoch
    state = acc.to_device(st)
    tl.upload(np.zeros(N * WGS * 8, np.uint64))
    acc.timer_begin()
    for i in range(N):
        kc, vt = caches[i]
        mc.KernelTask(k, (WGS * 512, 1, 1), (512, 1, 1),
                      [q, kc, vt, attn_out, psum, slab, row_g, state, np.uint32(n_rep), np.uint32(KV), np.uint32(S), np.float32(hd ** -0.5),
                       np.uint32(nsplit), np.uint32(i + 1), W(ptrs[i & 1][0]), W(ptrs[i & 1][1]), hidden, hidden, np.uint32(dim), np.uint32(128),
                       np.uint32(0 if GEMMA else 1), np.uint32(FAST), acc.wrap(tl.device_ptr + i * WGS * 64, WGS * 64)] +
                      ([nw, nw, fcos, fsin, np.float32(1e-6), np.float32(1.0)] if GEMMA else []))()
    ms = acc.timer_end_ms()
    acc.wait()
    t = tl.download(np.uint64, N * WGS * 8).reshape(N, WGS, 8).astype(np.int64)
    print(f"epoch {epoch}: {N} launches {ms * 1e3 / N:.2f} us per launch (eager), err word {int(state.download(np.int32, 12)[10]):#x}")
    if epoch < 3:
        continue
    for i in (1, N // 2, N - 1):
        s0 = t[i, :, 0].min()
        print(f" launch {i}: workgroup starts spread {(t[i, :, 0].max() - s0) / 100:.2f} us; end of the last workgroup {(t[i, :, 7].max() - s0) / 100:.2f} us after the first start")
        for p, nm in enumerate(names):
            d = (t[i, :, p + 1] - t[i, :, p]) / 100.0
            print(f"   {nm:28s} median {np.median(d):5.2f}  p10 {np.percentile(d, 10):5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():5.2f} us")
        for p in range(1, 8):
            d = (t[i, :, p] - s0) / 100.0
            print(f"   stamp {p} after first start: min {d.min():5.2f} median {np.median(d):5.2f} max {d.max():5.2f} us")
dec.release()
