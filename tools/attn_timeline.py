#!/usr/bin/env python3
"""Where a launch of mc_attn_fused_bfloat spends its time (tuning aid): the kernel launched by name with Llama-3-8B shapes at a
full cache of S slots, `n` launches back to back with consecutive layer tags (as the layers of a token), per-phase
s_memrealtime stamps of every workgroup (100 MHz), and the launch-to-launch time from the queue's events.
usage: attn_timeline.py [S=2048] [launches=32] [tiles=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import metalchat_amd as mc

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
TILES = int(sys.argv[3]) if len(sys.argv) > 3 else 1
H, KV, hd = 32, 8, 128
n_rep, nsplit = H // KV, (S + 64 * TILES - 1) // (64 * TILES)
acc = mc.HardwareAccelerator()
rng = np.random.default_rng(0)
bf = lambda a: (np.asarray(a, np.float32).view(np.uint32) >> 16).astype(np.uint16)
q = acc.to_device(bf(rng.normal(0, 1, H * hd)))
caches = [(acc.to_device(bf(rng.normal(0, 0.4, KV * S * hd))), acc.to_device(bf(rng.normal(0, 0.5, KV * hd * S)))) for _ in range(N)]
out = acc.alloc(H * hd * 2)
psum = acc.alloc(2 * H * nsplit * 8)
slab = acc.alloc(2 * H * hd * nsplit * 8)
psum.upload(np.zeros(2 * H * nsplit, np.uint64))
slab.upload(np.zeros(2 * H * hd * nsplit, np.uint64))
tl = acc.alloc(N * nsplit * KV * 8 * 8)
k = acc.load("mc_attn_fused_bfloat" if TILES == 1 else f"mc_attn_fused{TILES}_bfloat")  # (T > 1: tuning builds only)
names = ["start->scores+exp", "hand-off A (denominators)", "P.V", "publish partials", "hand-off B + reduce"]
for epoch in (1, 2, 3):
    st = np.zeros(12, np.int32)
    st[2] = S            # kv_len
    st[9] = epoch        # epoch
    state = acc.to_device(st)
    tl.upload(np.zeros(N * nsplit * KV * 8, np.uint64))
    acc.timer_begin()
    for i in range(N):
        kc, vt = caches[i]
        mc.KernelTask(k, (nsplit * KV * 256, 1, 1), (256, 1, 1),
                      [q, kc, vt, out, psum, slab, state, np.uint32(n_rep), np.uint32(KV), np.uint32(hd), np.uint32(S), np.float32(hd ** -0.5),
                       np.uint32(nsplit), np.uint32(i + 1), acc.wrap(tl.device_ptr + i * nsplit * KV * 64, nsplit * KV * 64),
                       np.uint32(int(os.environ.get("MC_HANDOFF_FAST", "1")))])()
    ms = acc.timer_end_ms()
    acc.wait()
    t = tl.download(np.uint64, N * nsplit * KV * 8).reshape(N, nsplit * KV, 8).astype(np.int64)
    err = int(state.download(np.int32, 12)[10])
    print(f"epoch {epoch}: {N} launches {ms * 1e3 / N:.2f} us per launch (eager), err word {err:#x}")
    if epoch < 3:
        continue
    for i in (1, N // 2, N - 1):
        s0 = t[i, :, 0].min()
        print(f" launch {i}: workgroup starts spread {(t[i, :, 0].max() - s0) / 100:.2f} us; end of the last workgroup {(t[i, :, 5].max() - s0) / 100:.2f} us after the first start")
        for p, nm in enumerate(names):
            d = (t[i, :, p + 1] - t[i, :, p]) / 100.0
            print(f"   {nm:28s} median {np.median(d):5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():5.2f} us")
    gaps = [(t[i + 1, :, 0].min() - t[i, :, 5].max()) / 100.0 for i in range(N - 1)]
    print(f" boundary (last end -> next first start): median {np.median(gaps):.2f} us")
