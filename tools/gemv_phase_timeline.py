#!/usr/bin/env python3
"""Where a launch of a linear-order GEMV spends its time (tuning aid; needs a TUNING build whose kernels stamp s_memrealtime
into the buffer passed as `lora_a` -- tools/experiments/README.md "gemv phase timeline"): the kernel launched by name on the
matrices of a two-block Llama-3-8B-width decoder, `n` launches back to back, per-wave stamps (100 MHz):
  0 wave start   1 row + first ring tiles requested   2 own sum of squares done (the wave's row packets have arrived)
  3 behind the rmsnorm barrier   4 row staged (second barrier)   5 main loop done   6 epilogue done
usage: MC_HSACO=<tuning build> gemv_phase_timeline.py [w13|qkv|w2|wo] [launches=16]"""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import metalchat_amd as mc
import modelgen as mg

WHICH = sys.argv[1] if len(sys.argv) > 1 else "w13"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
acc = mc.HardwareAccelerator()
dim, ffn, H, KV, hd, S = 4096, 14336, 32, 8, 128, 2048
cfg = dict(dtype=0, n_layers=2, vocab=2048, norm_eps=1e-5, max_seq_len=S, family=0, dim=dim, n_heads=H, n_kv_heads=KV, head_dim=hd, ffn_dim=ffn,
           rope_theta=500000.0, attn_scale=hd ** -0.5)
dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=128))
dec.init_synthetic(7)
rng = np.random.default_rng(0)
bf = lambda a: (np.asarray(a, np.float32).view(np.uint32) >> 16).astype(np.uint16)
kname = {"w13": "mc_gemv_i4_bfloat_lin2_p1_e2", "qkv": "mc_gemv_i4_bfloat_lin2_p1_e0", "w2": "mc_gemv_i4_bfloat_lin7_p0_e1",
         "wo": "mc_gemv_i4_bfloat_lin2_p0_e1"}[WHICH]
CUS = 256
W = lambda p: acc.wrap(p, 1 << 40)
ptrs = [dec.weight_ptrs(l, WHICH) for l in (0, 1)]
rows, K = ptrs[0][2], ptrs[0][3]
x = acc.to_device(bf(rng.normal(0, 1, K)))
norm = acc.to_device(bf(rng.uniform(0.5, 1.5, K)))
y = acc.alloc(rows * 2)
res = acc.to_device(bf(rng.normal(0, 1, rows)))
lds = K * 2 // 16 * 17 + 128 + 8 * 512
tl = acc.alloc(N * CUS * 8 * 8 * 8)
tl.upload(np.zeros(N * CUS * 8 * 8, np.uint64))
k = acc.load(kname)
for rep in range(2):
    acc.timer_begin()
    for i in range(N):
        wp, sp = ptrs[i & 1][0], ptrs[i & 1][1]
        mc.KernelTask(k, (CUS * 512, 1, 1), (512, 1, 1),
                      [W(wp), W(sp), x, y, res, norm, np.uint32(rows), np.uint32(K), np.uint32(128), np.float32(1e-5), np.float32(0.0),
                       (acc.wrap(tl.device_ptr + i * CUS * 8 * 64, CUS * 8 * 64) if rep else None), None, np.uint32(0), np.float32(0)],
                      lds_bytes=lds)()
    ms = acc.timer_end_ms()
    acc.wait()
    print(f"{kname}: {ms * 1e3 / N:.2f} us per launch (eager, {'stamped' if rep else 'plain'})")
t = tl.download(np.uint64, N * CUS * 8 * 8).reshape(N, CUS * 8, 8).astype(np.int64)
names = ["start -> requests out", "-> own packets in, squares summed", "-> behind the rmsnorm barrier", "-> row staged", "main loop", "epilogue"]
for i in (1, N // 2, N - 1):
    s0 = t[i, :, 0].min()
    print(f" launch {i}: wave starts spread {(t[i, :, 0].max() - s0) / 100:.2f} us; last wave ends {(t[i, :, 6].max() - s0) / 100:.2f} us after the first start")
    for p, nm in enumerate(names):
        if (t[i, :, p + 1] == 0).all() or (t[i, :, p] == 0).all():
            continue
        d = (t[i, :, p + 1] - t[i, :, p]) / 100.0
        print(f"   {nm:36s} median {np.median(d):5.2f}  p10 {np.percentile(d, 10):5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():5.2f} us")
    for p in range(1, 7):
        if (t[i, :, p] == 0).all():
            continue
        d = (t[i, :, p] - s0) / 100.0
        print(f"   stamp {p} after first start: min {d.min():5.2f} median {np.median(d):5.2f} max {d.max():5.2f} us")
# who is slow?  loop time by XCD (workgroup b runs on XCD b % 8 in practice), by wave of the workgroup, and inside a workgroup
i = N // 2
loop = ((t[i, :, 5] - t[i, :, 4]) / 100.0).reshape(CUS, 8)
start = ((t[i, :, 0] - t[i, :, 0].min()) / 100.0).reshape(CUS, 8)
end = ((t[i, :, 6] - t[i, :, 0].min()) / 100.0).reshape(CUS, 8)
print(f" launch {i}: main loop per XCD (median / max over its 32 workgroups x 8 waves), start of the XCD's first wave, end of its last:")
for x_ in range(8):
    sel = np.arange(CUS) % 8 == x_
    print(f"   XCD {x_}: loop median {np.median(loop[sel]):5.2f} max {loop[sel].max():5.2f}   first start {start[sel].min():5.2f}   last end {end[sel].max():5.2f}")
print("   by wave of the workgroup (median loop):", " ".join(f"{np.median(loop[:, w_]):5.2f}" for w_ in range(8)))
wg_max, wg_min = loop.max(1), loop.min(1)
print(f"   inside a workgroup: slowest - fastest wave median {np.median(wg_max - wg_min):.2f} us (max {np.max(wg_max - wg_min):.2f}); "
      f"slowest wave of a workgroup: min {wg_max.min():.2f} median {np.median(wg_max):.2f} max {wg_max.max():.2f} over the 256 workgroups")
dec.release()
