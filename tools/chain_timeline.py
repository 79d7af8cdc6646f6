#!/usr/bin/env python3
"""Where a launch of mc_w2_qkv_i4_bfloat_w7_q2 (chain_kernels.hip) spends its time (tuning aid): the kernel launched by name on the
w2 matrix of block 0 and the wq|wk|wv matrix of block 1 of a two-block Llama-3-8B-width decoder (synthetic weights), `n`
launches back to back with consecutive tags, s_memrealtime stamps of every wave (100 MHz), next to the two stand-alone launches.
usage: chain_timeline.py [launches=32]"""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import metalchat_amd as mc
import modelgen as mg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
acc = mc.HardwareAccelerator()
dim, ffn, H, KV, hd, S = 4096, 14336, 32, 8, 128, 2048
cfg = dict(dtype=0, n_layers=2, vocab=2048, norm_eps=1e-5, max_seq_len=S, family=0, dim=dim, n_heads=H, n_kv_heads=KV, head_dim=hd, ffn_dim=ffn,
           rope_theta=500000.0, attn_scale=hd ** -0.5)
dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=128))
dec.init_synthetic(7)
w2w, w2s, r2, k2, _ = dec.weight_ptrs(0, "w2")
qw, qs, rq, kq, _ = dec.weight_ptrs(1, "qkv")
assert (r2, k2, rq, kq) == (dim, ffn, (H + 2 * KV) * hd, dim)
rng = np.random.default_rng(0)
bf = lambda a: (np.asarray(a, np.float32).view(np.uint32) >> 16).astype(np.uint16)
gate = acc.to_device(bf(rng.normal(0, 0.05, ffn)))
hidden = acc.to_device(bf(rng.normal(0, 1, dim)))
norm = acc.to_device(bf(rng.uniform(0.5, 1.5, dim)))
hid_g = acc.to_device(np.zeros(dim // 2, np.uint64))
q_out = acc.alloc(H * hd * 2)
kc = acc.alloc(KV * S * hd * 2)
vt = acc.alloc(KV * hd * S * 2)
half = hd // 2
cb = acc.to_device(np.cos(rng.uniform(0, 6.28, (8, half))).astype(np.float32).reshape(-1))
sb = acc.to_device(np.sin(rng.uniform(0, 6.28, (8, half))).astype(np.float32).reshape(-1))
st = np.zeros(12, np.int32)
st[3], st[6], st[9] = 5, 3, 1
state = acc.to_device(st)
desc = acc.to_device(np.frombuffer(struct.pack("<QQQQQQIIII", q_out.device_ptr, kc.device_ptr, vt.device_ptr, cb.device_ptr, sb.device_ptr,
                                               state.device_ptr, H, KV, hd, S), np.uint8))
CUS = 256
lds_w2 = ffn * 2 // 16 * 17 + 128 + 8 * 512
lds_q = dim * 2 // 16 * 17 + 128 + 8 * 512
tl = acc.alloc(N * CUS * 8 * 8 * 8)
W = lambda p: acc.wrap(p, 1 << 40)
chain = acc.load("mc_w2_qkv_i4_bfloat_w7_q2")
k_w2 = acc.load("mc_gemv_i4_bfloat_lin7_p0_e1")
k_q = acc.load("mc_gemv_i4_bfloat_lin2_p1_e4")


def run_chain(stamps):
    for i in range(N):
        mc.KernelTask(chain, (CUS * 512, 1, 1), (512, 1, 1),
                      [W(w2w), W(w2s), gate, hidden, hidden, np.uint32(dim), np.uint32(ffn), np.uint32(128), hid_g, state, np.uint32(i + 1),
                       np.uint32(lds_w2), W(qw), W(qs), norm, desc, np.uint32(rq), np.uint32(128), np.float32(1e-5), np.float32(0.0),
                       (acc.wrap(tl.device_ptr + i * CUS * 8 * 64, CUS * 8 * 64) if stamps else None)],
                      lds_bytes=lds_w2 + 2 * 4352 + 128)()


def run_alone():
    for i in range(N):
        mc.KernelTask(k_w2, (CUS * 512, 1, 1), (512, 1, 1),
                      [W(w2w), W(w2s), gate, hidden, hidden, None, np.uint32(dim), np.uint32(ffn), np.uint32(128), np.float32(1e-5), np.float32(0.0),
                       None, None, np.uint32(0), np.float32(0)], lds_bytes=lds_w2)()
        mc.KernelTask(k_q, (CUS * 512, 1, 1), (512, 1, 1),
                      [W(qw), W(qs), hidden, q_out, desc, norm, np.uint32(rq), np.uint32(dim), np.uint32(128), np.float32(1e-5), np.float32(0.0),
                       None, None, np.uint32(0), np.float32(0)], lds_bytes=lds_q)()


for rep in range(3):
    acc.timer_begin(); run_alone(); ms_a = acc.timer_end_ms(); acc.wait()
    acc.timer_begin(); run_chain(False); ms_c = acc.timer_end_ms(); acc.wait()
    print(f"round {rep}: stand-alone w2 + wq|wk|wv {ms_a * 1e3 / N:.2f} us per pair of launches, chained {ms_c * 1e3 / N:.2f} us per launch (eager)")
tl.upload(np.zeros(N * CUS * 8 * 8, np.uint64))
run_chain(True)
acc.wait()
print("err word", hex(int(state.download(np.int32, 12)[10])))
t = tl.download(np.uint64, N * CUS * 8 * 8).reshape(N, CUS * 8, 8).astype(np.int64)
names = ["start -> last w2 tile requested", "w2 tail: reduce, residual, store, publish", "wait for the row (hand-off)", "rmsnorm + stage", "wq|wk|wv from registers + rope"]
for i in (1, N // 2, N - 1):
    s0 = t[i, :, 0].min()
    print(f" launch {i}: wave starts spread {(t[i, :, 0].max() - s0) / 100:.2f} us; last wave ends {(t[i, :, 5].max() - s0) / 100:.2f} us after the first start")
    for p, nm in enumerate(names):
        d = (t[i, :, p + 1] - t[i, :, p]) / 100.0
        print(f"   {nm:44s} median {np.median(d):5.2f}  p10 {np.percentile(d, 10):5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():5.2f} us")
    for p in range(1, 6):
        d = (t[i, :, p] - s0) / 100.0
        print(f"   stamp {p} after first start: min {d.min():5.2f} median {np.median(d):5.2f} max {d.max():5.2f} us")
dec.release()
