"""BASELINE.json's own sizes (Llama-3-8B int4 g128, bf16, vocab 128256) on the GPU.

The oracle cannot run a whole 8B step in seconds, so parity at full size is checked (a) on SAMPLED
ROWS of the dominant fused GEMVs -- the host regenerates those rows of the synthetic weights
(mc_synth_*), the oracle computes rmsnorm -> hadamard_broadcast -> bmm -> SiLU*mul / residual for
them, and the kernel's outputs at those rows must match -- and (b) through size-independent
properties of the whole 32-layer decoder: graph replay == eager chain == step by step, two decoders
agree token for token, the sink ring keeps exactly max_seq_len rows, the device sampler equals the
oracle's chain on the decoder's own 128256 logits."""
import numpy as np
import pytest

import parity
from oracle import mc_oracle as mo
from test_gemv_gpu import run_gemv

pytestmark = pytest.mark.gpu
BF16 = 0
M = dict(dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, vocab=128256, rope_theta=500000.0,
         norm_eps=1e-5)
SEED = 0x5EED


def make(acc, n_layers, max_seq_len=64, **over):
    import metalchat_amd as mc

    dec = mc.Decoder(acc, dtype=mc.BF16, n_layers=n_layers, max_seq_len=max_seq_len, attn_scale=128 ** -0.5,
                     weight_format=mc.WFMT_I4, group_size=128, **M, **over)
    dec.init_synthetic(SEED)
    return dec


def synth_rows(mid, rows, in_f, group=128):
    """(q int8 [n, in], scales f32 [n, in/group]) of the given rows of synthetic matrix `mid`."""
    import metalchat_amd as mc

    lib = mc.capi()
    q = np.array([[lib.mc_synth_weight(SEED, mid, int(r), c, 4) for c in range(in_f)] for r in rows], np.int8)
    s = np.array([[lib.mc_synth_scale(SEED, mid, int(r), g, in_f, 4) for g in range(in_f // group)] for r in rows],
                 np.float32)
    return q, s


def oracle_rows(q, s, x_T, group=128):
    """T(x Wd^T) for the sampled rows: hadamard_broadcast + bmm (quantization/lora.h:105-117)."""
    n, in_f = q.shape
    ng = in_f // group
    L = mo.layout
    wd = np.zeros((n, in_f), np.uint16)
    mo.hadamard_broadcast(BF16, 1, L((n * ng, group)), wd, L((n * ng, group)), q, L((n * ng,)), np.ascontiguousarray(s.reshape(-1)))
    y = np.zeros((1, 1, n), np.uint16)
    mo.bmm(BF16, L(y.shape), y, L((1, 1, in_f)), x_T, L((1, in_f, n), strides=(in_f * n, 1, in_f)), wd)
    return y.reshape(-1)


def norm_weights(mid, n):
    import metalchat_amd as mc

    lib = mc.capi()
    return mo.encode(BF16, np.array([lib.mc_synth_value(SEED, mid, i, 0, 1) for i in range(n)], np.float32))


def test_dominant_gemvs_on_sampled_rows_at_full_size(acc):
    dec = make(acc, 1)
    rng = np.random.default_rng(3)
    L = mo.layout
    dim, ffn = M["dim"], M["ffn_dim"]
    x = mo.encode(BF16, rng.normal(0, 1, dim).astype(np.float32))
    # ---- w1|w3: rmsnorm prologue + GEMV + SiLU*mul epilogue, the kernel bench.py's roofline is about
    nw = norm_weights(9, dim)                                     # layer 0 ffn_norm
    xn = np.zeros((1, dim), np.uint16)
    mo.rmsnorm(BF16, L((1, dim)), xn, L((1, dim)), x.reshape(1, -1), L((dim,)), nw, M["norm_eps"], 0.0)
    js = np.unique(np.concatenate([[0, 1, ffn - 1], rng.integers(0, ffn, 13)]))
    q1, s1 = synth_rows(4, js, dim)
    q3, s3 = synth_rows(6, js, dim)
    a, b = oracle_rows(q1, s1, xn.reshape(-1)), oracle_rows(q3, s3, xn.reshape(-1))
    act = np.zeros((1, len(js)), np.uint16)
    mo.silu(BF16, L(act.shape), act, L(act.shape), a.reshape(1, -1))
    ref = np.zeros((1, len(js)), np.uint16)
    mo.hadamard(BF16, L(ref.shape), ref, L(ref.shape), act, L(ref.shape), b.reshape(1, -1))
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w13")
    assert (rows, inf) == (2 * ffn, dim)
    got = run_gemv(acc, "mc_gemv_i4_bfloat_m4_p1_e2", wptr, sptr, x, ffn, rows, inf, 128, BF16, norm=nw, wgs=512)
    parity.check(BF16, got[js], ref.reshape(-1), rel=2e-3, max_ulp=1, max_frac=0.2, scale_aware=False, what="w1|w3 rows")
    # ---- w2 with the residual epilogue (K = 14336: seven chunks per row)
    g = mo.encode(BF16, rng.normal(0, 0.5, ffn).astype(np.float32))
    res = mo.encode(BF16, rng.normal(0, 1, dim).astype(np.float32))
    rs = np.unique(np.concatenate([[0, dim - 1], rng.integers(0, dim, 6)]))
    q2, s2 = synth_rows(5, rs, ffn)
    y2 = oracle_rows(q2, s2, g)
    ref2 = np.zeros((1, len(rs)), np.uint16)
    mo.add(BF16, L(ref2.shape), ref2, L(ref2.shape), res[rs].reshape(1, -1), L(ref2.shape), y2.reshape(1, -1))
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w2")
    got2 = run_gemv(acc, "mc_gemv_i4_bfloat_m4_p0_e1", wptr, sptr, g, dim, rows, inf, 128, BF16, res=res, wgs=256)
    parity.check(BF16, got2[rs], ref2.reshape(-1), rel=2e-3, max_ulp=1, max_frac=0.3, scale_aware=False, what="w2 rows")
    # ---- output head: 128256 rows, final norm prologue
    fw = norm_weights(0xFFFF0002, dim)
    mo.rmsnorm(BF16, L((1, dim)), xn, L((1, dim)), x.reshape(1, -1), L((dim,)), fw, M["norm_eps"], 0.0)
    vs = np.unique(np.concatenate([[0, M["vocab"] - 1], rng.integers(0, M["vocab"], 10)]))
    qh, sh = synth_rows(0xFFFF0001, vs, dim)
    refh = oracle_rows(qh, sh, xn.reshape(-1))
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(-1, "output")
    goth = run_gemv(acc, "mc_gemv_i4_bfloat_m4_p1_e0", wptr, sptr, x, M["vocab"], rows, inf, 128, BF16, norm=fw, wgs=512)
    parity.check(BF16, goth[vs], refh, rel=2e-3, max_ulp=1, max_frac=0.2, scale_aware=False, what="head rows")
    dec.release()


def test_whole_decoder_properties_at_full_size(acc):
    import metalchat_amd as mc

    n = 80                                   # max_seq_len 64: the last 16 tokens turn the sink ring
    eager = make(acc, 32, use_graph=0)
    chain = list(eager.generate(7, 0, n))
    graph = make(acc, 32, use_graph=1)
    assert list(graph.generate(7, 0, n)) == chain                       # hipGraph replay == eager launches
    k, v = graph.export_kv(31)
    assert k.shape == (64, M["n_kv_heads"], M["head_dim"])               # exactly max_seq_len logical rows
    stepper = make(acc, 32)
    tok, stepped = 7, []
    for pos in range(n):
        tok = stepper.step(tok, pos)
        stepped.append(tok)
    assert stepped == chain                                              # one call per token == chained on the device
    ks, vs = stepper.export_kv(31)
    assert np.array_equal(ks, k) and np.array_equal(vs, v)               # bit-identical caches (index work is exact)
    assert len(set(chain)) > 4                                           # not a degenerate constant stream
    # the device sampler on the decoder's own 128256 logits == the oracle's chain
    stepper.set_taps(True)
    stepper.set_sampler(mc.SAMPLER_DEFAULT, 50, 0.6, 0.9)
    stepper.set_seeds([(5, 6)])
    got = stepper.step(chain[-1], n)
    otok, otaps = mo.sample_default(BF16, stepper.logits(), init_state=5, init_seq=6, taps=True)
    assert got == otok
    parity.exact(stepper.sampler_taps(), otaps, "sampler chain at vocab 128256")
    for d in (eager, graph, stepper):
        d.release()
