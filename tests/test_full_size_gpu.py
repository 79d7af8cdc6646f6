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
    # The LINEAR-ORDER kernels a token really launches (mc_gemv_i4_bfloat_lin{2,7}_*: bench.py's roofline is about _lin2_p1_e2) at the
    # decoder's own geometry -- one 512-thread workgroup per compute unit, the (8, 6) deal of 56 row pairs per workgroup, s_setprio
    # on the 7 KiB rows, the decoder's LDS size -- under the single-kernel bound, and the classic kernels of the same arithmetic
    # beside them
    from test_lin_kernels_gpu import launch as lin_launch, lds_bytes as lin_lds
    cus = acc.compute_units()
    # ---- w1|w3: rmsnorm prologue + GEMV + SiLU*mul epilogue
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
    got = run_gemv(acc, "mc_gemv_i4_bfloat_m4d_p1_e2", wptr, sptr, x, ffn, rows, inf, 128, BF16, norm=nw, wgs=512)
    parity.check(BF16, got[js], ref.reshape(-1), rel=2e-3, max_ulp=1, max_frac=0.2, scale_aware=False, what="w1|w3 rows")
    assert dec.gemv_kernel_name("w13") == "mc_gemv_i4_bfloat_lin2_p1_e2"
    gotl = lin_launch(acc, "mc_gemv_i4_bfloat_lin2_p1_e2", wptr, sptr, x, ffn, rows, inf, 128, norm=nw, wgs=cus, lds=lin_lds("i4_lin2", inf))
    parity.check(BF16, gotl[js], ref.reshape(-1), rel=2e-3, max_ulp=1, max_frac=0.2, scale_aware=False, what="w1|w3 rows, linear order, the token's grid")
    parity.exact(gotl, got, "w1|w3: every one of the 14336 outputs, linear order == classic")
    # ---- w2 with the residual epilogue (K = 14336: seven chunks per row)
    g = mo.encode(BF16, rng.normal(0, 0.5, ffn).astype(np.float32))
    res = mo.encode(BF16, rng.normal(0, 1, dim).astype(np.float32))
    rs = np.unique(np.concatenate([[0, dim - 1], rng.integers(0, dim, 6)]))
    q2, s2 = synth_rows(5, rs, ffn)
    y2 = oracle_rows(q2, s2, g)
    ref2 = np.zeros((1, len(rs)), np.uint16)
    mo.add(BF16, L(ref2.shape), ref2, L(ref2.shape), res[rs].reshape(1, -1), L(ref2.shape), y2.reshape(1, -1))
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w2")
    got2 = run_gemv(acc, "mc_gemv_i4_bfloat_m4d_p0_e1", wptr, sptr, g, dim, rows, inf, 128, BF16, res=res, wgs=256)
    parity.check(BF16, got2[rs], ref2.reshape(-1), rel=2e-3, max_ulp=1, max_frac=0.3, scale_aware=False, what="w2 rows")
    assert dec.gemv_kernel_name("w2") == "mc_gemv_i4_bfloat_lin7_p0_e1"
    got2l = lin_launch(acc, "mc_gemv_i4_bfloat_lin7_p0_e1", wptr, sptr, g, dim, rows, inf, 128, res=res, wgs=cus, lds=lin_lds("i4_lin7", inf))
    parity.check(BF16, got2l[rs], ref2.reshape(-1), rel=2e-3, max_ulp=1, max_frac=0.2, scale_aware=False, what="w2 rows, linear order, the token's grid")
    parity.exact(got2l, got2, "w2: every one of the 4096 outputs, linear order == classic")
    # ---- output head: 128256 rows, final norm prologue
    fw = norm_weights(0xFFFF0002, dim)
    mo.rmsnorm(BF16, L((1, dim)), xn, L((1, dim)), x.reshape(1, -1), L((dim,)), fw, M["norm_eps"], 0.0)
    vs = np.unique(np.concatenate([[0, M["vocab"] - 1], rng.integers(0, M["vocab"], 10)]))
    qh, sh = synth_rows(0xFFFF0001, vs, dim)
    refh = oracle_rows(qh, sh, xn.reshape(-1))
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(-1, "output")
    goth = run_gemv(acc, "mc_gemv_i4_bfloat_m4d_p1_e0", wptr, sptr, x, M["vocab"], rows, inf, 128, BF16, norm=fw, wgs=512)
    parity.check(BF16, goth[vs], refh, rel=2e-3, max_ulp=1, max_frac=0.2, scale_aware=False, what="head rows")
    # the head as the token launches it: the greedy pick rides in the launch (gemv.h EPI_STORE_PICK, one key per workgroup)
    assert dec.gemv_kernel_name("head") == "mc_gemv_i4_bfloat_lin2_p1_e5"
    keys = acc.to_device(np.zeros(1024, np.uint64))
    desc = acc.to_device(np.array([keys.device_ptr, 0, 0, 0], np.uint64))   # pick_epilogue {key, ticket = NULL, state, tokens_out}
    gothl = lin_launch(acc, "mc_gemv_i4_bfloat_lin2_p1_e5", wptr, sptr, x, M["vocab"], rows, inf, 128, res=desc, norm=fw, wgs=cus,
                       lds=lin_lds("i4_lin2", inf))
    parity.check(BF16, gothl[vs], refh, rel=2e-3, max_ulp=1, max_frac=0.2, scale_aware=False, what="head rows, linear order + pick, the token's grid")
    parity.exact(gothl, goth, "head: every one of the 128256 logits, linear order == classic")
    best = int(keys.download(np.uint64, 1024).max())
    logits = mo.decode(BF16, gothl) if hasattr(mo, "decode") else (gothl.astype(np.uint32) << 16).view(np.float32)
    assert 0xFFFFFFFF - (best & 0xFFFFFFFF) == int(np.argmax(logits)), "the pick of the launch == the first maximum of its logits"
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


def test_graph_replay_equals_eager_at_the_benchmark_context(acc):
    # bench.py's own configuration -- 32 blocks, 2048 slots, hipGraph replay, the three-launch layer with wq|wk|wv, the attention,
    # their four in-launch hand-offs and the Wo GEMV in ONE launch (mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2) -- from an empty cache to 64
    # tokens past its end (the sink ring turns): 67 584 launches with hand-offs per path, every one of which must find this
    # step's tags; the replayed graph must produce the eager launches' tokens, and the caches must be identical.
    n = 2048 + 64
    eager = make(acc, 32, max_seq_len=2048, use_graph=0)
    eager.launch_log(True)
    chain = list(eager.generate(7, 0, n))
    names = set(eager.launched())
    assert "mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2" in names and "mc_gemv_i4_bfloat_lin2_p1_e2" in names, sorted(names)
    assert "mc_gemv_i4_bfloat_lin2_p1_e4" not in names, sorted(names)
    assert not [x for x in names if x.startswith("mc_attn_scores") or x == "mc_attn_fused_bfloat"], sorted(names)
    ke, ve = eager.export_kv(17)
    eager.release()
    graph = make(acc, 32, max_seq_len=2048, use_graph=1)
    assert list(graph.generate(7, 0, n)) == chain
    kg, vg = graph.export_kv(17)
    assert np.array_equal(ke, kg) and np.array_equal(ve, vg)
    assert len(set(chain)) > 4                                           # not a degenerate constant stream
    graph.release()


def test_prompt_pass_equals_token_by_token_at_full_size(acc):
    # 96 prompt rows through the 128 x 128 MFMA GEMMs (split-K on Wo / w2 / QKV at this length) and
    # the fused attention, against the same tokens fed one at a time through the decode path of a
    # second decoder: same function, different summation orders -> bf16-close logits, and both
    # continue with the same tokens.
    n = 96
    toks = np.random.default_rng(11).integers(0, M["vocab"], n)
    a = make(acc, 32, max_seq_len=128)
    b = make(acc, 32, max_seq_len=128)
    ta = a.prefill(toks, 0)
    for pos, t in enumerate(toks):
        tb = b.step(int(t), pos)
    la, lb = mo.decode(BF16, a.logits()).astype(np.float64), mo.decode(BF16, b.logits()).astype(np.float64)
    assert np.all(np.isfinite(la)) and np.all(np.isfinite(lb))
    nrm = np.linalg.norm(la - lb) / np.linalg.norm(lb)
    assert nrm < 2e-2, f"prompt pass vs stepwise logits: normwise rel {nrm:.3g}"
    ka, va = a.export_kv(31)
    kb, vb = b.export_kv(31)
    assert ka.shape == kb.shape == (n, M["n_kv_heads"], M["head_dim"])
    kd = np.linalg.norm(mo.decode(BF16, ka).astype(np.float64) - mo.decode(BF16, kb)) / np.linalg.norm(mo.decode(BF16, kb).astype(np.float64))
    assert kd < 2e-2, f"last-layer K cache: normwise rel {kd:.3g}"
    # greedy continuation: near-ties aside, the two decoders walk the same path
    same, t1, t2 = 0, ta, tb
    for i in range(8):
        same += int(t1 == t2)
        t1, t2 = a.step(t1, n + i), b.step(t2, n + i)
    assert same >= 6, (same, ta, tb)
    a.release()
    b.release()


# ------------------------------------------------------------------------------------------------
# End to end at full WIDTH against the oracle: the host regenerates the synthetic model (numpy
# port of synth.h, tests/synthgen.py) so the oracle runs exactly what init_synthetic put in HBM.
# Vocabulary and depth are reduced (the 128256-row head is covered by the sampled-row test).
# ------------------------------------------------------------------------------------------------
def synth_model(cfg, seed, bits=4, group=128):
    import synthgen as sg

    dt = cfg["dtype"]
    dim, H, KV, hd, ffn = cfg["dim"], cfg["n_heads"], cfg["n_kv_heads"], cfg["head_dim"], cfg["ffn_dim"]

    def lin(mid, out_f, in_f):
        return dict(kind=1, weight=sg.weights(seed, mid, out_f, in_f, bits),
                    scales=sg.scales(seed, mid, out_f, in_f // group, in_f, bits), group_size=group,
                    hbm_format=2 if bits == 4 else 1)

    def vec(mid, n):
        return mo.encode(dt, sg.values(seed, mid, n, 0))

    layers = []
    for i in range(cfg["n_layers"]):
        b = i * 16
        lw = dict(wq=lin(b + 0, H * hd, dim), wk=lin(b + 1, KV * hd, dim), wv=lin(b + 2, KV * hd, dim),
                  wo=lin(b + 3, dim, H * hd), w1=lin(b + 4, ffn, dim), w2=lin(b + 5, dim, ffn), w3=lin(b + 6, ffn, dim),
                  attention_norm=vec(b + 8, dim), ffn_norm=vec(b + 9, dim))
        if cfg.get("family", 0) == 1:
            lw.update(q_norm=vec(b + 10, hd), k_norm=vec(b + 11, hd), attention_post_norm=vec(b + 12, dim),
                      ffn_post_norm=vec(b + 13, dim))
            stride = cfg.get("sliding_stride", 0)
            lw["rope_table"] = 1 if (stride and (i + 1) % stride != 0 and cfg.get("rope_sliding_theta", 0.0) > 0) else 0
        layers.append(lw)
    emb = mo.encode(dt, sg.values(seed, 0xFFFF0000, cfg["vocab"] * dim, 1).reshape(cfg["vocab"], dim))
    return dict(layers=layers, embedding=dict(kind=0, weight=emb), output=lin(0xFFFF0001, cfg["vocab"], dim),
                final_norm=vec(0xFFFF0002, dim))


def test_numpy_generator_equals_the_abi_generator():
    import metalchat_amd as mc
    import synthgen as sg

    lib = mc.capi()
    w = sg.weights(SEED, 21, 37, 64, 4)
    s = sg.scales(SEED, 21, 37, 5, 640, 4)
    v0, v1, v2 = sg.values(SEED, 40, 50, 0), sg.values(SEED, 0xFFFF0000, 50, 1), sg.values(SEED, 7, 50, 2, 96)
    for r in (0, 5, 36):
        assert [lib.mc_synth_weight(SEED, 21, r, c, 4) for c in range(64)] == w[r].tolist()
        assert np.array_equal(np.array([lib.mc_synth_scale(SEED, 21, r, g, 640, 4) for g in range(5)], np.float32), s[r])
    assert sg.weights(SEED, 3, 9, 32, 8).min() >= -127
    for i in (0, 17, 49):
        assert np.float32(lib.mc_synth_value(SEED, 40, i, 0, 1)) == v0[i]
        assert np.float32(lib.mc_synth_value(SEED, 0xFFFF0000, i, 1, 1)) == v1[i]
        assert np.float32(lib.mc_synth_value(SEED, 7, i, 2, 96)) == v2[i]


FULL_WIDTH = {
    # Llama-3-8B widths (GQA 4, head_dim 128, ffn 14336), Gemma-7B widths (MHA, head_dim 256, ffn 24576)
    "llama3-8b": dict(family=0, dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, rope_theta=500000.0,
                      attn_scale=128 ** -0.5),
    "gemma-7b": dict(family=1, dim=3072, n_heads=16, n_kv_heads=16, head_dim=256, ffn_dim=24576, rope_theta=10000.0,
                     rope_sliding_theta=10000.0, sliding_stride=2, attn_scale=256 ** -0.5),  # block 0 is a sliding layer
}


@pytest.mark.parametrize("name", sorted(FULL_WIDTH))
def test_end_to_end_against_the_oracle_at_full_width(acc, name):
    import metalchat_amd as mc
    import modelgen as mg

    cfg = dict(dtype=BF16, n_layers=1, vocab=2048, max_seq_len=32, norm_eps=1e-5, **FULL_WIDTH[name])  # one block: the host generator takes ~30 s per block
    weights = synth_model(cfg, SEED)
    om = mo.Model(cfg, weights)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=128))
    dec.init_synthetic(SEED)
    dec.set_taps(True)
    tok, agree = 5, 0
    for pos in range(6):
        otok, ologits = om.step(tok, pos)
        got = dec.step(tok, pos)
        for layer in range(-1, cfg["n_layers"]):
            # vector-wise bound: one bf16 step (2^-8 = 3.9e-3); element-wise: 2 scaled steps
            parity.check(BF16, dec.hidden(layer), om.hidden(layer), rel=3.9e-3, max_ulp=2 if layer >= 0 else 0,
                         max_frac=0.5 if layer >= 0 else 0.0, what=f"{name} pos {pos} hidden[{layer}]")
        # (gemma: four norms per block and a 24576-long w2 reduction -- more logits land on the
        # neighbouring bf16 value than for llama; HOW FAR stays bounded by 2 scaled steps)
        # vector-wise 5e-3: a logit is a 4096-term bf16 dot product of a hidden row that already
        # differs in its last bit here and there (one bf16 step is 3.9e-3 relative)
        parity.check(BF16, dec.logits(), ologits, rel=5e-3, max_ulp=2, max_frac=0.7, what=f"{name} pos {pos} logits")
        agree += int(got == otok)
        tok = otok
    gk, gv = dec.export_kv(cfg["n_layers"] - 1)
    ok, ov = om.kv(cfg["n_layers"] - 1)
    parity.check(BF16, gk, ok, rel=3.9e-3, max_ulp=2, max_frac=0.5, what=f"{name} K")
    parity.check(BF16, gv, ov, rel=3.9e-3, max_ulp=2, max_frac=0.5, what=f"{name} V")
    assert agree >= 5
    # and the prompt pass of the same model (fused attention at this head_dim, split-K GEMMs)
    om2 = mo.Model(cfg, weights)
    ptoks = np.random.default_rng(2).integers(0, cfg["vocab"], 20)
    otok, ologits = om2.forward(ptoks, 0, 16 if cfg["family"] == 1 else 0)
    d2 = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=128))
    d2.init_synthetic(SEED)
    gtok = d2.prefill(ptoks, 0, 16 if cfg["family"] == 1 else 0)
    parity.check(BF16, d2.logits(), ologits, rel=7.8e-3, max_ulp=2, max_frac=0.7, what=f"{name} prompt logits")
    for d in (dec, d2):
        d.release()
    om.close()
    om2.close()


def test_int8_long_context_against_the_oracle_at_full_width(acc, monkeypatch):
    # BASELINE configs[2] territory: int8-held weights, a long context.  One Llama-3-8B-wide block,
    # max_seq_len 4096 with P.V forced into 4 ranges of cache slots + the reduce launch (the layout a
    # context of 8192 slots takes by itself), a 320-token prompt (two row tiles of the prompt GEMM; 600 tokens cost the oracle 55 s)
    # through the prompt pass (128 x 128 MFMA GEMMs on int8 weights, split-K) and decode steps behind
    # it -- all against the oracle on the regenerated weights.
    import metalchat_amd as mc
    import modelgen as mg

    monkeypatch.setenv("MC_PV_RANGES", "4")
    cfg = dict(dtype=BF16, n_layers=1, vocab=2048, max_seq_len=4096, norm_eps=1e-5, **FULL_WIDTH["llama3-8b"])
    weights = synth_model(cfg, SEED, bits=8)
    om = mo.Model(cfg, weights)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I8, group_size=128))
    dec.init_synthetic(SEED)
    n = 320
    ptoks = np.random.default_rng(4).integers(0, cfg["vocab"], n)
    otok, ologits = om.forward(ptoks, 0, 0)
    gtok = dec.prefill(ptoks, 0)
    # the rows feed each other through attention: measured 0.006 vector-wise and up to 2.5 scaled bf16
    # steps on single logits at 600 rows (0.005 / 2.05 at 100 rows)
    parity.check(BF16, dec.logits(), ologits, rel=7.8e-3, max_ulp=3, max_frac=0.9, what="int8 prompt logits")
    gk, gv = dec.export_kv(0)
    ok, ov = om.kv(0)
    assert gk.shape == ok.shape == (n, 8, 128)
    # K / V rows of the first block depend on their own row only: a handful of last-bit differences.  K is a composition -- the
    # GEMM's T(row sum), then the rotation c x1 - s x2 in fp32 -- so one step in x1 can reach the output as two where the two
    # products cancel (seen once, 1.12 of one step at the rms, with the 256-row tile's two K splits in place of four): the
    # suite's composition bound, two steps; V is the GEMM's output as it stands: one
    parity.check(BF16, gk, ok, rel=1e-3, max_ulp=2, max_frac=0.02, what="int8 prompt K")
    parity.check(BF16, gv, ov, rel=1e-3, max_ulp=1, max_frac=0.02, what="int8 prompt V")
    tok, agree = otok, int(gtok == otok)
    for i in range(4):
        o2, ol2 = om.step(tok, n + i)
        g2 = dec.step(tok, n + i)
        parity.check(BF16, dec.logits(), ol2, rel=7.8e-3, max_ulp=3, max_frac=0.9, what=f"int8 decode at context {n + i}")
        agree += int(g2 == o2)
        tok = o2
    assert agree >= 4
    dec.release()
    om.close()
