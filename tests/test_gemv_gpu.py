"""Kernel-level parity of the fused GEMV family (metalchat_amd/csrc/kernels/gemv.h) driven
through the C-ABI encoder, against the oracle's restatement of the reference's two-kernel
quantised linear (hadamard_broadcast + bmm: kernel/mul.metal:51-85, kernel/bmm.metal:25-82).

Only the fp32 summation order differs from the reference here, so the bf16 bound is the strict
one: every output within 1 bf16 step of the oracle and almost all of them bit-identical."""
import numpy as np
import pytest

import modelgen as mg
import parity
from oracle import mc_oracle as mo

pytestmark = pytest.mark.gpu
BF16, F32 = 0, 1


def oracle_linear(dt, spec, x_T):
    """y = T(x . Wd^T) with the oracle's kernels: dequantise with hadamard_broadcast, then bmm."""
    L = mo.layout
    w = spec["weight"]
    out_f, in_f = w.shape
    if spec["kind"] == 0:
        wd = w
    else:
        G = spec["group_size"] or in_f
        ng = in_f // G
        wd = np.zeros((out_f, in_f), dtype=mo.np_dtype(dt))
        s = np.ascontiguousarray(spec["scales"].reshape(-1), dtype=np.float32)
        # weight viewed [out*ng, G], scales viewed [out*ng]  (quantization/lora.h:105-110)
        mo.hadamard_broadcast(dt, F32, L((out_f * ng, G)), wd, L((out_f * ng, G)), w, L((out_f * ng,)), s)
    y = np.zeros((1, 1, out_f), dtype=mo.np_dtype(dt))
    # matmul(x, W.transpose): B = [1, K, N] view of row-major [N, K]  (nn/linear.h:70-81)
    bl = L((1, in_f, out_f), strides=(in_f * out_f, 1, in_f))
    mo.bmm(dt, L(y.shape), y, L((1, 1, in_f)), x_T, bl, wd)
    return y.reshape(-1)


def gemv_name(fmt, dt, pro, epi, fast=False, variant=""):
    return "mc_gemv_" + {0: "w", 1: "i8", 2: "i4"}[fmt] + "_" + ("bfloat" if dt == BF16 else "float") + \
        ("_fast" if fast else "") + variant + f"_p{pro}_e{epi}"


def run_gemv(acc, name, wptr, sptr, x_T, out_n, rows, in_f, group, dt, res=None, norm=None,
             eps=1e-5, mu=0.0, block=256, wgs=64, lora=None):
    import metalchat_amd as mc

    tb = 2 if dt == BF16 else 4
    k = acc.load(name)
    xb = acc.to_device(x_T)
    yb = acc.alloc(out_n * tb)
    wbuf = acc.wrap(wptr, 1 << 40)
    sbuf = acc.wrap(sptr, 1 << 40) if sptr else None
    rb = acc.to_device(res) if res is not None else None
    nb = acc.to_device(norm) if norm is not None else None
    kpl = {0: 8 if dt == BF16 else 4, 1: 16, 2: 32}[{"w": 0, "i8": 1, "i4": 2}[name.split("_")[2]]]
    chunk = 64 * kpl
    lds = (in_f + chunk - 1) // chunk * chunk * tb
    if "_m4d_" in name:
        lds = lds // 16 * 17  # the row is padded by 16 bytes per 256 for the transposed reads
    lds += 64
    t = mc.KernelTask(k, (wgs * block, 1, 1), (block, 1, 1),
                      [wbuf, sbuf, xb, yb, rb, nb, np.uint32(rows), np.uint32(in_f), np.uint32(group),
                       np.float32(eps), np.float32(mu)] +
                      ([acc.to_device(lora[0]), acc.to_device(lora[1]), np.uint32(lora[0].size),
                        np.float32(lora[2])] if lora else [None, None, np.uint32(0), np.float32(0)]),
                      lds_bytes=lds)
    t()
    acc.wait()
    return yb.download(np.uint16 if dt == BF16 else np.float32, out_n)


@pytest.fixture(scope="module")
def holder(acc):
    """A decoder is used only as the host-side packer: load reference-native weights, get the HBM
    addresses of the fused buffers back (mc_decoder_weight_ptrs)."""
    import metalchat_amd as mc

    made = []

    def make(dt, quant, group, dim=512, ffn=1024, seed=0):
        cfg = mg.tiny_cfg(dt, dim=dim, n_heads=4, n_kv_heads=2, head_dim=64, ffn_dim=ffn, n_layers=1,
                          vocab=256, max_seq_len=16)
        w = mg.make_model(cfg, seed=seed, quant=quant, group=group)
        fmt = {None: 0, "i8": 1, "i4": 2}[quant]
        dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=fmt, group_size=group if quant else 0))
        dec.load_model(w)
        made.append(dec)
        return cfg, w, dec, fmt

    yield make
    for d in made:
        d.release()


CASES = [(F32, None, 0), (F32, "i8", 32), (F32, "i4", 32), (F32, "i4", 128),
         (BF16, None, 0), (BF16, "i8", 32), (BF16, "i4", 32), (BF16, "i4", 128), (BF16, "i4", 256)]


@pytest.mark.parametrize("dt,quant,group", CASES)
def test_gemv_store_matches_two_kernel_reference(acc, holder, dt, quant, group):
    cfg, w, dec, fmt = holder(dt, quant, group, seed=21)
    rng = np.random.default_rng(5)
    lw = w["layers"][0]
    # wo: [dim, H*hd] plain rows; w2: [dim, ffn]
    for name, key in (("wo", "wo"), ("w2", "w2")):
        spec = lw[key]
        out_f, in_f = spec["weight"].shape
        x = mo.encode(dt, rng.normal(0, 1, in_f).astype(np.float32))
        wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, name)
        assert (rows, inf) == (out_f, in_f)
        got = run_gemv(acc, gemv_name(fmt, dt, 0, 0), wptr, sptr, x, out_f, out_f, in_f,
                       group if quant else 0, dt)
        ref = oracle_linear(dt, spec, x)
        if dt == F32:
            parity.check(dt, got, ref, rel=1e-5, what=f"{name} f32")
        else:
            r = parity.check(dt, got, ref, rel=1e-3, max_ulp=1, max_frac=0.01, scale_aware=False,
                             what=f"{name} bf16")
            assert r["frac"] <= 0.01


@pytest.mark.parametrize("dt", [F32, BF16])
def test_gemv_fused_qkv_rows_and_rmsnorm_prologue(acc, holder, dt):
    cfg, w, dec, fmt = holder(dt, "i4", 32, seed=22)
    L = mo.layout
    rng = np.random.default_rng(6)
    lw = w["layers"][0]
    dim = cfg["dim"]
    x = mo.encode(dt, rng.normal(0, 1, dim).astype(np.float32))
    nw = lw["attention_norm"]
    xn = np.zeros((1, dim), dtype=mo.np_dtype(dt))
    mo.rmsnorm(dt, L((1, dim)), xn, L((1, dim)), x.reshape(1, dim), L((dim,)), nw, 1e-5, 0.0)
    # q and k rows are stored with the rotation partners (j, j + hd/2) adjacent (gemv.h)
    def packed(y, hd):
        y = y.reshape(-1, 2, hd // 2)          # [head, e, j]
        return np.ascontiguousarray(y.transpose(0, 2, 1)).reshape(-1)  # [head, j, e]
    hd = cfg["head_dim"]
    ref = np.concatenate([packed(oracle_linear(dt, lw["wq"], xn.reshape(-1)), hd),
                          packed(oracle_linear(dt, lw["wk"], xn.reshape(-1)), hd),
                          oracle_linear(dt, lw["wv"], xn.reshape(-1))])
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "qkv")
    got = run_gemv(acc, gemv_name(fmt, dt, 1, 0), wptr, sptr, x, rows, rows, inf, 32, dt, norm=nw)
    if dt == F32:
        parity.check(dt, got, ref, rel=1e-5, what="qkv f32")
    else:
        # the normalised row can differ from the oracle's by one bf16 step in a few elements
        # (block-sum order), which then reaches every output: composition bound
        parity.check(dt, got, ref, rel=2e-3, max_ulp=2, max_frac=0.2, what="qkv bf16")


@pytest.mark.parametrize("dt", [F32, BF16])
def test_gemv_w13_silu_mul_and_residual_epilogues(acc, holder, dt):
    cfg, w, dec, fmt = holder(dt, "i4", 32, seed=23)
    L = mo.layout
    rng = np.random.default_rng(7)
    lw = w["layers"][0]
    dim, ffn = cfg["dim"], cfg["ffn_dim"]
    x = mo.encode(dt, rng.normal(0, 1, dim).astype(np.float32))
    nw = lw["ffn_norm"]
    xn = np.zeros((1, dim), dtype=mo.np_dtype(dt))
    mo.rmsnorm(dt, L((1, dim)), xn, L((1, dim)), x.reshape(1, dim), L((dim,)), nw, 1e-5, 0.0)
    g1 = oracle_linear(dt, lw["w1"], xn.reshape(-1)).reshape(1, ffn)
    g3 = oracle_linear(dt, lw["w3"], xn.reshape(-1)).reshape(1, ffn)
    a = np.zeros_like(g1)
    mo.silu(dt, L(g1.shape), a, L(g1.shape), g1)
    ref = np.zeros_like(g1)
    mo.hadamard(dt, L(g1.shape), ref, L(g1.shape), a, L(g1.shape), g3)
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w13")
    assert rows == 2 * ffn
    got = run_gemv(acc, gemv_name(fmt, dt, 1, 2), wptr, sptr, x, ffn, rows, inf, 32, dt, norm=nw)
    if dt == F32:
        parity.check(dt, got, ref, rel=2e-5, what="w13 silu f32")
    else:
        parity.check(dt, got, ref, rel=3e-3, max_ulp=2, max_frac=0.3, what="w13 silu bf16")
    # residual epilogue: out = T(res + T(w2 g))
    g = mo.encode(dt, rng.normal(0, 1, ffn).astype(np.float32))
    res = mo.encode(dt, rng.normal(0, 1, dim).astype(np.float32))
    y = oracle_linear(dt, lw["w2"], g).reshape(1, dim)
    ref2 = np.zeros_like(y)
    mo.add(dt, L(y.shape), ref2, L(y.shape), res.reshape(1, dim), L(y.shape), y)
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w2")
    got2 = run_gemv(acc, gemv_name(fmt, dt, 0, 1), wptr, sptr, g, dim, rows, inf, 32, dt, res=res)
    if dt == F32:
        parity.check(dt, got2, ref2, rel=1e-5, what="w2 resid f32")
    else:
        parity.check(dt, got2, ref2, rel=1e-3, max_ulp=1, max_frac=0.01, scale_aware=False, what="w2 resid bf16")


@pytest.mark.parametrize("dt", [F32, BF16])
def test_gemv_lora_adaptation_term(acc, holder, dt):
    """The row-result hook of lora_linear (quantization/lora.h:119-121) against the reference's op
    sequence: a = bmm(x, A^T), b = bmm(a, B^T), ad = scalar_mul(b, scale), y = add(x Wd^T, ad)."""
    L = mo.layout
    cfg, w, dec, fmt = holder(dt, "i4", 32, seed=27)
    rng = np.random.default_rng(9)
    spec = w["layers"][0]["wo"]
    out_f, in_f = spec["weight"].shape
    rank, scale = 16, 2.0
    x = mo.encode(dt, rng.normal(0, 1, in_f).astype(np.float32))
    A = mo.encode(dt, (rng.uniform(-1, 1, (rank, in_f)) / np.sqrt(in_f)).astype(np.float32))
    B = mo.encode(dt, (rng.uniform(-1, 1, (out_f, rank)) * 0.1).astype(np.float32))
    npd = mo.np_dtype(dt)
    a = np.zeros((1, 1, rank), npd)
    mo.bmm(dt, L(a.shape), a, L((1, 1, in_f)), x, L((1, in_f, rank), strides=(in_f * rank, 1, in_f)), A)
    b = np.zeros((1, 1, out_f), npd)
    mo.bmm(dt, L(b.shape), b, L((1, 1, rank)), a, L((1, rank, out_f), strides=(rank * out_f, 1, rank)), B)
    ad = np.zeros(out_f, npd)
    mo.scalar_mul(dt, L((1, out_f)), ad, L((1, out_f)), b.reshape(-1), scale)
    base = oracle_linear(dt, spec, x)
    ref = np.zeros(out_f, npd)
    mo.add(dt, L((1, out_f)), ref, L((1, out_f)), base, L((1, out_f)), ad)
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "wo")
    got = run_gemv(acc, gemv_name(fmt, dt, 0, 0), wptr, sptr, x, out_f, out_f, in_f, 32, dt,
                   lora=(a.reshape(-1), B, scale))
    if dt == F32:
        parity.check(dt, got, ref, rel=1e-5, what="lora f32")
    else:
        parity.check(dt, got, ref, rel=1e-3, max_ulp=1, max_frac=0.02, what="lora bf16")
    # rank 0 = no adaptor: the plain result
    plain = run_gemv(acc, gemv_name(fmt, dt, 0, 0), wptr, sptr, x, out_f, out_f, in_f, 32, dt)
    assert not np.array_equal(plain, got)


@pytest.mark.parametrize("group", [128, pytest.param(256, marks=pytest.mark.slow)])   # (BASELINE's group; 256: the same sweep, 16 s)
def test_gemv_matrix_pipe_dequant_is_exact_for_every_weight(acc, holder, group):
    """Q_M4D dequantises on v_mfma_f32_4x4x4_16b_bf16 and gathers the activations through
    ds_read_b64_tr_b16.  A one-hot row x = e_k makes y[o] = Wd[o, k], so every position of the
    128-weight blocks, in whole and in ragged chunks, is compared BIT FOR BIT with the oracle's
    T(T(q) * T(s)) -- and a wrong gather would pick the wrong column."""
    cfg, w, dec, fmt = holder(BF16, "i4", group, dim=2304, ffn=4096, seed=33)
    lw = w["layers"][0]
    for name, key in (("w2", "w2"), ("wo", "wo")):
        spec = lw[key]
        out_f, in_f = spec["weight"].shape
        wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, name)
        ks = sorted(set(list(range(0, 128)) + list(range(in_f - 128, in_f)) + list(range(2048 - 64, min(in_f, 2048 + 64))) +
                        list(range(128 * 5, 128 * 6, 3))))
        ks = [k for k in ks if 0 <= k < in_f]
        for k in ks:
            xf = np.zeros(in_f, np.float32)
            xf[k] = 1.0
            x = mo.encode(BF16, xf)
            got = run_gemv(acc, gemv_name(fmt, BF16, 0, 0, variant="_m4d"), wptr, sptr, x, out_f, out_f, in_f, group, BF16)
            ref = oracle_linear(BF16, spec, x)
            assert np.array_equal(got, ref), f"{name} column {k}: {np.flatnonzero(got != ref)[:8]}"
    # and an ordinary row: same answer as the VALU dequantisation up to the fp32 summation order
    rng = np.random.default_rng(9)
    spec = lw["w2"]
    out_f, in_f = spec["weight"].shape
    x = mo.encode(BF16, rng.normal(0, 1, in_f).astype(np.float32))
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w2")
    got = run_gemv(acc, gemv_name(fmt, BF16, 0, 0, variant="_m4d"), wptr, sptr, x, out_f, out_f, in_f, group, BF16)
    r = parity.check(BF16, got, oracle_linear(BF16, spec, x), rel=1e-3, max_ulp=1, max_frac=0.01, scale_aware=False,
                     what="w2 m4d")
    assert r["frac"] <= 0.01


def test_gemv_geometry_independent(acc, holder):
    """Integer/indexing property: the result does not depend on how rows are dealt to waves."""
    cfg, w, dec, fmt = holder(BF16, "i4", 32, seed=24)
    rng = np.random.default_rng(8)
    x = mo.encode(BF16, rng.normal(0, 1, cfg["ffn_dim"]).astype(np.float32))
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w2")
    base = run_gemv(acc, gemv_name(fmt, BF16, 0, 0), wptr, sptr, x, rows, rows, inf, 32, BF16, block=256, wgs=1)
    for block, wgs in ((64, 3), (128, 7), (512, 2), (256, 200)):
        got = run_gemv(acc, gemv_name(fmt, BF16, 0, 0), wptr, sptr, x, rows, rows, inf, 32, BF16,
                       block=block, wgs=wgs)
        parity.exact(got, base, f"block {block} wgs {wgs}")


def test_synthetic_weights_match_host_generator(acc):
    """mc_decoder_init_synthetic fills HBM with exactly the values mc_synth_* return on the host
    (the full-size benchmark model can therefore be spot-checked row by row)."""
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(BF16, dim=256, n_heads=4, n_kv_heads=2, head_dim=64, ffn_dim=512, n_layers=2,
                      vocab=256, max_seq_len=16)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=2, group_size=128))
    seed = 99
    dec.init_synthetic(seed)
    lib = mc.capi()
    perm = [0, 2, 4, 6, 1, 3, 5, 7]
    H, KV, hd = 4, 2, 64
    for layer in (0, 1):
        for name, nrows in (("qkv", 12), ("w13", 9), ("wo", 5)):
            wptr, sptr, rows, inf, ng = dec.weight_ptrs(layer, name)
            wb = acc.wrap(wptr, rows * inf // 2)
            sb = acc.wrap(sptr, (rows + 3) // 4 * 4 * ng * 2)
            for r in np.linspace(0, rows - 1, nrows).astype(int):
                packed = wb.download(np.uint32, inf // 8, offset=int(r) * inf // 2)
                # scales live in row quads [rows/4][ng][4] (gemv.h)
                quad = sb.download(np.uint16, ng * 4, offset=(int(r) // 4) * ng * 4 * 2)
                sc = quad.reshape(ng, 4)[:, int(r) % 4]
                if name == "qkv":
                    def natural(lr):  # packed row head*hd + 2j + e -> natural head*hd + j + e*hd/2
                        head, w = lr // hd, lr % hd
                        return head * hd + (w >> 1) + (w & 1) * (hd // 2)
                    if r < H * hd: m, sr = layer * 16 + 0, natural(r)
                    elif r < (H + KV) * hd: m, sr = layer * 16 + 1, natural(r - H * hd)
                    else: m, sr = layer * 16 + 2, r - (H + KV) * hd
                elif name == "w13":
                    m, sr = layer * 16 + (6 if r & 1 else 4), r >> 1
                else:
                    m, sr = layer * 16 + 3, r
                exp = np.array([lib.mc_synth_weight(seed, m, int(sr), c, 4) for c in range(inf)])
                got = np.zeros(inf, dtype=np.int64)
                for dw in range(inf // 8):
                    for p in range(8):
                        got[dw * 8 + perm[p]] = ((int(packed[dw]) >> (4 * p)) & 15) - 8
                parity.exact(got, exp, f"{name} row {r}")
                exps = mo.to_bf16(np.array([lib.mc_synth_scale(seed, m, int(sr), g, inf, 4)
                                            for g in range(ng)], np.float32))
                parity.exact(sc, exps, f"{name} scales row {r}")
    dec.release()


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("quant,group,dim,ffn", [("i4", 32, 96 * 32, 5632), ("i8", 32, 1056, 2080), (None, 0, 224, 352)])
def test_gemv_ragged_rows_and_row_tails(acc, holder, dt, quant, group, dim, ffn):
    """Edge cases of the tiling: `in` not a multiple of the 64-lane chunk (Gemma dim 3072 and
    TinyLlama ffn 5632 with int4: chunk 2048), and an output row count that is not a multiple of
    the four rows a wavefront owns (vocab-like 250 rows through the head)."""
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(dt, dim=dim, n_heads=2, n_kv_heads=1, head_dim=32, ffn_dim=ffn, n_layers=1,
                      vocab=250, max_seq_len=16)
    w = mg.make_model(cfg, seed=41, quant=quant, group=group or 32)
    fmt = {None: 0, "i8": 1, "i4": 2}[quant]
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=fmt, group_size=group))
    dec.load_model(w)
    rng = np.random.default_rng(9)
    lw = w["layers"][0]
    for name, spec in (("w2", lw["w2"]), ("output", w["output"])):
        out_f, in_f = spec["weight"].shape
        x = mo.encode(dt, rng.normal(0, 1, in_f).astype(np.float32))
        wptr, sptr, rows, inf, ng = dec.weight_ptrs(0 if name == "w2" else -1, name)
        got = run_gemv(acc, gemv_name(fmt, dt, 0, 0), wptr, sptr, x, out_f, out_f, in_f, group, dt, wgs=7)
        ref = oracle_linear(dt, dict(spec, group_size=group), x)
        if dt == F32:
            parity.check(dt, got, ref, rel=1e-5, what=f"{name} ragged f32")
        else:
            parity.check(dt, got, ref, rel=1e-3, max_ulp=1, max_frac=0.01, scale_aware=False,
                         what=f"{name} ragged bf16")
    dec.release()
