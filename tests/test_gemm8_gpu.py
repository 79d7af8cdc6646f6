"""Kernel-level parity of the round-5 prompt GEMM (`mc_pf_gemm8_{i4,i8,w}_bfloat_e{0,1,2,3}`, kernels/pf_gemm8.h: 256 x 256 tiles, two
wave groups alternating between LDS and the matrix pipe, quantised W dequantised inside the loop) -- the kernels `mc_decoder_prefill`
launches for prompts of 384 rows and more (decoder.cc g8_ok), launched BY NAME through the Part-1 seam on weights packed by the decoder, against the
oracle's restatement of the reference's linear on len > 1 rows: hadamard_broadcast (Wd = T(T(q) T(s)), kernel/mul.metal:51-85) + bmm
(kernel/bmm.metal:25-82, nn/linear.h:70-81), residual add in T (kernel/arithmetic.metal:13-46), silu and hadamard in T
(kernel/activation.metal:13-78, kernel/mul.metal:13-48) as nn/transformer.h:53-59 composes them.

Bounds: the single-kernel bound of the suite (every output within ONE bf16 step of the oracle -- a step measured at max(|value|, rms) --, at
most 1 % different at all: the fp32 sums are added in another order, nothing else differs); one-hot rows BIT FOR BIT; the activation epilogue within two steps (a one-step difference in either
factor moves the product by at most one more).  Ragged M (rows past M read as zeros through the buffer's bounds and are not stored),
M below and above one tile, K ranges (split-K) summed on the host in z order, scale groups of 32 (two per K tile) and 128."""
import numpy as np
import pytest

import modelgen as mg
import parity
from oracle import mc_oracle as mo

pytestmark = pytest.mark.gpu
BF16, F32 = 0, 1

FORMATS = {"i4g128": ("i4", 2, 128), "i4g32": ("i4", 2, 32), "i8g32": ("i8", 1, 32), "w": (None, 0, 0)}


class Models:
    def __init__(self, acc):
        self.acc, self.made = acc, {}

    def get(self, fmt):
        import metalchat_amd as mc

        if fmt not in self.made:
            quant, code, group = FORMATS[fmt]
            # wq|wk|wv [1024][512], wo [512][512], w1|w3 [1280][512] (5 column tiles), w2 [512][640] (K = 10 tiles)
            cfg = mg.tiny_cfg(BF16, dim=512, n_heads=4, n_kv_heads=2, head_dim=128, ffn_dim=640, n_layers=1, vocab=64, max_seq_len=16)
            w = mg.make_model(cfg, seed=sum(map(ord, fmt)), quant=quant, group=group or 32)
            dec = mc.Decoder(self.acc, **mg.decoder_kwargs(cfg, weight_format=code, group_size=group))
            dec.load_model(w)
            self.made[fmt] = (cfg, w, dec)
        return self.made[fmt]

    def close(self):
        for _, _, d in self.made.values():
            d.release()


@pytest.fixture(scope="module")
def models(acc):
    m = Models(acc)
    yield m
    m.close()


def oracle_rows(spec, X):
    """Y[M][out] = T(X Wd^T): hadamard_broadcast, then ONE bmm over the M rows (the reference's len > 1 call)"""
    L = mo.layout
    w = spec["weight"]
    out_f, in_f = w.shape
    if spec["kind"] == 0:
        wd = w
    else:
        G = spec["group_size"] or in_f
        ng = in_f // G
        wd = np.zeros((out_f, in_f), dtype=mo.np_dtype(BF16))
        s = np.ascontiguousarray(spec["scales"].reshape(-1), dtype=np.float32)
        mo.hadamard_broadcast(BF16, F32, L((out_f * ng, G)), wd, L((out_f * ng, G)), w, L((out_f * ng,)), s)
    M = X.shape[0]
    y = np.zeros((1, M, out_f), dtype=mo.np_dtype(BF16))
    bl = L((1, in_f, out_f), strides=(in_f * out_f, 1, in_f))
    mo.bmm(BF16, L(y.shape), y, L((1, M, in_f)), np.ascontiguousarray(X), bl, wd)
    return y.reshape(M, out_f)


def launch(acc, name, wptr, sptr, X, M, N, K, group, out_elems, out_dtype=np.uint16, res=None, splits=1):
    import metalchat_amd as mc

    xb = acc.to_device(X.reshape(-1))
    yb = acc.to_device(np.full(out_elems, 0x7FC0 if out_dtype == np.uint16 else np.nan, out_dtype))  # (poisoned: a tile nobody stores shows)
    rb = acc.to_device(res.reshape(-1)) if res is not None else None
    t = mc.KernelTask(acc.load(name), (((N + 255) // 256) * 512, (M + 255) // 256, splits), (512, 1, 1),
                      [acc.wrap(wptr, 1 << 40), (acc.wrap(sptr, 1 << 40) if sptr else None), xb, yb, rb, np.uint32(M), np.uint32(N), np.uint32(K),
                       np.uint32(group), None, None, np.uint32(0), np.float32(0)])
    t()
    acc.wait()
    return yb.download(out_dtype, out_elems)


def strict(got, ref, what):
    # one bf16 step measured at max(|value|, rms of the row set): with 10^5 outputs per case a few lie so close to zero that the
    # order of the fp32 additions alone moves them by several of THEIR steps (the GEMV tests, 256 outputs per case, never meet one)
    return parity.check(BF16, got, ref, rel=1e-3, max_ulp=1, max_frac=0.01, scale_aware=True, what=what)


@pytest.mark.parametrize("fmt", list(FORMATS))
@pytest.mark.parametrize("M", [192, 256, 300, 513])
def test_store_residual_and_partial_sums_match_the_oracle(acc, models, fmt, M):
    cfg, w, dec = models.get(fmt)
    group = FORMATS[fmt][2]
    f = fmt[:2] if fmt != "w" else "w"
    rng = np.random.default_rng(M)
    for which, key in (("qkv", None), ("w2", "w2"), ("wo", "wo")):
        wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, which)
        if key is None:
            continue   # (wq|wk|wv rows are permuted for the rotation's pairs: w2 and wo are in the reference's row order)
        spec = w["layers"][0][key]
        X = mo.encode(BF16, rng.normal(0, 1, (M, inf)).astype(np.float32))
        ref = oracle_rows(spec, X)
        got = launch(acc, f"mc_pf_gemm8_{f}_bfloat_e0", wptr, sptr, X, M, rows, inf, group, M * rows)
        strict(got, ref.reshape(-1), f"{fmt} {which} e0 M {M}")
        # + residual, added in T
        res = mo.encode(BF16, rng.normal(0, 1, (M, rows)).astype(np.float32))
        L = mo.layout
        want = np.zeros_like(ref)
        mo.add(BF16, L(ref.shape), want, L(ref.shape), res, L(ref.shape), ref)
        got = launch(acc, f"mc_pf_gemm8_{f}_bfloat_e1", wptr, sptr, X, M, rows, inf, group, M * rows, res=res)
        strict(got, want.reshape(-1), f"{fmt} {which} e1 M {M}")
        # K ranges: fp32 partial sums [splits][M][N], added in z order, one rounding (mc_pf_splitk_reduce_bfloat)
        for splits in (2, 3):
            part = launch(acc, f"mc_pf_gemm8_{f}_bfloat_e2", wptr, sptr, X, M, rows, inf, group, splits * M * rows, np.float32, splits=splits)
            part = part.reshape(splits, M, rows)
            assert np.all(np.isfinite(part)), "a tile of a K range was not stored"
            tot = np.zeros((M, rows), np.float32)
            for z in range(splits):
                tot = tot + part[z]
            strict(mo.encode(BF16, tot).reshape(-1), ref.reshape(-1), f"{fmt} {which} e2 x{splits} M {M}")


@pytest.mark.parametrize("fmt", list(FORMATS))
@pytest.mark.parametrize("M", [256, 300])
def test_activation_epilogue_matches_the_oracle(acc, models, fmt, M):
    """e3 on the fused w1|w3 matrix (rows (2 j, 2 j + 1) = (w1 row j, w3 row j)): out[m][j] = T(silu_T(T(w1 x)) * T(w3 x))"""
    cfg, w, dec = models.get(fmt)
    group = FORMATS[fmt][2]
    f = fmt[:2] if fmt != "w" else "w"
    lw = w["layers"][0]
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w13")
    rng = np.random.default_rng(M + 7)
    X = mo.encode(BF16, rng.normal(0, 1, (M, inf)).astype(np.float32))
    g1, g3 = oracle_rows(lw["w1"], X), oracle_rows(lw["w3"], X)
    L = mo.layout
    a = np.zeros_like(g1)
    mo.silu(BF16, L(g1.shape), a, L(g1.shape), g1)
    ref = np.zeros_like(g1)
    mo.hadamard(BF16, L(g1.shape), ref, L(g1.shape), a, L(g1.shape), g3)
    # the table of exponentials the epilogue reads (mc_exp_table_bfloat: exp_precise of every bfloat16)
    import metalchat_amd as mc

    etab = acc.alloc(65536 * 4)
    mc.KernelTask(acc.load("mc_exp_table_bfloat"), (256 * 256, 1, 1), (256, 1, 1), [etab])()
    acc.wait()
    xb = acc.to_device(X.reshape(-1))
    yb = acc.to_device(np.full(M * rows // 2, 0x7FC0, np.uint16))
    mc.KernelTask(acc.load(f"mc_pf_gemm8_{f}_bfloat_e3"), (((rows + 255) // 256) * 512, (M + 255) // 256, 1), (512, 1, 1),
                  [acc.wrap(wptr, 1 << 40), (acc.wrap(sptr, 1 << 40) if sptr else None), xb, yb, etab, np.uint32(M), np.uint32(rows), np.uint32(inf),
                   np.uint32(group), None, None, np.uint32(0), np.float32(0)])()
    acc.wait()
    got = yb.download(np.uint16, M * rows // 2)
    parity.check(BF16, got, ref.reshape(-1), rel=2e-3, max_ulp=2, max_frac=0.03, scale_aware=True, what=f"{fmt} w1|w3 e3 M {M}")


def test_one_hot_rows_return_every_dequantised_weight(acc, models):
    """X = unit rows e_k (k = 0 .. K - 1, M = K rows): Y[k][o] = Wd[o][k] exactly -- every nibble of every lane's 16-run, both K
    tiles of a scale group, every row of every W quarter, BIT FOR BIT against hadamard_broadcast's T(T(q) T(s))."""
    for fmt in ("i4g128", "i4g32", "i8g32"):
        cfg, w, dec = models.get(fmt)
        group = FORMATS[fmt][2]
        spec = w["layers"][0]["w2"]
        wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w2")
        X = np.zeros((inf, inf), np.float32)
        np.fill_diagonal(X, 1.0)
        Xe = mo.encode(BF16, X)
        L = mo.layout
        G = spec["group_size"]
        wd = np.zeros((rows, inf), dtype=mo.np_dtype(BF16))
        s = np.ascontiguousarray(spec["scales"].reshape(-1), dtype=np.float32)
        mo.hadamard_broadcast(BF16, F32, L((rows * (inf // G), G)), wd, L((rows * (inf // G), G)), spec["weight"], L((rows * (inf // G),)), s)
        got = launch(acc, f"mc_pf_gemm8_{fmt[:2]}_bfloat_e0", wptr, sptr, Xe, inf, rows, inf, group, inf * rows).reshape(inf, rows)
        parity.exact(got, np.ascontiguousarray(wd.T), f"{fmt}: one-hot rows")
