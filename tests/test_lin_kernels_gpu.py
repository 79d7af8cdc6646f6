"""Kernel-level parity of the LINEAR-ORDER GEMV kernels -- the ones the decoder actually launches for the BASELINE
configs and bench.py reports (`mc_gemv_i4_bfloat_lin{1,2,4,7,12,14}_*`, `_lin3s_`, `mc_gemv_i8_bfloat_ling{4,14}_*`,
`mc_gemv_w_bfloat_ling{4,8,11,16}_*`, gemv.h LNCH / LSPLIT / LGEN) -- launched BY NAME through the Part-1 seam against the
oracle's two-kernel restatement of the reference's quantised linear (hadamard_broadcast + bmm: kernel/mul.metal:51-85,
kernel/bmm.metal:25-82 composed by quantization/lora.h:94-122), with the bounds of the classic kernels' tests
(tests/test_gemv_gpu.py): every output within ONE bf16 step of the oracle, at most 1 % of them different at all.

  * store / residual / rmsnorm prologue / SiLU.mul / GELU.mul / partial-sum prologue (`_p3_`) / gemma post-norm prologue
    (`_p2_`) / RoPE + cache write (`_e4`) / greedy pick (`_e5`) for every family;
  * a ONE-HOT sweep (x = e_k makes y[o] = Wd[o, k]): every nibble position of every lane's packet of a whole 1 KiB chunk,
    the edges of every other chunk, both rows of every pair, waves whose span begins or ends in the middle of the range --
    BIT FOR BIT against T(T(q) T(s));
  * linear-order == classic kernel of the same arithmetic, bit for bit (what tools/lin_check.py only printed).

The rmsnorm prologue is made exactly reproducible by rows whose squares add without rounding (multiples of 1/4): the
normalised row is then the oracle's bit for bit and the strict bound applies to the whole kernel."""
import struct

import numpy as np
import pytest

import modelgen as mg
import parity
from oracle import mc_oracle as mo
from test_gemv_gpu import oracle_linear

pytestmark = pytest.mark.gpu
BF16 = 0
WAVES = 8

# family -> (weight format of modelgen, HBM format code, K, KiB of packed weights per row)
FAMILIES = {
    "i4_lin1": ("i4", 2, 2048), "i4_lin2": ("i4", 2, 4096), "i4_lin4": ("i4", 2, 8192), "i4_lin7": ("i4", 2, 14336),
    "i4_lin12": ("i4", 2, 24576), "i4_lin14": ("i4", 2, 28672), "i4_lin3s": ("i4", 2, 3072),
    "i8_ling4": ("i8", 1, 4096), "i8_ling14": ("i8", 1, 14336),
    "w_ling4": (None, 0, 2048), "w_ling8": (None, 0, 4096), "w_ling11": (None, 0, 5632), "w_ling16": (None, 0, 8192),
}
GROUP = 128


def kname(family, pro, epi):
    fmt, rest = family.split("_")
    return f"mc_gemv_{fmt}_bfloat_{rest}_p{pro}_e{epi}"


def classic_name(family, pro, epi):
    fmt = family.split("_")[0]
    return f"mc_gemv_{fmt}_bfloat_{'m4d_' if fmt == 'i4' else ''}p{pro}_e{epi}"


def lds_bytes(family, K, classic=False):
    """decoder.cc gemv(): the activation row padded to whole chunks (int4: + 16 bytes per 256 for the transposed reads),
    128 bytes of reduction scratch, 512 bytes of parked row sums per wave."""
    fmt = family.split("_")[0]
    chunk = {"i4": 2048, "i8": 1024, "w": 512}[fmt]
    row = (K + chunk - 1) // chunk * chunk * 2
    if family == "i4_lin3s" and not classic:
        row = 3 * 2048 * 2
    if fmt == "i4":
        row = row // 16 * 17
    return row + 128 + (0 if classic else WAVES * 512)


def launch(acc, name, wptr, sptr, x, y_elems, rows, K, group, res=None, norm=None, wgs=1, block=64 * WAVES, lds=0, mu=0.0,
           y_dtype=np.uint16, slot_a=None, slot_b=None):
    import metalchat_amd as mc

    k = acc.load(name)
    xb = x if hasattr(x, "device_ptr") else acc.to_device(x)
    yb = acc.alloc(y_elems * np.dtype(y_dtype).itemsize)
    yb.upload(np.zeros(y_elems, y_dtype))
    rb = res if (res is None or hasattr(res, "device_ptr")) else acc.to_device(res)
    nb = norm if (norm is None or hasattr(norm, "device_ptr")) else acc.to_device(norm)
    t = mc.KernelTask(k, (wgs * block, 1, 1), (block, 1, 1),
                      [acc.wrap(wptr, 1 << 40), (acc.wrap(sptr, 1 << 40) if sptr else None), xb, yb, rb, nb, np.uint32(rows),
                       np.uint32(K), np.uint32(group), np.float32(1e-5), np.float32(mu), slot_a, slot_b, np.uint32(0), np.float32(0)],
                      lds_bytes=lds)
    t()
    acc.wait()
    return yb.download(y_dtype, y_elems)


def dyadic_row(rng, n):
    """bf16 row of multiples of 1/4 in [-2, 2]: the sum of squares is exact in fp32 whatever the order of the additions,
    so the kernel's normalised row is the oracle's bit for bit."""
    return mo.encode(BF16, (rng.integers(-8, 9, n) / 4.0).astype(np.float32))


def oracle_rmsnorm(x, w, mu=0.0):
    n = x.size
    out = np.zeros((1, n), np.uint16)
    mo.rmsnorm(BF16, mo.layout((1, n)), out, mo.layout((1, n)), x.reshape(1, n), mo.layout((n,)), w, 1e-5, mu)
    return out.reshape(-1)


class Models:
    """decoders used as the host-side packer (mc_decoder_load_* -> the fused HBM layout) + the reference-native specs"""

    def __init__(self, acc):
        self.acc = acc
        self.made = {}

    def get(self, family, shape):
        """shape 'w2': a [256, K] matrix (long rows, few of them); 'wide': K = dim -- qkv [512, K], w13 [512, K], head [1000, K]"""
        import metalchat_amd as mc

        key = (family, shape)
        if key in self.made:
            return self.made[key]
        quant, fmt, K = FAMILIES[family]
        if shape == "w2":
            cfg = mg.tiny_cfg(BF16, dim=256, n_heads=2, n_kv_heads=1, head_dim=128, ffn_dim=K, n_layers=1, vocab=64, max_seq_len=16)
        else:
            cfg = mg.tiny_cfg(BF16, dim=K, n_heads=2, n_kv_heads=1, head_dim=128, ffn_dim=256, n_layers=1, vocab=1000, max_seq_len=16)
        w = mg.make_model(cfg, seed=sum(map(ord, family + shape)) % 1000, quant=quant, group=GROUP)
        dec = mc.Decoder(self.acc, **mg.decoder_kwargs(cfg, weight_format=fmt, group_size=GROUP if quant else 0))
        dec.load_model(w)
        self.made[key] = (cfg, w, dec)
        return self.made[key]

    def close(self):
        for _, _, d in self.made.values():
            d.release()


@pytest.fixture(scope="module")
def models(acc):
    m = Models(acc)
    yield m
    m.close()


def strict(got, ref, what, max_frac=0.01):
    r = parity.check(BF16, got, ref, rel=1e-3, max_ulp=1, max_frac=max_frac, scale_aware=False, what=what)
    return r


def grp(family):
    return GROUP if FAMILIES[family][0] else 0


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("family", list(FAMILIES))
def test_store_and_residual_epilogues_match_the_oracle(acc, models, family):
    cfg, w, dec = models.get(family, "w2")
    K = FAMILIES[family][2]
    spec = w["layers"][0]["w2"]
    rng = np.random.default_rng(11)
    x = mo.encode(BF16, rng.normal(0, 1, K).astype(np.float32))
    res = mo.encode(BF16, rng.normal(0, 1, 256).astype(np.float32))
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w2")
    assert (rows, inf) == (256, K)
    ref = oracle_linear(BF16, spec, x)
    ref_res = np.zeros((1, 256), np.uint16)
    mo.add(BF16, mo.layout((1, 256)), ref_res, mo.layout((1, 256)), res.reshape(1, -1), mo.layout((1, 256)), ref.reshape(1, -1))
    # 128 row pairs: one workgroup (16 pairs per wave), five (3.2 per wave: spans of 4 and 3), 16 (one per wave), 32 (the int8 /
    # bfloat kernels then take ONE ROW per wave, gemv.h `half`), 40 (waves without work)
    for wgs in (1, 5, 16, 32, 40):
        got = launch(acc, kname(family, 0, 0), wptr, sptr, x, rows, rows, K, grp(family), wgs=wgs, lds=lds_bytes(family, K))
        strict(got, ref, f"{family} p0_e0 wgs {wgs}")
        got = launch(acc, kname(family, 0, 1), wptr, sptr, x, rows, rows, K, grp(family), res=res, wgs=wgs, lds=lds_bytes(family, K))
        strict(got, ref_res.reshape(-1), f"{family} p0_e1 wgs {wgs}")


@pytest.mark.parametrize("family", list(FAMILIES))
def test_rmsnorm_prologue_and_activation_epilogues_match_the_oracle(acc, models, family):
    cfg, w, dec = models.get(family, "wide")
    K = FAMILIES[family][2]
    lw = w["layers"][0]
    rng = np.random.default_rng(12)
    x = dyadic_row(rng, K)
    nw = lw["ffn_norm"]
    xn = oracle_rmsnorm(x, nw)
    L = mo.layout
    g1 = oracle_linear(BF16, lw["w1"], xn).reshape(1, -1)
    g3 = oracle_linear(BF16, lw["w3"], xn).reshape(1, -1)
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w13")
    assert (rows, inf) == (512, K)
    # rmsnorm + store: rows (w1 j, w3 j) interleaved
    ref_rows = np.stack([g1.reshape(-1), g3.reshape(-1)], 1).reshape(-1)
    for wgs in (1, 7, 32):
        got = launch(acc, kname(family, 1, 0), wptr, sptr, x, rows, rows, K, grp(family), norm=nw, wgs=wgs, lds=lds_bytes(family, K))
        strict(got, ref_rows, f"{family} p1_e0 wgs {wgs}")
    for epi, act in ((2, mo.silu), (3, mo.gelu)):
        a = np.zeros_like(g1)
        act(BF16, L(g1.shape), a, L(g1.shape), g1)
        ref = np.zeros_like(g1)
        mo.hadamard(BF16, L(g1.shape), ref, L(g1.shape), a, L(g1.shape), g3)
        for wgs in (1, 7, 32):
            got = launch(acc, kname(family, 1, epi), wptr, sptr, x, rows // 2, rows, K, grp(family), norm=nw, wgs=wgs,
                         lds=lds_bytes(family, K))
            # act(T(a)) * T(b): a one-step difference in either factor moves the product by at most one more step
            parity.check(BF16, got, ref.reshape(-1), rel=2e-3, max_ulp=2, max_frac=0.03, scale_aware=False, what=f"{family} p1_e{epi} wgs {wgs}")


@pytest.mark.parametrize("family", [f for f in FAMILIES if f != "i4_lin3s"])
def test_partial_sum_prologue_is_the_reduce_launch(acc, models, family):
    """`_p3_`: x = four fp32 partial rows (mc_attn_pv_T's context ranges); the prologue adds them in range order and rounds
    ONCE to T -- mc_attn_pv_reduce_T (bmm.metal:80: one rounding of the fp32 sum) -- then the GEMV."""
    cfg, w, dec = models.get(family, "w2")
    K = FAMILIES[family][2]
    spec = w["layers"][0]["w2"]
    rng = np.random.default_rng(13)
    parts = rng.normal(0, 0.5, (4, K)).astype(np.float32)
    s = np.zeros(K, np.float32)
    for r in range(4):
        s = (s + parts[r]).astype(np.float32)
    x = mo.encode(BF16, s)
    res = mo.encode(BF16, rng.normal(0, 1, 256).astype(np.float32))
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w2")
    ref = oracle_linear(BF16, spec, x)
    ref_res = np.zeros((1, 256), np.uint16)
    mo.add(BF16, mo.layout((1, 256)), ref_res, mo.layout((1, 256)), res.reshape(1, -1), mo.layout((1, 256)), ref.reshape(1, -1))
    for wgs in (1, 16):
        got = launch(acc, kname(family, 3, 0), wptr, sptr, parts.reshape(-1), rows, rows, K, grp(family), wgs=wgs, lds=lds_bytes(family, K))
        strict(got, ref, f"{family} p3_e0 wgs {wgs}")
        got = launch(acc, kname(family, 3, 1), wptr, sptr, parts.reshape(-1), rows, rows, K, grp(family), res=res, wgs=wgs,
                     lds=lds_bytes(family, K))
        strict(got, ref_res.reshape(-1), f"{family} p3_e1 wgs {wgs}")
        # ... and it IS the plain kernel on the rounded sum, bit for bit
        plain = launch(acc, kname(family, 0, 1), wptr, sptr, x, rows, rows, K, grp(family), res=res, wgs=wgs, lds=lds_bytes(family, K))
        parity.exact(got, plain, f"{family} p3_e1 == p0_e1 on the reduced row")


def one_hot_columns(K, chunk):
    ks = set(range(0, min(K, chunk)))                     # every lane, dword and nibble / byte of the first chunk
    for c in range(chunk, K, chunk):                      # the edges of every other chunk
        ks.update(range(c - 24, c + 40))
    ks.update(range(K - 64, K))
    ks.update(range(chunk + 5, K, 211))
    return sorted(k for k in ks if 0 <= k < K)


@pytest.mark.parametrize("family", list(FAMILIES))
def test_one_hot_rows_return_every_weight_bit_for_bit(acc, models, family):
    cfg, w, dec = models.get(family, "w2")
    quant, fmt, K = FAMILIES[family]
    spec = w["layers"][0]["w2"]
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w2")
    # Wd = T(T(q) T(s)) once (kernel/mul.metal:78-82); a one-hot row times it is the column itself
    if quant:
        wd = np.zeros((rows, K), np.uint16)
        sc = np.ascontiguousarray(spec["scales"].reshape(-1), np.float32)
        mo.hadamard_broadcast(BF16, 1, mo.layout((rows * ng, GROUP)), wd, mo.layout((rows * ng, GROUP)), spec["weight"], mo.layout((rows * ng,)), sc)
    else:
        wd = spec["weight"]
    rows_used = 60 if family != "i4_lin3s" else 64     # 30 pairs over 8 waves: spans of 4 and 3 pairs (lin3s: whole quads)
    name = kname(family, 0, 0)
    chunk = {"i4": 2048, "i8": 1024, "w": 512}[family.split("_")[0]]
    ks = one_hot_columns(K, chunk)
    xb = acc.alloc(K * 2)
    lds = lds_bytes(family, K)
    for i, k in enumerate(ks):
        x = np.zeros(K, np.uint16)
        x[k] = 0x3F80  # 1.0
        xb.upload(x)
        got = launch(acc, name, wptr, sptr, xb, rows_used, rows_used, K, grp(family), wgs=1, lds=lds)
        if i < 3:  # the column really is what the oracle's bmm returns for this row
            xe = mo.encode(BF16, np.eye(1, K, k, dtype=np.float32).reshape(-1))
            parity.exact(oracle_linear(BF16, spec, xe)[:rows_used], wd[:rows_used, k], f"{family} oracle column {k}")
        assert np.array_equal(got, wd[:rows_used, k]), f"{family} column {k}: rows {np.flatnonzero(got != wd[:rows_used, k])[:8]}"
    # the negated one-hot too (sign of the products), a few columns
    for k in ks[:: max(1, len(ks) // 16)]:
        x = np.zeros(K, np.uint16)
        x[k] = 0xBF80
        xb.upload(x)
        got = launch(acc, name, wptr, sptr, xb, rows_used, rows_used, K, grp(family), wgs=1, lds=lds)
        assert np.array_equal(got, wd[:rows_used, k] ^ np.where((wd[:rows_used, k] & 0x7FFF) != 0, 0x8000, 0).astype(np.uint16)), f"{family} column -{k}"


@pytest.mark.parametrize("family", list(FAMILIES))
@pytest.mark.parametrize("pro", [0, 1])
def test_linear_order_equals_the_classic_kernel(acc, models, family, pro):
    """The per-weight arithmetic and the order of a row's additions are the classic kernel's (chunk after chunk per lane,
    the same wave reduction): bit identity, except where the grouping of the per-lane sums differs by construction --
    `_lin3s_` (one accumulator per packet of a super row) and int8 on the matrix pipe (mac8b_n vs v_dot2c)."""
    cfg, w, dec = models.get(family, "wide" if pro else "w2")
    K = FAMILIES[family][2]
    rng = np.random.default_rng(17)
    x = mo.encode(BF16, rng.normal(0, 1, K).astype(np.float32))
    which = "w13" if pro else "w2"
    nw = w["layers"][0]["ffn_norm"] if pro else None
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, which)
    if pro and K * 2 // 16 > 4 * 512:
        pytest.skip("the classic prologue leaves its register path for rows this long: another order of the sum of squares")
    lin = launch(acc, kname(family, pro, 0), wptr, sptr, x, rows, rows, K, grp(family), norm=nw, wgs=4, lds=lds_bytes(family, K))
    # the classic kernel with the same workgroup size (same threads per packet in the prologue's sum of squares)
    cls = launch(acc, classic_name(family, pro, 0), wptr, sptr, x, rows, rows, K, grp(family), norm=nw, wgs=4, block=512,
                 lds=lds_bytes(family, K, classic=True))
    if family in ("i4_lin3s", "i8_ling4", "i8_ling14"):
        r = parity.check(BF16, lin, cls, rel=1e-3, max_ulp=1, max_frac=0.01, scale_aware=False, what=f"{family} p{pro} vs classic")
    else:
        parity.exact(lin, cls, f"{family} p{pro}: linear order vs classic")


@pytest.mark.parametrize("family", [f for f in FAMILIES if f != "i4_lin3s"])
def test_greedy_pick_epilogue(acc, models, family):
    """`_e5` = `_e0` + the greedy pick of the stored row (transformer.h:357-364 with an argmax sampler): the logits are the
    store kernel's bit for bit, the workgroups' keys fold to the FIRST index of the maximum, in both pick modes."""
    cfg, w, dec = models.get(family, "wide")
    K = FAMILIES[family][2]
    rng = np.random.default_rng(19)
    x = dyadic_row(rng, K)
    nw = w["final_norm"]
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(-1, "output")
    assert (rows, inf) == (1000, K)
    lds = lds_bytes(family, K)
    wgs = 9
    plain = launch(acc, kname(family, 1, 0), wptr, sptr, x, rows, rows, K, grp(family), norm=nw, wgs=wgs, lds=lds)
    strict(plain, oracle_linear(BF16, w["output"], oracle_rmsnorm(x, nw)), f"{family} head")
    vals = mo.from_bf16(plain)
    want = int(np.flatnonzero(vals == vals.max())[0])
    # mode 2: one key per workgroup, folded by the host here (mc_argmax_keys on the device)
    keys = acc.alloc(8 * wgs)
    keys.upload(np.zeros(wgs, np.uint64))
    state = acc.to_device(np.zeros(12, np.int32))
    toks = acc.to_device(np.full(4, -1, np.int32))
    desc = acc.to_device(np.frombuffer(struct.pack("<QQQQ", keys.device_ptr, 0, state.device_ptr, toks.device_ptr), np.uint8))
    got = launch(acc, kname(family, 1, 5), wptr, sptr, x, rows, rows, K, grp(family), res=desc, norm=nw, wgs=wgs, lds=lds)
    parity.exact(got, plain, f"{family} e5 logits")
    k = int(keys.download(np.uint64, wgs).max())
    assert 0xFFFFFFFF - (k & 0xFFFFFFFF) == want, (family, want)
    # mode 1: atomic key + ticket, the last workgroup writes the token and clears both
    kt = acc.to_device(np.zeros(2, np.uint64))
    desc1 = acc.to_device(np.frombuffer(struct.pack("<QQQQ", kt.device_ptr, kt.device_ptr + 8, state.device_ptr, toks.device_ptr), np.uint8))
    got = launch(acc, kname(family, 1, 5), wptr, sptr, x, rows, rows, K, grp(family), res=desc1, norm=nw, wgs=wgs, lds=lds)
    parity.exact(got, plain, f"{family} e5 logits (ticket)")
    assert int(state.download(np.int32, 12)[0]) == want and int(toks.download(np.int32, 4)[0]) == want
    assert not kt.download(np.uint64, 2).any()
    # exact ties: the LOWER index wins -- the same row twice in the matrix is not available here, so tie the inputs
    # instead: a zero row gives 1000 equal logits (+0), the pick must be index 0
    z = np.zeros(K, np.uint16)
    keys.upload(np.zeros(wgs, np.uint64))
    launch(acc, kname(family, 1, 5), wptr, sptr, z, rows, rows, K, grp(family), res=desc, norm=nw, wgs=wgs, lds=lds)
    assert 0xFFFFFFFF - (int(keys.download(np.uint64, wgs).max()) & 0xFFFFFFFF) == 0


@pytest.mark.parametrize("family", list(FAMILIES))
def test_rope_and_cache_write_epilogue(acc, models, family):
    """`_e4` (kernel/rope.metal:49-59 + nn/cache.h:209-213 on the fused wq|wk|wv rows): exactly the rotation of the `_e0`
    kernel's rows -- which the test above pins against the oracle -- written where the decode attention reads them."""
    cfg, w, dec = models.get(family, "wide")
    K = FAMILIES[family][2]
    H, KV, hd, ms = 2, 1, 128, 16
    half = hd // 2
    rng = np.random.default_rng(23)
    x = dyadic_row(rng, K)
    nw = w["layers"][0]["attention_norm"]
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "qkv")
    assert (rows, inf) == ((H + 2 * KV) * hd, K)
    lds = lds_bytes(family, K)
    slot, rrow, nrows = 5, 3, 8
    fcos = np.cos(rng.uniform(0, 6.28, (nrows, half))).astype(np.float32)
    fsin = np.sin(rng.uniform(0, 6.28, (nrows, half))).astype(np.float32)
    for wgs in (1, 6):
        y = launch(acc, kname(family, 1, 0), wptr, sptr, x, rows, rows, K, grp(family), norm=nw, wgs=wgs, lds=lds)
        q_out = acc.to_device(np.zeros(H * hd, np.uint16))
        kc = acc.to_device(np.zeros(KV * ms * hd, np.uint16))
        vt = acc.to_device(np.zeros(KV * hd * ms, np.uint16))
        st = np.zeros(12, np.int32)
        st[3], st[6] = slot, rrow
        state = acc.to_device(st)
        cb, sb = acc.to_device(fcos.reshape(-1)), acc.to_device(fsin.reshape(-1))
        desc = acc.to_device(np.frombuffer(struct.pack("<QQQQQQIIII", q_out.device_ptr, kc.device_ptr, vt.device_ptr, cb.device_ptr,
                                                       sb.device_ptr, state.device_ptr, H, KV, hd, ms), np.uint8))
        launch(acc, kname(family, 1, 4), wptr, sptr, x, rows, rows, K, grp(family), res=desc, norm=nw, wgs=wgs, lds=lds)
        yf = mo.from_bf16(y)
        heads = yf[: (H + KV) * hd].reshape(H + KV, half, 2)      # packed: (2j, 2j + 1) = natural (j, j + hd/2)
        x1, x2 = heads[:, :, 0], heads[:, :, 1]
        c, s = fcos[rrow][None, :], fsin[rrow][None, :]
        o1 = mo.to_bf16((c * x1).astype(np.float32) - (s * x2).astype(np.float32))
        o2 = mo.to_bf16((s * x1).astype(np.float32) + (c * x2).astype(np.float32))
        rot = np.concatenate([o1, o2], 1)                          # natural order per head
        parity.exact(q_out.download(np.uint16, H * hd).reshape(H, hd), rot[:H], f"{family} e4 q wgs {wgs}")
        kgot = kc.download(np.uint16, KV * ms * hd).reshape(KV, ms, hd)
        parity.exact(kgot[:, slot], rot[H:], f"{family} e4 k row")
        assert not np.delete(kgot, slot, axis=1).any()
        vgot = vt.download(np.uint16, KV * hd * ms).reshape(KV * hd, ms)
        parity.exact(vgot[:, slot], y[(H + KV) * hd:], f"{family} e4 v row")
        assert not np.delete(vgot, slot, axis=1).any()


@pytest.mark.parametrize("family", ["i4_lin2", "i4_lin3s", "i4_lin1", "i4_lin4"])
def test_gemma_post_norm_prologue(acc, models, family):
    """`_p2_` (nn/transformer.h:132-139): h = T(res + rmsnorm_post(x)) left in HBM by workgroup 0, row = rmsnorm(h)."""
    cfg, w, dec = models.get(family, "wide")
    K = FAMILIES[family][2]
    lw = w["layers"][0]
    rng = np.random.default_rng(29)
    x = dyadic_row(rng, K)
    post_w = mo.encode(BF16, rng.uniform(-0.25, 0.25, K).astype(np.float32))
    nw = mo.encode(BF16, rng.uniform(-0.25, 0.25, K).astype(np.float32))
    res = mo.encode(BF16, rng.normal(0, 1, K).astype(np.float32))
    mu = 1.0
    L = mo.layout
    pn = oracle_rmsnorm(x, post_w, mu)
    h = np.zeros((1, K), np.uint16)
    mo.add(BF16, L((1, K)), h, L((1, K)), res.reshape(1, -1), L((1, K)), pn.reshape(1, -1))
    xn = oracle_rmsnorm(h.reshape(-1), nw, mu)
    g1 = oracle_linear(BF16, lw["w1"], xn).reshape(1, -1)
    g3 = oracle_linear(BF16, lw["w3"], xn).reshape(1, -1)
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w13")
    pwb, rsb = acc.to_device(post_w), acc.to_device(res)
    hb = acc.alloc(K * 2)
    # (the linear-order `_p2_` kernels take the three pointers as arguments: `res` = the residual row, the adaptor slots = the
    #  post-norm weight and h_out -- gemv.h; decoder.cc gemv() passes them the same way)
    pn = dict(res=rsb, slot_a=pwb, slot_b=hb)
    got = launch(acc, kname(family, 2, 0), wptr, sptr, x, rows, rows, K, grp(family), norm=nw, wgs=5, lds=lds_bytes(family, K), mu=mu, **pn)
    parity.exact(hb.download(np.uint16, K), h.reshape(-1), f"{family} p2 hidden row")  # (the first norm is exact: dyadic row)
    ref_rows = np.stack([g1.reshape(-1), g3.reshape(-1)], 1).reshape(-1)
    # the second sum of squares is an ordinary one (another order than the oracle's): composition bound
    parity.check(BF16, got, ref_rows, rel=2e-3, max_ulp=2, max_frac=0.2, what=f"{family} p2_e0")
    a = np.zeros_like(g1)
    mo.gelu(BF16, L(g1.shape), a, L(g1.shape), g1)
    ref = np.zeros_like(g1)
    mo.hadamard(BF16, L(g1.shape), ref, L(g1.shape), a, L(g1.shape), g3)
    hb.upload(np.zeros(K, np.uint16))
    got = launch(acc, kname(family, 2, 3), wptr, sptr, x, rows // 2, rows, K, grp(family), norm=nw, wgs=5, lds=lds_bytes(family, K), mu=mu, **pn)
    parity.exact(hb.download(np.uint16, K), h.reshape(-1), f"{family} p2_e3 hidden row")
    parity.check(BF16, got, ref.reshape(-1), rel=3e-3, max_ulp=2, max_frac=0.3, what=f"{family} p2_e3")


# ---------------------------------------------------------------------------------------------------------------------
# Round 6: long rows, few of them -- the K range of a row pair over four waves of a workgroup (gemv_ksplit.h, mc_gemv_i4_bfloat_lin12k4_p0_e{0,1}:
# Gemma-7B's w2).  The per-weight arithmetic is the linear-order kernels'; a row's fp32 sum is four chains of three chunks added in quarter
# order instead of one chain of twelve: the bound of `_lin3s_` against the classic family (one bf16 step, <= 1 % of the outputs).
K4 = "i4_lin12k4"


def k4_lds():
    return lds_bytes("i4_lin12", 24576)


def test_ksplit_store_and_residual_epilogues_match_the_oracle(acc, models):
    cfg, w, dec = models.get("i4_lin12", "w2")
    K = 24576
    spec = w["layers"][0]["w2"]
    rng = np.random.default_rng(21)
    x = mo.encode(BF16, rng.normal(0, 1, K).astype(np.float32))
    res = mo.encode(BF16, rng.normal(0, 1, 256).astype(np.float32))
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w2")
    ref = oracle_linear(BF16, spec, x)
    ref_res = np.zeros((1, 256), np.uint16)
    mo.add(BF16, mo.layout((1, 256)), ref_res, mo.layout((1, 256)), res.reshape(1, -1), mo.layout((1, 256)), ref.reshape(1, -1))
    lin = launch(acc, kname("i4_lin12", 0, 0), wptr, sptr, x, rows, rows, K, GROUP, wgs=16, lds=k4_lds())
    # 128 row pairs: 16 workgroups (8 pairs each: four per half), 21 (7 and 6: halves of 4 + 3 and 3 + 3), 32 (4), 40 (4 and 3), 128 (one: half 1 idle)
    for wgs in (16, 21, 32, 40, 128):
        got = launch(acc, kname(K4, 0, 0), wptr, sptr, x, rows, rows, K, GROUP, wgs=wgs, lds=k4_lds())
        strict(got, ref, f"{K4} p0_e0 wgs {wgs}")
        r = parity.check(BF16, got, lin, rel=1e-3, max_ulp=1, max_frac=0.01, scale_aware=False, what=f"{K4} vs lin12, wgs {wgs}")
        got = launch(acc, kname(K4, 0, 1), wptr, sptr, x, rows, rows, K, GROUP, res=res, wgs=wgs, lds=k4_lds())
        strict(got, ref_res.reshape(-1), f"{K4} p0_e1 wgs {wgs}")


def test_ksplit_one_hot_rows_return_every_weight_bit_for_bit(acc, models):
    cfg, w, dec = models.get("i4_lin12", "w2")
    K = 24576
    spec = w["layers"][0]["w2"]
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w2")
    wd = np.zeros((rows, K), np.uint16)
    sc = np.ascontiguousarray(spec["scales"].reshape(-1), np.float32)
    mo.hadamard_broadcast(BF16, 1, mo.layout((rows * ng, GROUP)), wd, mo.layout((rows * ng, GROUP)), spec["weight"], mo.layout((rows * ng,)), sc)
    rows_used = 60   # 30 pairs over 4 workgroups: 8, 8, 7, 7 -- halves of 4 + 4 and 4 + 3
    ks = one_hot_columns(K, 2048)
    xb = acc.alloc(K * 2)
    for k in ks[::3] + [6143, 6144, 12287, 12288, 18431, 18432]:   # (every third column of the sweep, and the quarters' edges)
        x = np.zeros(K, np.uint16)
        x[k] = 0x3F80  # 1.0: three of a row's four quarter sums are exact zeros
        xb.upload(x)
        got = launch(acc, kname(K4, 0, 0), wptr, sptr, xb, rows_used, rows_used, K, GROUP, wgs=4, lds=k4_lds())
        assert np.array_equal(got, wd[:rows_used, k]), f"{K4} column {k}: rows {np.flatnonzero(got != wd[:rows_used, k])[:8]}"
