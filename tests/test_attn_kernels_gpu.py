"""Kernel-level parity of the DECODE ATTENTION kernels -- the one-launch form `mc_attn_fused_bfloat` (scores, softmax, P.V with
two in-launch hand-offs) and the two-launch form `mc_attn_scores_bfloat` + `mc_attn_pv_bfloat` -- launched BY NAME through the
Part-1 seam on a given query row and cache, against the oracle's kernels composed as nn::attention::operator() composes them
(include/metalchat/nn/attention.h:181-203): repeat_kv, bmm -> T, scalar_mul in T, softmax WITHOUT max shift -> T
(kernel/softmax.metal:44-47), bmm -> T.  Only the order of the fp32 sums differs (the softmax denominator and the P.V
contraction are added range by range), so the bound is the single-kernel one: every output within one bf16 step of the oracle
-- one step at the row's scale for outputs that are small through cancellation (a 1000-term sum of products of either sign).

Cases: Llama-3-8B (4 query heads per kv head, head_dim 128), TinyLlama (8 per kv head, head_dim 64), Gemma-7B shapes (MHA,
head_dim 256), head_dim 32; caches that are full, nearly empty (most ranges of the launch have nothing to publish), and that end
in the middle of a 64-slot range / 16-slot tile."""
import numpy as np
import pytest

import modelgen as mg
import parity
from oracle import mc_oracle as mo
from test_gemv_gpu import oracle_linear

pytestmark = pytest.mark.gpu
BF16 = 0
PB = 64


def oracle_attention(q, k, v, n_rep, scale):
    """q [H, hd], k / v [n, KV, hd] (bf16 bits) -> [H, hd] bf16 bits"""
    L = mo.layout
    H, hd = q.shape
    n = k.shape[0]
    krep = np.ascontiguousarray(np.repeat(k.transpose(1, 0, 2), n_rep, axis=0))   # [H, n, hd]  (functional/transform.h:20-90)
    vrep = np.ascontiguousarray(np.repeat(v.transpose(1, 0, 2), n_rep, axis=0))
    s = np.zeros((H, 1, n), np.uint16)
    mo.bmm(BF16, L(s.shape), s, L((H, 1, hd)), q.reshape(H, 1, hd), L((H, hd, n), strides=(n * hd, 1, hd)), krep)
    s2 = np.zeros_like(s)
    mo.scalar_mul(BF16, L((H, n)), s2, L((H, n)), s.reshape(H, n), scale)
    p = np.zeros((H, n), np.uint16)
    mo.softmax(BF16, L((H, n)), p, L((H, n)), s2.reshape(H, n))
    o = np.zeros((H, 1, hd), np.uint16)
    mo.bmm(BF16, L(o.shape), o, L((H, 1, n)), p.reshape(H, 1, n), L((H, n, hd)), vrep)
    return o.reshape(H, hd)


def device_caches(acc, k, v, max_seq):
    n, KV, hd = k.shape
    kc = np.zeros((KV, max_seq, hd), np.uint16)
    vt = np.zeros((KV, hd, max_seq), np.uint16)
    kc[:, :n] = k.transpose(1, 0, 2)
    vt[:, :, :n] = v.transpose(1, 2, 0)
    # slots past kv_len hold garbage of an earlier conversation: they must not matter
    rng = np.random.default_rng(9)
    kc[:, n:] = mo.encode(BF16, rng.normal(0, 30, (KV, max_seq - n, hd)).astype(np.float32))
    vt[:, :, n:] = mo.encode(BF16, rng.normal(0, 30, (KV, hd, max_seq - n)).astype(np.float32))
    return acc.to_device(kc.reshape(-1)), acc.to_device(vt.reshape(-1))


def state_buffer(acc, kv_len, epoch):
    st = np.zeros(12, np.int32)
    st[2], st[9] = kv_len, epoch
    return acc.to_device(st)


CASES = [
    # H, KV, hd, max_seq, kv_len
    (32, 8, 128, 2048, 2048), (32, 8, 128, 2048, 2047), (32, 8, 128, 2048, 1), (32, 8, 128, 2048, 65), (32, 8, 128, 2048, 1000),
    (32, 4, 64, 2048, 2048), (32, 4, 64, 2048, 77),
    (16, 16, 256, 2048, 2048), (16, 16, 256, 2048, 130),
    (8, 2, 32, 512, 500), (16, 1, 128, 1024, 1024),
]
# hand-offs with / without the XCD-local fast path (handoff.h); 3: with it AND the kv heads dealt with a stride of the next multiple
# of 8 (grid = nsplit x stride, the workgroups without a head leave at once: decoder.cc handoff_mode_alone)
@pytest.mark.parametrize("fast", [1, 0, 3])
@pytest.mark.parametrize("tiles", [1, 2])   # 64-slot ranges (mc_attn_fused_bfloat) / 128-slot ranges (mc_attn_fused_t2_bfloat: S = 8192 in the decoder)
@pytest.mark.parametrize("H,KV,hd,max_seq,n", CASES + [(32, 8, 128, 8192, 8000)])
def test_one_launch_attention_matches_the_oracle(acc, H, KV, hd, max_seq, n, fast, tiles):
    import metalchat_amd as mc

    if tiles == 2 and (hd not in (64, 128) or max_seq % (2 * PB)):
        pytest.skip("128-slot ranges are built for head_dim 64 and 128")
    rng = np.random.default_rng(H * 7 + hd + n)
    n_rep, nsplit = H // KV, (max_seq + PB * tiles - 1) // (PB * tiles)
    q = mo.encode(BF16, rng.normal(0, 1, (H, hd)).astype(np.float32))
    k = mo.encode(BF16, rng.normal(0, 0.4, (n, KV, hd)).astype(np.float32))
    v = mo.encode(BF16, rng.normal(0, 0.5, (n, KV, hd)).astype(np.float32))
    scale = float(mo.from_bf16(mo.to_bf16(np.array([hd ** -0.5], np.float32)))[0])
    ref = oracle_attention(q, k, v, n_rep, scale)
    kc, vt = device_caches(acc, k, v, max_seq)
    qb = acc.to_device(q.reshape(-1))
    out = acc.alloc(H * hd * 2)
    psum = acc.to_device(np.zeros(2 * H * nsplit, np.uint64))   # (slow words, then the XCD-local fast words: handoff.h)
    slab = acc.to_device(np.zeros(2 * H * hd * nsplit, np.uint64))
    kern = acc.load("mc_attn_fused_bfloat" if tiles == 1 else "mc_attn_fused_t2_bfloat")
    # several launches over the same granule buffers with the tags consecutive launches of a token (and consecutive tokens)
    # carry: a granule of an earlier launch must never be taken for this one's
    for epoch, layer_tag in ((1, 1), (1, 2), (2, 1), (7, 255)):
        out.upload(np.zeros(H * hd, np.uint16))
        state = state_buffer(acc, n, epoch)
        stride = (KV + 7) // 8 * 8 if fast & 2 else KV
        mc.KernelTask(kern, (nsplit * stride * 256, 1, 1), (256, 1, 1),
                      [qb, kc, vt, out, psum, slab, state, np.uint32(n_rep), np.uint32(KV), np.uint32(hd), np.uint32(max_seq),
                       np.float32(scale), np.uint32(nsplit), np.uint32(layer_tag), None, np.uint32(fast)])()
        acc.wait()
        assert int(state.download(np.int32, 12)[10]) == 0, "a hand-off of the launch gave up"
        got = out.download(np.uint16, H * hd)
        parity.check(BF16, got, ref, rel=2e-3, max_ulp=1, max_frac=0.03, scale_aware=True,
                     what=f"one-launch attention H{H} KV{KV} hd{hd} n{n} tag ({epoch}, {layer_tag}) mode {fast}")


@pytest.mark.parametrize("H,KV,hd,max_seq,n", CASES)
def test_two_launch_attention_matches_the_oracle(acc, H, KV, hd, max_seq, n):
    import metalchat_amd as mc

    rng = np.random.default_rng(H * 7 + hd + n)
    n_rep, nsplit = H // KV, (max_seq + PB - 1) // PB
    q = mo.encode(BF16, rng.normal(0, 1, (H, hd)).astype(np.float32))
    k = mo.encode(BF16, rng.normal(0, 0.4, (n, KV, hd)).astype(np.float32))
    v = mo.encode(BF16, rng.normal(0, 0.5, (n, KV, hd)).astype(np.float32))
    scale = float(mo.from_bf16(mo.to_bf16(np.array([hd ** -0.5], np.float32)))[0])
    ref = oracle_attention(q, k, v, n_rep, scale)
    kc, vt = device_caches(acc, k, v, max_seq)
    qb = acc.to_device(q.reshape(-1))
    state = state_buffer(acc, n, 1)
    expv = acc.alloc(H * max_seq * 4)
    psum = acc.alloc(H * nsplit * 4)
    for ranges, block in ((1, 1024), (4, 256)):
        out = acc.to_device(np.zeros(H * hd, np.uint16))
        parts = acc.alloc(ranges * H * hd * 4)
        mc.KernelTask(acc.load("mc_attn_scores_bfloat"), (nsplit * 256, KV, 1), (256, 1, 1),
                      [qb, kc, expv, psum, None, state, np.uint32(n_rep), np.uint32(hd), np.uint32(max_seq), np.float32(scale),
                       np.uint32(nsplit)])()
        mc.KernelTask(acc.load("mc_attn_pv_bfloat"), (hd // 16 * block, KV, ranges), (block, 1, 1),
                      [expv, psum, vt, out, state, np.uint32(n_rep), np.uint32(hd), np.uint32(max_seq), np.uint32(nsplit), parts,
                       np.uint32(H)])()
        if ranges > 1:
            mc.KernelTask(acc.load("mc_attn_pv_reduce_bfloat"), ((H * hd + 255) // 256 * 256, 1, 1), (256, 1, 1),
                          [parts, out, np.uint32(H * hd), np.uint32(ranges)])()
        acc.wait()
        got = out.download(np.uint16, H * hd)
        parity.check(BF16, got, ref, rel=2e-3, max_ulp=1, max_frac=0.03, scale_aware=True,
                     what=f"two-launch attention H{H} KV{KV} hd{hd} n{n} ranges {ranges}")


def test_a_hand_off_whose_producers_never_run_gives_up_and_reports(acc):
    """The waits inside a launch are bounded (handoff.h: 50 ms of s_memrealtime, then state.err is set and every later wait of the
    token returns at once).  Here half of the workgroups of mc_attn_fused_bfloat are simply not launched: the others wait for
    partial denominators nobody will publish, must come back within the bound with the error word set (what
    mc_decoder_step / _generate turn into MC_ERR_RUNTIME, decoder.cc check_handoffs) -- and the same buffers serve a complete
    launch of the next step correctly."""
    import time

    import metalchat_amd as mc

    H, KV, hd, max_seq, n = 32, 8, 128, 2048, 2048
    rng = np.random.default_rng(5)
    n_rep, nsplit = H // KV, max_seq // PB
    q = mo.encode(BF16, rng.normal(0, 1, (H, hd)).astype(np.float32))
    k = mo.encode(BF16, rng.normal(0, 0.4, (n, KV, hd)).astype(np.float32))
    v = mo.encode(BF16, rng.normal(0, 0.5, (n, KV, hd)).astype(np.float32))
    scale = float(mo.from_bf16(mo.to_bf16(np.array([hd ** -0.5], np.float32)))[0])
    kc, vt = device_caches(acc, k, v, max_seq)
    qb = acc.to_device(q.reshape(-1))
    out = acc.alloc(H * hd * 2)
    psum = acc.to_device(np.zeros(2 * H * nsplit, np.uint64))   # (slow words, then the XCD-local fast words: handoff.h)
    slab = acc.to_device(np.zeros(2 * H * hd * nsplit, np.uint64))
    kern = acc.load("mc_attn_fused_bfloat")

    def launch(state, wgs):
        mc.KernelTask(kern, (wgs * 256, 1, 1), (256, 1, 1),
                      [qb, kc, vt, out, psum, slab, state, np.uint32(n_rep), np.uint32(KV), np.uint32(hd), np.uint32(max_seq),
                       np.float32(scale), np.uint32(nsplit), np.uint32(1), None, np.uint32(1)])()
        acc.wait()

    state = state_buffer(acc, n, 1)
    t0 = time.perf_counter()
    launch(state, nsplit * KV // 2)          # the ranges 16..31 of every kv head are missing
    took = time.perf_counter() - t0
    err = int(state.download(np.int32, 12)[10]) & 0xFFFFFFFF
    assert err != 0, "the launch must report the hand-off it gave up on"
    assert 0.04 < took < 0.4, f"bounded wait: {took:.3f} s"
    # the next step (a new epoch): complete grid, same granule buffers, a clean state
    state = state_buffer(acc, n, 2)
    launch(state, nsplit * KV)
    assert int(state.download(np.int32, 12)[10]) == 0
    got = out.download(np.uint16, H * hd)
    parity.check(BF16, got, oracle_attention(q, k, v, n_rep, scale), rel=2e-3, max_ulp=1, max_frac=0.03, scale_aware=True,
                 what="one-launch attention after a launch that gave up")


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("shape,n", [("llama3-8b", 2048), ("llama3-8b", 1000), ("llama3-8b", 3), ("hd64", 2047), ("llama3-8b-1024", 1024)])
def test_attention_and_wo_in_one_launch_matches_the_oracle(acc, shape, n, fast):
    """`mc_attn_wo_i4_bfloat_*` (attn_block_kernels.hip: the decode attention AND the Wo GEMV + residual of one block,
    nn/attention.h:191-205 + nn/transformer.h:132-133) launched BY NAME on a query row, a cache and a hidden row of its own, against
    the oracle's kernels composed as the reference composes them: the attention above, hadamard_broadcast + bmm for Wo
    (quantization/lora.h:94-122), add in T (kernel/arithmetic.metal:13-46).  The attention row may differ from the oracle's in the
    last place of a few elements (the order of the fp32 sums): the composition bound of the suite, two steps at the row's scale.
    The attention row the launch leaves in HBM is held to the single-kernel bound."""
    import metalchat_amd as mc

    if shape == "hd64":
        H, KV, hd, dim, max_seq = 32, 8, 64, 2048, 2048
    elif shape == "llama3-8b-1024":   # 128 workgroups: two row pairs of Wo per wave
        H, KV, hd, dim, max_seq = 32, 8, 128, 4096, 1024
    else:
        H, KV, hd, dim, max_seq = 32, 8, 128, 4096, 2048
    cfg = mg.tiny_cfg(BF16, dim=dim, n_heads=H, n_kv_heads=KV, head_dim=hd, ffn_dim=256, n_layers=1, vocab=64, max_seq_len=max_seq)
    w = mg.make_model(cfg, seed=77 + hd, quant="i4", group=128)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=128))
    dec.load_model(w)
    wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "wo")
    assert (rows, inf) == (dim, H * hd)
    rng = np.random.default_rng(H + hd + n)
    n_rep, nsplit = H // KV, (max_seq + PB - 1) // PB
    q = mo.encode(BF16, rng.normal(0, 1, (H, hd)).astype(np.float32))
    k = mo.encode(BF16, rng.normal(0, 0.4, (n, KV, hd)).astype(np.float32))
    v = mo.encode(BF16, rng.normal(0, 0.5, (n, KV, hd)).astype(np.float32))
    hidden = mo.encode(BF16, rng.normal(0, 1, dim).astype(np.float32))
    scale = float(mo.from_bf16(mo.to_bf16(np.array([hd ** -0.5], np.float32)))[0])
    att = oracle_attention(q, k, v, n_rep, scale)
    proj = oracle_linear(BF16, w["layers"][0]["wo"], att.reshape(1, 1, -1))
    L = mo.layout
    ref = np.zeros((1, dim), np.uint16)
    mo.add(BF16, L((1, dim)), ref, L((1, dim)), hidden.reshape(1, -1), L((1, dim)), proj.reshape(1, -1))
    kc, vt = device_caches(acc, k, v, max_seq)
    qb = acc.to_device(q.reshape(-1))
    psum = acc.to_device(np.zeros(2 * H * nsplit, np.uint64))   # (slow words, then the XCD-local fast words: handoff.h)
    slab = acc.to_device(np.zeros(2 * H * hd * nsplit, np.uint64))
    row_g = acc.to_device(np.zeros(H * hd // 2, np.uint64))
    attn_out = acc.alloc(H * hd * 2)
    kern = acc.load(f"mc_attn_wo_i4_bfloat_hd{hd}_k{H * hd // 2048}")
    W = lambda p_: acc.wrap(p_, 1 << 40)
    for epoch, layer_tag in ((1, 1), (1, 2), (9, 200)):   # consecutive launches of a token, a later token: the same granule buffers
        hb = acc.to_device(hidden)                        # read as the residual and overwritten IN PLACE, as the decoder launches it
        attn_out.upload(np.zeros(H * hd, np.uint16))
        state = state_buffer(acc, n, epoch)
        mc.KernelTask(kern, (nsplit * KV * 512, 1, 1), (512, 1, 1),
                      [qb, kc, vt, attn_out, psum, slab, row_g, state, np.uint32(n_rep), np.uint32(KV), np.uint32(max_seq), np.float32(scale),
                       np.uint32(nsplit), np.uint32(layer_tag), W(wptr), W(sptr), hb, hb, np.uint32(dim), np.uint32(128), np.uint32(1),
                       np.uint32(fast), None])()
        acc.wait()
        assert int(state.download(np.int32, 12)[10]) == 0, "a hand-off of the launch gave up"
        parity.check(BF16, attn_out.download(np.uint16, H * hd), att, rel=2e-3, max_ulp=1, max_frac=0.03, scale_aware=True,
                     what=f"attention row of the one launch, {shape} n{n} tag ({epoch}, {layer_tag})")
        parity.check(BF16, hb.download(np.uint16, dim), ref.reshape(-1), rel=4e-3, max_ulp=2, max_frac=0.3, scale_aware=True,
                     what=f"attention + Wo + residual in one launch, {shape} n{n} tag ({epoch}, {layer_tag})")
    dec.release()


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("n", [2048, 1000, 1, 65])
@pytest.mark.parametrize("shape", ["llama3-8b-int4", "llama3.2-1b-bf16", "tinyllama-bf16", "llama3-8b-int8", "llama3-8b-int8-8192", "llama3-8b-int4-4096",
                                   "llama3-8b-int4-8192", "llama3-8b-int8-4096", "llama3.2-1b-bf16-8192", "tinyllama-bf16-4096"])
def test_norm_qkv_rope_attention_and_wo_in_one_launch_matches_the_oracle(acc, n, fast, shape):
    """`mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2` (attn_block_kernels.hip qkv_in_launch: attention_norm, wq|wk|wv, RoPE, the cache
    write, the decode attention, Wo and the residual of one block -- nn/transformer.h:130-133, nn/attention.h:170-205 -- the kernel
    the benchmark's token launches 32 times) launched BY NAME on a hidden row and a cache of its own, against the oracle's kernels
    composed as the reference composes them: rmsnorm, three linears (hadamard_broadcast + bmm), rope at the step's table row, the
    rows appended to the cache (nn/cache.h:209-213), the attention of test_one_launch_attention_matches_the_oracle, Wo, add in T.
    The hidden row is dyadic, so the normalised row is the oracle's bit for bit (test_lin_kernels_gpu.py) and the cache rows the
    launch writes are held to the GEMV's single-kernel bound; the block's output to the composition bound of the suite."""
    import metalchat_amd as mc
    from test_lin_kernels_gpu import dyadic_row, oracle_rmsnorm

    # (the second shape: `mc_attn_qkv_wo_w_bfloat_hd64_k4_q4`, the same launch for PLAIN bfloat weights -- nn::linear, Llama-3.2-1B,
    #  the reference's default model, src/llama.cc:19-31)
    # (the third: TinyLlama's 4 kv heads x 8 query heads, launched as 8 VIRTUAL kv heads of 4 query heads -- kv_shift = 1, round 5,
    #  decode_kernels.hip attn_fused_bf: two virtual heads read one cache head and write its new row, the same bits, twice)
    # (the fourth and fifth: int8 weights, round 5 -- `mc_attn_qkv_wo_i8_bfloat_hd128_k4_q4_t1` with 64-slot ranges and `_t4` with 256-slot
    #  ranges at max_seq_len 8192, where the cache is filled up to n x 4 rows so that ranges are full, ragged and empty as in the others)
    # (the sixth and seventh: the int4 launch with 128- and 256-slot ranges, `_t2` / `_t4`, round 5 -- contexts of 4096 and 8192 slots)
    int4 = shape.startswith("llama3-8b-int4")
    int8 = shape.startswith("llama3-8b-int8")
    tiles = 4 if shape.endswith("-8192") else (2 if shape.endswith("-4096") else 1)
    # (... and the wide ranges of the int8 launch at S = 4096 and of the plain-bfloat launch: Llama-3.2-1B at 8192, TinyLlama's virtual heads at 4096)
    H, KV, hd, dim, max_seq = (32, 8, 128, 4096, 2048 * tiles) if (int4 or int8) else (32, 4 if shape.startswith("tinyllama-bf16") else 8, 64, 2048, 2048 * tiles)
    n = n * tiles
    vsh = 1 if shape.startswith("tinyllama-bf16") else 0
    KVV = KV << vsh
    half = hd // 2
    cfg = mg.tiny_cfg(BF16, dim=dim, n_heads=H, n_kv_heads=KV, head_dim=hd, ffn_dim=256, n_layers=1, vocab=64, max_seq_len=max_seq)
    w = mg.make_model(cfg, seed=303, quant="i4" if int4 else ("i8" if int8 else None), group=128)
    dec = mc.Decoder(acc, **(mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4 if int4 else mc.WFMT_I8, group_size=128) if (int4 or int8) else mg.decoder_kwargs(cfg)))
    dec.load_model(w)
    lw = w["layers"][0]
    wo_p, wo_s, rows, inf, _ = dec.weight_ptrs(0, "wo")
    qk_p, qk_s, qrows, qinf, _ = dec.weight_ptrs(0, "qkv")
    assert (rows, inf, qrows, qinf) == (dim, H * hd, (H + 2 * KV) * hd, dim)
    rng = np.random.default_rng(1000 + n)
    n_rep, nsplit = H // KV, max_seq // (PB * tiles)
    slot, rrow, nrows = n - 1, 5, 8            # the step writes slot n - 1 and reads n slots
    x = dyadic_row(rng, dim)
    k = mo.encode(BF16, rng.normal(0, 0.4, (n, KV, hd)).astype(np.float32))
    v = mo.encode(BF16, rng.normal(0, 0.5, (n, KV, hd)).astype(np.float32))
    fcos = np.zeros((nrows, half), np.float32)
    fsin = np.zeros((nrows, half), np.float32)
    L = mo.layout
    mo.rope_freqs(L(fcos.shape), fcos, L(fsin.shape), fsin, hd, 0, 500000.0)
    scale = float(mo.from_bf16(mo.to_bf16(np.array([hd ** -0.5], np.float32)))[0])
    # ---- the oracle's block
    xn = oracle_rmsnorm(x, lw["attention_norm"])
    q0 = oracle_linear(BF16, lw["wq"], xn.reshape(1, 1, -1)).reshape(H, hd)
    k0 = oracle_linear(BF16, lw["wk"], xn.reshape(1, 1, -1)).reshape(KV, hd)
    v0 = oracle_linear(BF16, lw["wv"], xn.reshape(1, 1, -1)).reshape(KV, hd)
    q1, k1 = np.zeros_like(q0), np.zeros_like(k0)
    mo.rope(BF16, L(q0.shape), q1, L(q0.shape), q0, L(fcos.shape), fcos, L(fsin.shape), fsin, 1, H, rrow)
    mo.rope(BF16, L(k0.shape), k1, L(k0.shape), k0, L(fcos.shape), fcos, L(fsin.shape), fsin, 1, KV, rrow)
    k[slot], v[slot] = k1, v0
    att = oracle_attention(q1, k, v, n_rep, scale)
    proj = oracle_linear(BF16, lw["wo"], att.reshape(1, 1, -1))
    ref = np.zeros((1, dim), np.uint16)
    mo.add(BF16, L((1, dim)), ref, L((1, dim)), x.reshape(1, -1), L((1, dim)), proj.reshape(1, -1))
    # ---- the launch: the cache holds the n - 1 earlier rows, garbage in the step's slot and behind it
    kpast, vpast = k.copy(), v.copy()
    kpast[slot] = mo.encode(BF16, rng.normal(0, 30, (KV, hd)).astype(np.float32))
    vpast[slot] = mo.encode(BF16, rng.normal(0, 30, (KV, hd)).astype(np.float32))
    kc, vt = device_caches(acc, kpast, vpast, max_seq)
    psum = acc.to_device(np.zeros(2 * H * nsplit, np.uint64))
    slab = acc.to_device(np.zeros(2 * H * hd * nsplit, np.uint64))
    row_g = acc.to_device(np.zeros(H * hd // 2, np.uint64))
    qkv_g = acc.to_device(np.zeros(2 * (H + 2 * KVV) * hd // 2, np.uint64))
    attn_out = acc.alloc(H * hd * 2)
    nw = acc.to_device(lw["attention_norm"])
    cb, sb = acc.to_device(fcos.reshape(-1)), acc.to_device(fsin.reshape(-1))
    kern = acc.load("mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2" + (f"_t{tiles}" if tiles > 1 else "") if int4 else (f"mc_attn_qkv_wo_i8_bfloat_hd128_k4_q4_t{tiles}" if int8 else "mc_attn_qkv_wo_w_bfloat_hd64_k4_q4" + (f"_t{tiles}" if tiles > 1 else "")))
    W = lambda p_: acc.wrap(p_, 1 << 40) if p_ else None
    for epoch, layer_tag in ((1, 1), (1, 2), (9, 200)):
        hb = acc.to_device(x)                  # read as the block input and the residual, overwritten IN PLACE, as the decoder launches it
        attn_out.upload(np.zeros(H * hd, np.uint16))
        st = np.zeros(12, np.int32)
        st[2], st[3], st[6], st[9] = n, slot, rrow, epoch
        state = acc.to_device(st)
        mc.KernelTask(kern, (nsplit * KVV * 512, 1, 1), (512, 1, 1),
                      [kc, vt, attn_out, psum, slab, row_g, qkv_g, state, np.uint32(n_rep >> vsh), np.uint32(KVV), np.uint32(max_seq),
                       np.float32(scale), np.uint32(nsplit), np.uint32(layer_tag), W(wo_p), W(wo_s), hb, hb, np.uint32(dim),
                       np.uint32(128 if (int4 or int8) else 0), nw, W(qk_p), W(qk_s), cb, sb, np.float32(1e-5), np.float32(0.0), np.uint32(fast), None,
                       np.uint32(vsh)])()
        acc.wait()
        assert int(state.download(np.int32, 12)[10]) == 0, "a hand-off of the launch gave up"
        kgot = kc.download(np.uint16, KV * max_seq * hd).reshape(KV, max_seq, hd)
        vgot = vt.download(np.uint16, KV * hd * max_seq).reshape(KV, hd, max_seq)
        parity.check(BF16, kgot[:, slot].reshape(-1), k1.reshape(-1), rel=3e-3, max_ulp=2, max_frac=0.2, what=f"K row the launch wrote (GEMV + rotation), n{n}")
        parity.check(BF16, vgot[:, :, slot].reshape(-1), v0.reshape(-1), rel=2e-3, max_ulp=1, max_frac=0.05, what=f"V row the launch wrote, n{n}")
        if n > 1:
            parity.exact(kgot[:, : slot], kpast[: slot].transpose(1, 0, 2), "the earlier K rows are untouched")
        parity.check(BF16, attn_out.download(np.uint16, H * hd), att.reshape(-1), rel=4e-3, max_ulp=2, max_frac=0.3, scale_aware=True,
                     what=f"attention row of the one launch, n{n} tag ({epoch}, {layer_tag})")
        parity.check(BF16, hb.download(np.uint16, dim), ref.reshape(-1), rel=4e-3, max_ulp=2, max_frac=0.3, scale_aware=True,
                     what=f"norm + wq|wk|wv + rope + attention + Wo + residual in one launch, n{n} tag ({epoch}, {layer_tag})")
    dec.release()


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("tiles,n", [(2, 2048), (2, 1000), (2, 1), (2, 129), (2, 40), (2, 200), (1, 1024), (1, 65), (4, 4096), (4, 2900)])   # (the step's slot in every 32-slot group of a range)
def test_gemma_norms_rope_attention_and_wo_in_one_launch_matches_the_oracle(acc, tiles, n, fast):
    """`mc_attn_wo_qkn_i4_bfloat_hd256_k2_t{1,2}` (round 5; attn_block_kernels.hip attn_wo_body with decode_kernels.hip q_from_qkv_rows<256, 512>):
    gemma3's q_norm / k_norm over whole heads, the rotation, the cache write (nn/attention.h:170-177), the decode attention and Wo WITHOUT a
    residual (nn/attention.h:191-205; the block's post-norm adds it, nn/transformer.h:132-133) in one launch from the RAW wq|wk|wv rows,
    launched BY NAME on Gemma-7B's shapes (16 heads x 256, 16 kv heads; ranges of 64 tiles slots: 128 at S = 2048 so that the launch is one
    workgroup per CU) against the oracle's kernels composed as the reference composes them: rmsnorm per head (mu = 1), rope at the step's
    table row, the rows appended to the cache, the attention of test_one_launch_attention_matches_the_oracle, hadamard_broadcast + bmm for Wo."""
    import metalchat_amd as mc

    H, KV, hd, dim = 16, 16, 256, 3072
    max_seq, half = 1024 * tiles, hd // 2
    cfg = mg.tiny_cfg(BF16, dim=dim, n_heads=H, n_kv_heads=KV, head_dim=hd, ffn_dim=256, n_layers=1, vocab=64, max_seq_len=max_seq)
    w = mg.make_model(cfg, seed=91, quant="i4", group=128)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=128))
    dec.load_model(w)
    wo_p, wo_s, rows, inf, _ = dec.weight_ptrs(0, "wo")
    assert (rows, inf) == (dim, H * hd)
    rng = np.random.default_rng(4000 + n + tiles)
    n_rep, nsplit = H // KV, max_seq // (PB * tiles)
    slot, rrow, nrows, eps = n - 1, 3, 6, 1e-6
    q0 = mo.encode(BF16, rng.normal(0, 1.5, (H, hd)).astype(np.float32))
    k0 = mo.encode(BF16, rng.normal(0, 0.7, (KV, hd)).astype(np.float32))
    v0 = mo.encode(BF16, rng.normal(0, 0.5, (KV, hd)).astype(np.float32))
    qw = mo.encode(BF16, rng.uniform(-0.3, 0.4, hd).astype(np.float32))
    kw = mo.encode(BF16, rng.uniform(-0.3, 0.4, hd).astype(np.float32))
    k = mo.encode(BF16, rng.normal(0, 0.4, (n, KV, hd)).astype(np.float32))
    v = mo.encode(BF16, rng.normal(0, 0.5, (n, KV, hd)).astype(np.float32))
    fcos, fsin = np.zeros((nrows, half), np.float32), np.zeros((nrows, half), np.float32)
    L = mo.layout
    mo.rope_freqs(L(fcos.shape), fcos, L(fsin.shape), fsin, hd, 0, 10000.0)
    scale = float(mo.from_bf16(mo.to_bf16(np.array([hd ** -0.5], np.float32)))[0])
    # ---- the oracle's composition
    qn, kn = np.zeros_like(q0), np.zeros_like(k0)
    mo.rmsnorm(BF16, L(q0.shape), qn, L(q0.shape), q0, L((hd,)), qw, eps, 1.0)
    mo.rmsnorm(BF16, L(k0.shape), kn, L(k0.shape), k0, L((hd,)), kw, eps, 1.0)
    q1, k1 = np.zeros_like(qn), np.zeros_like(kn)
    mo.rope(BF16, L(qn.shape), q1, L(qn.shape), qn, L(fcos.shape), fcos, L(fsin.shape), fsin, 1, H, rrow)
    mo.rope(BF16, L(kn.shape), k1, L(kn.shape), kn, L(fcos.shape), fcos, L(fsin.shape), fsin, 1, KV, rrow)
    k[slot], v[slot] = k1, v0
    att = oracle_attention(q1, k, v, n_rep, scale)
    proj = oracle_linear(BF16, w["layers"][0]["wo"], att.reshape(1, 1, -1)).reshape(-1)
    # ---- the launch: raw rows as the wq|wk|wv GEMV stores them (q and k heads with the rotation's partners adjacent: [2 j] = natural [j],
    # [2 j + 1] = natural [j + hd / 2]; v natural), the cache with garbage in the step's slot
    def packed(a):
        o = np.zeros_like(a)
        o[:, 0::2], o[:, 1::2] = a[:, :half], a[:, half:]
        return o
    raw = acc.to_device(np.concatenate([packed(q0).reshape(-1), packed(k0).reshape(-1), v0.reshape(-1)]))
    kpast, vpast = k.copy(), v.copy()
    kpast[slot] = mo.encode(BF16, rng.normal(0, 30, (KV, hd)).astype(np.float32))
    vpast[slot] = mo.encode(BF16, rng.normal(0, 30, (KV, hd)).astype(np.float32))
    kc, vt = device_caches(acc, kpast, vpast, max_seq)
    psum = acc.to_device(np.zeros(2 * H * nsplit, np.uint64))
    slab = acc.to_device(np.zeros(2 * H * hd * nsplit, np.uint64))
    row_g = acc.to_device(np.zeros(H * hd // 2, np.uint64))
    attn_out = acc.alloc(H * hd * 2)
    qwb, kwb = acc.to_device(qw), acc.to_device(kw)
    cb, sb = acc.to_device(fcos.reshape(-1)), acc.to_device(fsin.reshape(-1))
    kern = acc.load(f"mc_attn_wo_qkn_i4_bfloat_hd256_k2_t{tiles}")
    W = lambda p_: acc.wrap(p_, 1 << 40)
    for epoch, layer_tag in ((1, 1), (1, 2), (9, 200)):
        yb = acc.to_device(np.full(dim, 0x7FC0, np.uint16))
        attn_out.upload(np.zeros(H * hd, np.uint16))
        st = np.zeros(12, np.int32)
        st[2], st[3], st[6], st[9] = n, slot, rrow, epoch
        state = acc.to_device(st)
        mc.KernelTask(kern, (nsplit * KV * 512, 1, 1), (512, 1, 1),
                      [raw, kc, vt, attn_out, psum, slab, row_g, state, np.uint32(n_rep), np.uint32(KV), np.uint32(max_seq), np.float32(scale),
                       np.uint32(nsplit), np.uint32(layer_tag), W(wo_p), W(wo_s), None, yb, np.uint32(dim), np.uint32(128), np.uint32(0), np.uint32(fast), None,
                       qwb, kwb, cb, sb, np.float32(eps), np.float32(1.0)])()
        acc.wait()
        assert int(state.download(np.int32, 12)[10]) == 0, "a hand-off of the launch gave up"
        kgot = kc.download(np.uint16, KV * max_seq * hd).reshape(KV, max_seq, hd)
        vgot = vt.download(np.uint16, KV * hd * max_seq).reshape(KV, hd, max_seq)
        parity.check(BF16, kgot[:, slot].reshape(-1), k1.reshape(-1), rel=3e-3, max_ulp=2, max_frac=0.2, what=f"K row the launch wrote (k_norm + rotation), n{n}")
        parity.exact(vgot[:, :, slot], v0, "V row the launch wrote")
        if n > 1:
            parity.exact(kgot[:, : slot], kpast[: slot].transpose(1, 0, 2), "the earlier K rows are untouched")
        parity.check(BF16, attn_out.download(np.uint16, H * hd), att.reshape(-1), rel=4e-3, max_ulp=2, max_frac=0.3, scale_aware=True,
                     what=f"attention row of the one launch, t{tiles} n{n} tag ({epoch}, {layer_tag})")
        parity.check(BF16, yb.download(np.uint16, dim), proj, rel=4e-3, max_ulp=2, max_frac=0.3, scale_aware=True,
                     what=f"q/k-norm + rope + attention + Wo in one launch, t{tiles} n{n} tag ({epoch}, {layer_tag})")
    dec.release()


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("n,tiles", [(2048, 2), (1000, 2), (1, 2), (129, 2), (40, 2), (200, 2), (1024, 1), (70, 1), (4096, 4), (3000, 4), (310, 4)])   # (tiles = 2: the step's slot in every 32-slot group of a 128-slot range)
@pytest.mark.parametrize("post", [0, 1])
def test_gemma_block_from_the_row_to_wo_in_one_launch_matches_the_oracle(acc, post, n, tiles, fast):
    """`mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p{1,2}_t{1,2}` (round 5; attn_block_kernels.hip qkv_qkn_in_launch): a gemma3 block from the row it
    is handed to Wo's output in ONE launch at Gemma-7B's widths (K = 3072: rows of 1.5 KiB; 16 heads x 256, 16 kv heads; S = 2048 as 16 ranges
    of 128 slots) -- `_p2_`: the previous linear's post-norm + the residual add, left in HBM (nn/transformer.h:138-139), then attention_norm,
    wq|wk|wv, q_norm / k_norm, rope, the cache write, the decode attention and Wo (nn/transformer.h:130-133, nn/attention.h:170-205);
    `_p1_`: from attention_norm on -- launched BY NAME against the oracle's kernels composed as the reference composes them.  The row is
    dyadic (its first normalisation is then the oracle's bit for bit, test_lin_kernels_gpu.py)."""
    import metalchat_amd as mc
    from test_lin_kernels_gpu import dyadic_row

    H, KV, hd, dim, max_seq = 16, 16, 256, 3072, 1024 * tiles   # (`_t1`: S = 1024 as 16 ranges of 64 slots)
    half = hd // 2
    cfg = mg.tiny_cfg(BF16, family=1, dim=dim, n_heads=H, n_kv_heads=KV, head_dim=hd, ffn_dim=256, n_layers=1, vocab=64, max_seq_len=max_seq,
                      rope_theta=10000.0, norm_eps=1e-6)
    w = mg.make_model(cfg, seed=515, quant="i4", group=128)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=128))
    dec.load_model(w)
    lw = w["layers"][0]
    wo_p, wo_s, rows, inf, _ = dec.weight_ptrs(0, "wo")
    qk_p, qk_s, qrows, qinf, _ = dec.weight_ptrs(0, "qkv")
    assert (rows, inf, qrows, qinf) == (dim, H * hd, (H + 2 * KV) * hd, dim)
    rng = np.random.default_rng(7000 + n + post)
    n_rep, nsplit = H // KV, max_seq // (PB * tiles)
    slot, rrow, nrows, eps, mu = n - 1, 4, 6, 1e-6, 1.0
    L = mo.layout
    x = dyadic_row(rng, dim)

    def norm(v, wgt):
        out = np.zeros((1, v.size), np.uint16)
        mo.rmsnorm(BF16, L((1, v.size)), out, L((1, v.size)), v.reshape(1, -1), L((v.size,)), wgt, eps, mu)
        return out.reshape(-1)

    if post:
        res = mo.encode(BF16, rng.normal(0, 1, dim).astype(np.float32))
        y = norm(x, lw["ffn_post_norm"])
        h = np.zeros((1, dim), np.uint16)
        mo.add(BF16, L((1, dim)), h, L((1, dim)), res.reshape(1, -1), L((1, dim)), y.reshape(1, -1))
        h = h.reshape(-1)
    else:
        res, h = None, x
    xn = norm(h, lw["attention_norm"])
    q0 = oracle_linear(BF16, lw["wq"], xn.reshape(1, 1, -1)).reshape(H, hd)
    k0 = oracle_linear(BF16, lw["wk"], xn.reshape(1, 1, -1)).reshape(KV, hd)
    v0 = oracle_linear(BF16, lw["wv"], xn.reshape(1, 1, -1)).reshape(KV, hd)
    qn, kn = np.zeros_like(q0), np.zeros_like(k0)
    mo.rmsnorm(BF16, L(q0.shape), qn, L(q0.shape), q0, L((hd,)), lw["q_norm"], eps, mu)
    mo.rmsnorm(BF16, L(k0.shape), kn, L(k0.shape), k0, L((hd,)), lw["k_norm"], eps, mu)
    fcos, fsin = np.zeros((nrows, half), np.float32), np.zeros((nrows, half), np.float32)
    mo.rope_freqs(L(fcos.shape), fcos, L(fsin.shape), fsin, hd, 0, 10000.0)
    q1, k1 = np.zeros_like(qn), np.zeros_like(kn)
    mo.rope(BF16, L(qn.shape), q1, L(qn.shape), qn, L(fcos.shape), fcos, L(fsin.shape), fsin, 1, H, rrow)
    mo.rope(BF16, L(kn.shape), k1, L(kn.shape), kn, L(fcos.shape), fcos, L(fsin.shape), fsin, 1, KV, rrow)
    k = mo.encode(BF16, rng.normal(0, 0.4, (n, KV, hd)).astype(np.float32))
    v = mo.encode(BF16, rng.normal(0, 0.5, (n, KV, hd)).astype(np.float32))
    k[slot], v[slot] = k1, v0
    scale = float(mo.from_bf16(mo.to_bf16(np.array([hd ** -0.5], np.float32)))[0])
    att = oracle_attention(q1, k, v, n_rep, scale)
    proj = oracle_linear(BF16, lw["wo"], att.reshape(1, 1, -1)).reshape(-1)
    # ---- the launch
    kpast, vpast = k.copy(), v.copy()
    kpast[slot] = mo.encode(BF16, rng.normal(0, 30, (KV, hd)).astype(np.float32))
    vpast[slot] = mo.encode(BF16, rng.normal(0, 30, (KV, hd)).astype(np.float32))
    kc, vt = device_caches(acc, kpast, vpast, max_seq)
    psum = acc.to_device(np.zeros(2 * H * nsplit, np.uint64))
    slab = acc.to_device(np.zeros(2 * H * hd * nsplit, np.uint64))
    row_g = acc.to_device(np.zeros(H * hd // 2, np.uint64))
    qkv_g = acc.to_device(np.zeros(2 * (H + 2 * KV) * hd // 2, np.uint64))
    attn_out = acc.alloc(H * hd * 2)
    nw, qnb, knb = acc.to_device(lw["attention_norm"]), acc.to_device(lw["q_norm"]), acc.to_device(lw["k_norm"])
    pwb = acc.to_device(lw["ffn_post_norm"]) if post else None
    resb = acc.to_device(res) if post else None
    cb, sb = acc.to_device(fcos.reshape(-1)), acc.to_device(fsin.reshape(-1))
    kern = acc.load(f"mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p{1 + post}_t{tiles}")
    W = lambda p_: acc.wrap(p_, 1 << 40)
    for epoch, layer_tag in ((1, 1), (1, 2), (9, 200)):
        xb = acc.to_device(x)
        yb = acc.to_device(np.full(dim, 0x7FC0, np.uint16))
        hb = acc.to_device(np.full(dim, 0x7FC0, np.uint16)) if post else None
        attn_out.upload(np.zeros(H * hd, np.uint16))
        st = np.zeros(12, np.int32)
        st[2], st[3], st[6], st[9] = n, slot, rrow, epoch
        state = acc.to_device(st)
        mc.KernelTask(kern, (nsplit * KV * 512, 1, 1), (512, 1, 1),
                      [kc, vt, attn_out, psum, slab, row_g, qkv_g, state, np.uint32(n_rep), np.uint32(KV), np.uint32(max_seq), np.float32(scale),
                       np.uint32(nsplit), np.uint32(layer_tag), W(wo_p), W(wo_s), xb, yb, np.uint32(dim), np.uint32(128), nw, W(qk_p), W(qk_s), cb, sb,
                       np.float32(eps), np.float32(mu), np.uint32(fast), None, qnb, knb, pwb, resb, hb])()
        acc.wait()
        assert int(state.download(np.int32, 12)[10]) == 0, "a hand-off of the launch gave up"
        if post:
            parity.check(BF16, hb.download(np.uint16, dim), h, rel=2e-3, max_ulp=1, max_frac=0.02, what=f"the hidden row workgroup 0 leaves (post-norm + residual), n{n}")
        kgot = kc.download(np.uint16, KV * max_seq * hd).reshape(KV, max_seq, hd)
        vgot = vt.download(np.uint16, KV * hd * max_seq).reshape(KV, hd, max_seq)
        parity.check(BF16, kgot[:, slot].reshape(-1), k1.reshape(-1), rel=4e-3, max_ulp=2, max_frac=0.3, scale_aware=True, what=f"K row the launch wrote, p{1 + post} n{n}")
        parity.check(BF16, vgot[:, :, slot].reshape(-1), v0.reshape(-1), rel=3e-3, max_ulp=2 if post else 1, max_frac=0.3 if post else 0.05, scale_aware=True,
                     what=f"V row the launch wrote, p{1 + post} n{n}")
        if n > 1:
            parity.exact(kgot[:, : slot], kpast[: slot].transpose(1, 0, 2), "the earlier K rows are untouched")
        # (`_p2_`: the row behind two norms is not the oracle's bit for bit, its q / k / v rows differ in the last place here and there, and
        #  q_norm / k_norm renormalise that: one element of the attention row measured 2.94 scaled steps at n = 300 -- the SAME element and
        #  distance (and 34 % of its elements one step off) with 64-, 128- and 256-slot ranges, i.e. in the form that is bit for bit the separate launches; three allowed, as in
        #  test_context_gpu.py::test_gemma_7b_widths_at_the_benchmark_context)
        parity.check(BF16, attn_out.download(np.uint16, H * hd), att.reshape(-1), rel=4e-3, max_ulp=3 if post else 2, max_frac=0.45 if post else 0.3, scale_aware=True,
                     what=f"attention row of the one launch, p{1 + post} n{n} tag ({epoch}, {layer_tag})")
        parity.check(BF16, yb.download(np.uint16, dim), proj, rel=4e-3, max_ulp=3 if post else 2, max_frac=0.45 if post else 0.3, scale_aware=True,
                     what=f"norms + wq|wk|wv + q/k-norm + rope + attention + Wo in one launch, p{1 + post} n{n} tag ({epoch}, {layer_tag})")
    dec.release()


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("n", [2048, 1000, 1, 65])
def test_llama3_70b_norm_qkv_rope_and_attention_in_one_launch_matches_the_oracle(acc, n, fast):
    """`mc_attn_qkv_i4_bfloat_hd128_q4` (round 5; attn_block_kernels.hip attn_qkv_body): attention_norm, wq|wk|wv on rows of 4 KiB (K = 8192), RoPE,
    the cache write and the decode attention of a Llama-3-70B block (64 heads x 128, 8 kv heads of 8 query heads: 640 rotation pairs per kv
    head, 20 per workgroup, gathered in two passes) in ONE launch, the attention row left in HBM for the Wo GEMV -- nn/transformer.h:130,
    nn/attention.h:170-203 -- launched BY NAME against the oracle's kernels composed as the reference composes them."""
    import metalchat_amd as mc
    from test_lin_kernels_gpu import dyadic_row, oracle_rmsnorm

    H, KV, hd, dim, max_seq = 64, 8, 128, 8192, 2048
    half = hd // 2
    cfg = mg.tiny_cfg(BF16, dim=dim, n_heads=H, n_kv_heads=KV, head_dim=hd, ffn_dim=256, n_layers=1, vocab=64, max_seq_len=max_seq)
    w = mg.make_model(cfg, seed=707, quant="i4", group=128)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=128))
    dec.load_model(w)
    lw = w["layers"][0]
    qk_p, qk_s, qrows, qinf, _ = dec.weight_ptrs(0, "qkv")
    assert (qrows, qinf) == ((H + 2 * KV) * hd, dim)
    rng = np.random.default_rng(2000 + n)
    n_rep, nsplit = H // KV, max_seq // PB
    slot, rrow, nrows = n - 1, 5, 8
    x = dyadic_row(rng, dim)
    k = mo.encode(BF16, rng.normal(0, 0.4, (n, KV, hd)).astype(np.float32))
    v = mo.encode(BF16, rng.normal(0, 0.5, (n, KV, hd)).astype(np.float32))
    fcos, fsin = np.zeros((nrows, half), np.float32), np.zeros((nrows, half), np.float32)
    L = mo.layout
    mo.rope_freqs(L(fcos.shape), fcos, L(fsin.shape), fsin, hd, 0, 500000.0)
    scale = float(mo.from_bf16(mo.to_bf16(np.array([hd ** -0.5], np.float32)))[0])
    xn = oracle_rmsnorm(x, lw["attention_norm"])
    q0 = oracle_linear(BF16, lw["wq"], xn.reshape(1, 1, -1)).reshape(H, hd)
    k0 = oracle_linear(BF16, lw["wk"], xn.reshape(1, 1, -1)).reshape(KV, hd)
    v0 = oracle_linear(BF16, lw["wv"], xn.reshape(1, 1, -1)).reshape(KV, hd)
    q1, k1 = np.zeros_like(q0), np.zeros_like(k0)
    mo.rope(BF16, L(q0.shape), q1, L(q0.shape), q0, L(fcos.shape), fcos, L(fsin.shape), fsin, 1, H, rrow)
    mo.rope(BF16, L(k0.shape), k1, L(k0.shape), k0, L(fcos.shape), fcos, L(fsin.shape), fsin, 1, KV, rrow)
    k[slot], v[slot] = k1, v0
    att = oracle_attention(q1, k, v, n_rep, scale)
    kpast, vpast = k.copy(), v.copy()
    kpast[slot] = mo.encode(BF16, rng.normal(0, 30, (KV, hd)).astype(np.float32))
    vpast[slot] = mo.encode(BF16, rng.normal(0, 30, (KV, hd)).astype(np.float32))
    kc, vt = device_caches(acc, kpast, vpast, max_seq)
    psum = acc.to_device(np.zeros(2 * H * nsplit, np.uint64))
    slab = acc.to_device(np.zeros(2 * H * hd * nsplit, np.uint64))
    qkv_g = acc.to_device(np.zeros(2 * (H + 2 * KV) * hd // 2, np.uint64))
    attn_out = acc.alloc(H * hd * 2)
    nw = acc.to_device(lw["attention_norm"])
    cb, sb = acc.to_device(fcos.reshape(-1)), acc.to_device(fsin.reshape(-1))
    kern = acc.load("mc_attn_qkv_i4_bfloat_hd128_q4")
    W = lambda p_: acc.wrap(p_, 1 << 40)
    xb = acc.to_device(x)
    for epoch, layer_tag in ((1, 1), (1, 2), (9, 200)):
        attn_out.upload(np.zeros(H * hd, np.uint16))
        st = np.zeros(12, np.int32)
        st[2], st[3], st[6], st[9] = n, slot, rrow, epoch
        state = acc.to_device(st)
        mc.KernelTask(kern, (nsplit * KV * 512, 1, 1), (512, 1, 1),
                      [kc, vt, attn_out, psum, slab, qkv_g, state, np.uint32(n_rep), np.uint32(KV), np.uint32(max_seq), np.float32(scale), np.uint32(nsplit),
                       np.uint32(layer_tag), xb, np.uint32(128), nw, W(qk_p), W(qk_s), cb, sb, np.float32(1e-5), np.float32(0.0), np.uint32(fast), None])()
        acc.wait()
        assert int(state.download(np.int32, 12)[10]) == 0, "a hand-off of the launch gave up"
        kgot = kc.download(np.uint16, KV * max_seq * hd).reshape(KV, max_seq, hd)
        vgot = vt.download(np.uint16, KV * hd * max_seq).reshape(KV, hd, max_seq)
        parity.check(BF16, kgot[:, slot].reshape(-1), k1.reshape(-1), rel=3e-3, max_ulp=2, max_frac=0.2, what=f"K row the launch wrote (GEMV + rotation), n{n}")
        parity.check(BF16, vgot[:, :, slot].reshape(-1), v0.reshape(-1), rel=2e-3, max_ulp=1, max_frac=0.05, what=f"V row the launch wrote, n{n}")
        if n > 1:
            parity.exact(kgot[:, : slot], kpast[: slot].transpose(1, 0, 2), "the earlier K rows are untouched")
        parity.check(BF16, attn_out.download(np.uint16, H * hd), att.reshape(-1), rel=4e-3, max_ulp=2, max_frac=0.3, scale_aware=True,
                     what=f"attention row of the one launch, n{n} tag ({epoch}, {layer_tag})")
    dec.release()
