"""The prompt pass: mc_decoder_prefill = nn::llama3 / nn::gemma3 operator() on
len > 1 tokens, against the oracle's mco_model_forward (oracle/mc_oracle.c): logits and greedy
token of the last row, the hidden row after every layer, the K / V cache rows it wrote (bit-exact
indexing, T tolerance on values), and the decode steps that continue from it.  Tolerances as in
test_decode_gpu.py.  PARITY UNPINNED by the reference (it asserts nothing about prompt logits)."""
import numpy as np
import pytest

import modelgen as mg
import parity
from oracle import mc_oracle as mo

pytestmark = pytest.mark.gpu
BF16, F32 = 0, 1


def tol(dt):
    # T = float shows the algorithm is the reference's (1e-4 everywhere).  In bf16 the rows of a
    # prompt feed each other through attention, so one-step differences (fp32 summation order of the
    # MFMA GEMM vs the oracle's sequential loop) compound across rows and layers: the vector-wise
    # bound is 2 bf16 steps (2^-7 = 7.8e-3) and every element stays within 2 scaled steps.
    return (1e-4, 1.0) if dt == F32 else (7.8e-3, 0.7)


def check_against_oracle(acc, cfg, weights, dec_over, tokens, start_pos=0, window=0, follow=3, warm=()):
    import metalchat_amd as mc

    dt = cfg["dtype"]
    rel, frac = tol(dt)
    om = mo.Model(cfg, weights)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, **dec_over))
    dec.load_model(weights)
    dec.set_taps(True)
    for p, t in enumerate(warm):          # an earlier context, fed token by token on both sides
        om.step(t, p)
        dec.step(t, p)
    otok, ologits = om.forward(tokens, start_pos, window)
    gtok = dec.prefill(tokens, start_pos, window)
    for layer in range(0, cfg["n_layers"]):
        parity.check(dt, dec.hidden(layer), om.hidden(layer), rel=rel, max_ulp=2, max_frac=frac,
                     what=f"prefill hidden[{layer}] (last row)")
    parity.check(dt, dec.logits(), ologits, rel=rel, max_ulp=2, max_frac=frac, what="prefill logits")
    for layer in (0, cfg["n_layers"] - 1):
        gk, gv = dec.export_kv(layer)
        ok, ov = om.kv(layer)
        assert gk.shape == ok.shape == (start_pos + len(tokens), cfg["n_kv_heads"], cfg["head_dim"])
        parity.check(dt, gk, ok, rel=rel, max_ulp=2, max_frac=frac, what=f"prefill K[{layer}]")
        parity.check(dt, gv, ov, rel=rel, max_ulp=2, max_frac=frac, what=f"prefill V[{layer}]")
    agree = int(gtok == otok)
    # decode continues from the prompt
    tok, pos = otok, start_pos + len(tokens)
    for _ in range(follow):
        if pos >= cfg["max_seq_len"]:
            break
        o2, ol2 = om.step(tok, pos)
        g2 = dec.step(tok, pos)
        parity.check(dt, dec.logits(), ol2, rel=rel, max_ulp=2, max_frac=frac, what=f"decode after prefill, pos {pos}")
        agree += int(g2 == o2)
        tok, pos = o2, pos + 1
    dec.release()
    om.close()
    return agree


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("quant,fmt,group", [(None, 0, 0), ("i8", 1, 32), ("i4", 2, 32), ("i4", 2, 128)])
def test_llama_prompt_matches_oracle(acc, dt, quant, fmt, group):
    cfg = mg.tiny_cfg(dt, max_seq_len=64)
    weights = mg.make_model(cfg, seed=71, quant=quant, group=group or 32)
    tokens = np.random.default_rng(fmt + 3).integers(0, cfg["vocab"], 21).tolist()   # 21: ragged vs the 16 / 64 tiles
    agree = check_against_oracle(acc, cfg, weights, dict(weight_format=fmt, group_size=group), tokens)
    assert agree >= (4 if dt == F32 else 3)


@pytest.mark.parametrize("n", [2, 16, 17, 64, 100, 256, 300])
def test_prompt_lengths_around_the_tile_edges(acc, n):
    cfg = mg.tiny_cfg(BF16, max_seq_len=320, n_layers=1)
    weights = mg.make_model(cfg, seed=72, quant="i4", group=32)
    tokens = np.random.default_rng(n).integers(0, cfg["vocab"], n).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, follow=1)


def test_head_dim_128_gqa4_prompt(acc):
    cfg = mg.tiny_cfg(BF16, dim=512, n_heads=4, n_kv_heads=1, head_dim=128, ffn_dim=1024, max_seq_len=64, n_layers=1)
    weights = mg.make_model(cfg, seed=73, quant="i4", group=128)
    tokens = np.random.default_rng(5).integers(0, cfg["vocab"], 40).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=128), tokens)


def test_wide_model_takes_the_split_k_gemm(acc):
    # K = 1024 / 2048 with few output tiles: the 128 x 128 GEMM splits K (2 and 4 ways here) and a
    # reduce kernel finishes the rows -- with adaptors, so the reduce epilogue carries them too
    cfg = mg.tiny_cfg(BF16, dim=1024, n_heads=8, n_kv_heads=2, head_dim=128, ffn_dim=2048, n_layers=1,
                      vocab=512, max_seq_len=96)
    weights = mg.make_model(cfg, seed=79, quant="i4", group=128, lora_rank=8)
    tokens = np.random.default_rng(9).integers(0, cfg["vocab"], 70).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=128), tokens, follow=1)


@pytest.mark.parametrize("dt", [F32, BF16])
def test_gemma3_prompt_with_sliding_window(acc, dt):
    cfg = mg.tiny_cfg(dt, family=1, n_layers=3, rope_sliding_theta=10000.0, sliding_stride=2,
                      attn_scale=float(1.0 / np.sqrt(48.0)), max_seq_len=64)
    weights = mg.make_model(cfg, seed=74, quant="i4", group=32)
    tokens = np.random.default_rng(6).integers(0, cfg["vocab"], 30).tolist()
    # window 5 < len: the second triangle of make_sliding_causal_mask (nn/attention.h:302-321) bites
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, window=5)


def test_gemma3_long_prompt_sliding_window(acc):
    cfg = mg.tiny_cfg(BF16, family=1, n_layers=2, rope_sliding_theta=10000.0, sliding_stride=2,
                      attn_scale=float(1.0 / np.sqrt(48.0)), max_seq_len=320)
    weights = mg.make_model(cfg, seed=80, quant="i4", group=32)
    tokens = np.random.default_rng(10).integers(0, cfg["vocab"], 270).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, window=37, follow=1)


def test_second_chunk_sees_only_itself_like_the_reference(acc):
    # make_causal_mask(len, end_pos) leaves the columns of the earlier context at -inf
    # (nn/attention.h:283-299): a prompt chunk at start_pos > 0 attends to its own rows only.
    cfg = mg.tiny_cfg(F32, max_seq_len=64)
    weights = mg.make_model(cfg, seed=75, quant="i4", group=32)
    rng = np.random.default_rng(7)
    warm = rng.integers(0, cfg["vocab"], 6).tolist()
    tokens = rng.integers(0, cfg["vocab"], 9).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, start_pos=6, warm=warm)


def test_prefill_argument_errors_and_len_one(acc):
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(F32, max_seq_len=16, n_layers=1)
    weights = mg.make_model(cfg, seed=76)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg))
    dec.load_model(weights)
    with pytest.raises(mc.McError, match=r"requested length \(17\) is larger than the cache size \(16\)"):
        dec.prefill(list(range(17)), 0)
    with pytest.raises(mc.McError, match="inside max_seq_len"):
        dec.prefill(list(range(8)), 10)
    with pytest.raises(mc.McError, match="outside the vocabulary"):
        dec.prefill([1, cfg["vocab"]], 0)
    # len == 1 is the decode step
    a = dec.prefill([5], 0)
    la = dec.logits()
    ref = mc.Decoder(acc, **mg.decoder_kwargs(cfg))
    ref.load_model(weights)
    assert ref.step(5, 0) == a and np.array_equal(ref.logits(), la)
    for d in (dec, ref):
        d.release()


@pytest.mark.parametrize("dt,n", [(F32, 21), (BF16, 21), (BF16, 90)])
def test_qlora_prompt(acc, dt, n):
    # lora_linear on every projection (quantization/lora.h:94-122) with M > 1 rows; 90 rows takes
    # the 128 x 128 tiling
    cfg = mg.tiny_cfg(dt, max_seq_len=96)
    weights = mg.make_model(cfg, seed=78, quant="i4", group=32, lora_rank=16)
    tokens = np.random.default_rng(8).integers(0, cfg["vocab"], n).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, follow=2)
