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


def in_torch_free_child(request):
    """The opt-in library GEMM needs a process whose only ROCm stack is the system's: once `import torch` has happened (tests/ckptgen.py
    writes its checkpoints with it, earlier in a full-suite session) the wheel's private HIP runtime is in the global scope and
    hipblasLtCreate dies inside it ("no ROCm-capable device is detected", exit 1) -- the reason tests/test_pipeline_gpu.py starts its RCCL
    rank in a child without torch.  So: in such a session the test re-runs itself, alone, in a fresh interpreter and reports that
    result.  True: the child ran (and passed)."""
    import os
    import subprocess
    import sys

    if "torch" not in sys.modules or os.environ.get("MC_TEST_CHILD"):
        return False
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", request.node.nodeid, "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=root,
                       env=dict(os.environ, MC_TEST_CHILD="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    return True


def tol(dt):
    # T = float shows the algorithm is the reference's (1e-4 everywhere).  In bf16 the rows of a
    # prompt feed each other through attention, so one-step differences (fp32 summation order of the
    # MFMA GEMM vs the oracle's sequential loop) compound across rows and layers: the vector-wise
    # bound is 2 bf16 steps (2^-7 = 7.8e-3) and every element stays within 2 scaled steps.
    return (1e-4, 1.0) if dt == F32 else (7.8e-3, 0.7)


def check_against_oracle(acc, cfg, weights, dec_over, tokens, start_pos=0, window=0, follow=3, warm=(), expect_kernel=None):
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
    if expect_kernel:
        dec.launch_log(True)
    gtok = dec.prefill(tokens, start_pos, window)
    if expect_kernel:
        assert expect_kernel in set(dec.launched()), sorted(set(dec.launched()))
        dec.launch_log(False)
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
    with pytest.raises(mc.McError, match="straddle the end of the cache"):
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


# ------------------------------------------------------------------------------------------------
# nn::sink_cache::copy with len > 1 behind a full cache (nn/cache.h:187-204): post region rotated left by
# len, the chunk takes the last len rows -- against the oracle's literal restatement, with a turned ring
# on the HIP side (decode steps past max_seq_len before the chunk) and decode steps behind it.
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,family", [(F32, 0), (BF16, 0), (F32, 1)])
def test_prompt_chunk_behind_a_full_cache(acc, dtype, family):
    import metalchat_amd as mc

    over = dict(max_seq_len=32, n_layers=2, family=family)
    if family == 1:
        over.update(rope_sliding_theta=10000.0, sliding_stride=2)
    cfg = mg.tiny_cfg(dtype, **over)
    weights = mg.make_model(cfg, seed=21, quant="i4", group=32)
    rel, frac = tol(dtype)
    om = mo.Model(cfg, weights)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=32))
    dec.load_model(weights)
    rng = np.random.default_rng(5)
    win = 8 if family == 1 else 0
    # a first prompt, then decode PAST the end of the cache: the HIP ring turns (ring_base = 9)
    p0 = rng.integers(0, cfg["vocab"], 20)
    otok, _ = om.forward(p0, 0, win)
    dec.prefill(p0, 0, win)
    tok, pos = otok, 20
    for _ in range(21):                     # positions 20 .. 40
        o2, ol = om.step(tok, pos)
        dec.step(tok, pos)
        tok, pos = o2, pos + 1
    parity.check(dtype, dec.logits(), ol, rel=rel, max_ulp=2, max_frac=frac, what="decode before the chunk")
    # the chunk: 10 rows at start_pos 41 >= max_seq_len
    chunk = rng.integers(0, cfg["vocab"], 10)
    otok, ologits = om.forward(chunk, pos, win)
    gtok = dec.prefill(chunk, pos, win)
    parity.check(dtype, dec.logits(), ologits, rel=rel, max_ulp=2, max_frac=frac, what="chunk behind a full cache: logits")
    for layer in range(cfg["n_layers"]):
        gk, gv = dec.export_kv(layer)
        ok, ov = om.kv(layer)
        assert gk.shape == ok.shape == (32, cfg["n_kv_heads"], cfg["head_dim"])
        parity.check(dtype, gk, ok, rel=rel, max_ulp=2, max_frac=frac, what=f"chunk behind a full cache: K[{layer}]")
        parity.check(dtype, gv, ov, rel=rel, max_ulp=2, max_frac=frac, what=f"chunk behind a full cache: V[{layer}]")
    pos += 10
    tok = otok
    agree = int(gtok == otok)
    for _ in range(6):                      # and decode on: the ring turns again from its linear state
        o2, ol = om.step(tok, pos)
        g2 = dec.step(tok, pos)
        parity.check(dtype, dec.logits(), ol, rel=rel, max_ulp=2, max_frac=frac, what=f"decode behind the chunk, pos {pos}")
        agree += int(g2 == o2)
        tok, pos = o2, pos + 1
    assert agree >= (7 if dtype == F32 else 5)
    # a second chunk right away (ring turned by the six steps), as long as the whole cache
    chunk2 = rng.integers(0, cfg["vocab"], 32)
    otok, ologits = om.forward(chunk2, pos, win)
    dec.prefill(chunk2, pos, win)
    parity.check(dtype, dec.logits(), ologits, rel=rel, max_ulp=2, max_frac=frac, what="cache-sized chunk: logits")
    gk, _ = dec.export_kv(0)
    ok, _ = om.kv(0)
    parity.check(dtype, gk, ok, rel=rel, max_ulp=2, max_frac=frac, what="cache-sized chunk: K[0]")
    # a new conversation on the same decoder
    otok, ologits = om.forward(p0, 0, win)
    dec.prefill(p0, 0, win)
    parity.check(dtype, dec.logits(), ologits, rel=rel, max_ulp=2, max_frac=frac, what="new conversation: logits")
    dec.release()
    om.close()


def test_prompt_chunk_errors_are_the_reference_errors(acc):
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(BF16, max_seq_len=32, n_layers=1)
    weights = mg.make_model(cfg, seed=2, quant="i4", group=32)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=32))
    dec.load_model(weights)
    om = mo.Model(cfg, weights)
    toks = np.arange(10) % cfg["vocab"]
    dec.prefill(toks, 0)
    om.forward(toks, 0)
    before = dec.export_kv(0)[0].copy()
    # rows [28, 38) straddle the end of a 32-row cache: the reference's clamped slice fails clone's same_numel check
    with pytest.raises(mc.McError, match="straddle the end of the cache"):
        dec.prefill(toks, 28)
    with pytest.raises(ValueError):
        om.forward(toks, 28)
    # nn/cache.h:178-183
    with pytest.raises(mc.McError, match=r"sink_cache: requested length \(33\) is larger than the cache size \(32\)"):
        dec.prefill(np.zeros(33, np.int64), 40)
    parity.exact(dec.export_kv(0)[0], before, "a rejected chunk leaves the cache alone")
    dec.release()
    om.close()


def test_prompt_pass_through_a_local_pipeline_equals_the_single_stage(acc):
    # mc_pipeline_prefill: the [len][dim] hidden rows hop stage to stage; same kernels per layer -> same bits
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(BF16, max_seq_len=32, n_layers=4)
    weights = mg.make_model(cfg, seed=8, quant="i4", group=32)
    single = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=32))
    single.load_model(weights)
    toks = np.random.default_rng(3).integers(0, cfg["vocab"], 18)
    want_tok = single.prefill(toks, 0)
    want_logits = single.logits().copy()
    want_more = list(single.generate(want_tok, 18, 20))     # runs past the end of the cache
    want_chunk = single.prefill(toks[:9], 38)                # a chunk behind the full cache
    want_logits2 = single.logits().copy()
    for world in (2, 4):
        stages = []
        for r in range(world):
            lb, le = mc.pipeline_layer_range(r, world, cfg["n_layers"])
            d = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=32, layer_begin=lb, layer_end=le))
            d.load_model(weights)
            stages.append(d)
        pipe = mc.Pipeline.local(stages)
        assert pipe.prefill(toks, 0) == want_tok
        parity.exact(stages[-1].logits(), want_logits, f"world {world}: prompt logits")
        assert list(pipe.generate(want_tok, 18, 20)) == want_more
        assert pipe.prefill(toks[:9], 38) == want_chunk
        parity.exact(stages[-1].logits(), want_logits2, f"world {world}: logits of the chunk behind the full cache")
        pipe.release()
        for d in stages:
            d.release()
    single.release()


@pytest.mark.parametrize("n", [2, 8, 33, 64])
def test_short_prompt_takes_the_weight_streaming_gemm(acc, n, monkeypatch):
    """Prompts of up to 64 rows on int4 weights with scale groups of 128 multiply from the quad-interleaved copy of the weights
    (prefill_kernels.hip mc_pf2_gemm_i4_bfloat: matrix-pipe dequantisation straight into MFMA operands; the copy is built by
    mc_pf2_repack_i4 when the first such prompt arrives): against the oracle like every prompt pass, with adaptors (the reduce
    epilogue carries them), K = 1024 / 2048 (8 and 16 steps of 128, split over the grid) -- and the launch log must name the
    kernels.  A 65-row prompt takes the tiled GEMM again; weights uploaded afterwards rebuild the copy."""
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(BF16, dim=1024, n_heads=8, n_kv_heads=2, head_dim=128, ffn_dim=2048, n_layers=2, vocab=512, max_seq_len=96)
    weights = mg.make_model(cfg, seed=81, quant="i4", group=128, lora_rank=8)
    tokens = np.random.default_rng(n).integers(0, cfg["vocab"], n).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=128), tokens, follow=1)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=2, group_size=128))
    dec.load_model(weights)
    dec.launch_log(True)
    t_new = dec.prefill(tokens, 0)
    names = set(dec.launched())
    assert "mc_pf2_gemm_i4_bfloat" in names and "mc_pf2_repack_i4" in names, sorted(names)
    logits_new = dec.logits().copy()
    dec.launch_log(True)
    dec.prefill(tokens, 0)
    assert "mc_pf2_repack_i4" not in set(dec.launched()), "the copy is built once"
    dec.launch_log(True)
    dec.prefill(np.random.default_rng(1).integers(0, cfg["vocab"], 65).tolist(), 0)
    assert "mc_pf2_gemm_i4_bfloat" not in set(dec.launched())
    # new weights: the copy follows
    w2 = mg.make_model(cfg, seed=82, quant="i4", group=128, lora_rank=8)
    dec.load_model(w2)
    dec.launch_log(True)
    dec.prefill(tokens, 0)
    assert "mc_pf2_repack_i4" in set(dec.launched())
    got2 = dec.logits().copy()
    dec.release()
    # ... and equals a decoder that never saw the first weights, bit for bit; the tiled GEMM (MC_PF2=0) within the suite's bound
    fresh = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=2, group_size=128))
    fresh.load_model(w2)
    fresh.prefill(tokens, 0)
    parity.exact(got2, fresh.logits(), "logits after a weight reload")
    fresh.release()
    monkeypatch.setenv("MC_PF2", "0")
    old = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=2, group_size=128))
    old.load_model(weights)
    old.launch_log(True)
    t_old = old.prefill(tokens, 0)
    assert "mc_pf2_gemm_i4_bfloat" not in set(old.launched())
    rel, frac = tol(BF16)
    parity.check(BF16, logits_new, old.logits(), rel=rel, max_ulp=2, max_frac=frac, what="weight-streaming GEMM vs tiled GEMM")
    old.release()
    assert t_new is not None and t_old is not None


@pytest.mark.parametrize("n", [33, 70, 300])
def test_split_k_reduce_folded_into_the_consumers_is_bit_identical(acc, n, monkeypatch):
    """A prompt GEMM that splits K hands its fp32 partial sums to the kernel that consumes its rows (rope + cache write, the next
    rmsnorm with the residual, act * mul: prefill_kernels.hip mc_pf_*_parts_bfloat) instead of to a reduce launch.  The consumers
    round T(sum in z order) exactly as the reduce did: logits, hidden rows and caches are the same bits with MC_PF_FOLD=0."""
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(BF16, dim=1024, n_heads=8, n_kv_heads=2, head_dim=128, ffn_dim=2048, n_layers=3, vocab=512, max_seq_len=320)
    weights = mg.make_model(cfg, seed=83, quant="i4", group=128)
    tokens = np.random.default_rng(n).integers(0, cfg["vocab"], n).tolist()
    out = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("MC_PF_FOLD", fold)
        dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=2, group_size=128))
        dec.load_model(weights)
        dec.set_taps(True)
        dec.launch_log(True)
        tok = dec.prefill(tokens, 0)
        names = set(dec.launched())
        if fold == "1":
            assert {"mc_pf_rope_cache_parts_bfloat", "mc_pf_act_mul_parts_bfloat", "mc_pf_rmsnorm_parts_bfloat"} & names, sorted(names)
        else:
            assert not [x for x in names if "_parts_" in x], sorted(names)
        out[fold] = (tok, dec.logits().copy(), [dec.hidden(layer).copy() for layer in range(cfg["n_layers"])],
                     [tuple(a.copy() for a in dec.export_kv(layer)) for layer in range(cfg["n_layers"])])
        dec.release()
    assert out["1"][0] == out["0"][0]
    parity.exact(out["1"][1], out["0"][1], "logits, reduce folded vs reduce launches")
    for layer in range(cfg["n_layers"]):
        parity.exact(out["1"][2][layer], out["0"][2][layer], f"hidden[{layer}]")
        parity.exact(out["1"][3][layer][0], out["0"][3][layer][0], f"K[{layer}]")
        parity.exact(out["1"][3][layer][1], out["0"][3][layer][1], f"V[{layer}]")


def test_exp_table_holds_the_rounded_double_exp_of_every_bfloat16_and_the_lds_window_reads_it_back(acc):
    """prefill_kernels.hip: the prompt pass takes exp of bfloat16 values (scores, silu arguments) from a 65536-entry table
    built with the same exp_precise the kernels used inline.  (a) every entry is the correctly rounded exp the oracle
    computes ((float)exp((double)x), oracle/mc_oracle.py softmax / silu); (b) the attention's 8192-entry LDS window
    with clamped magnitude bits returns the table's value for EVERY bfloat16 bit pattern (NaNs pass through)."""
    import metalchat_amd as mc

    tab = acc.alloc(65536 * 4)
    out = acc.alloc(65536 * 4)
    mc.KernelTask(acc.load("mc_exp_table_bfloat"), (65536, 1, 1), (256, 1, 1), [tab])()
    mc.KernelTask(acc.load("mc_pf_exp_window_bfloat"), (8 * 256, 1, 1), (256, 1, 1), [tab, out])()
    acc.wait()
    t = tab.download(np.float32, 65536)
    w = out.download(np.float32, 65536)
    x = (np.arange(65536, dtype=np.uint32) << 16).view(np.float32)
    with np.errstate(over="ignore", under="ignore", invalid="ignore"):
        want = np.exp(x.astype(np.float64)).astype(np.float32)
    nan = np.isnan(x)
    assert np.array_equal(t[~nan].view(np.uint32), want[~nan].view(np.uint32))
    assert np.isnan(t[nan]).all()
    assert np.array_equal(w[~nan].view(np.uint32), t[~nan].view(np.uint32))
    assert np.array_equal(w[nan].view(np.uint32), x[nan].view(np.uint32))
    assert t[0xFF80] == 0.0 and t[0x7F80] == np.inf and t[0] == 1.0 and t[0x8000] == 1.0


@pytest.mark.parametrize("fmt,quant,group", [(2, "i4", 128), (1, "i8", 32), (0, None, 0)])
@pytest.mark.parametrize("n", [256, 300])
def test_activation_in_the_gemm_epilogue_is_bit_identical_to_the_separate_launch(acc, n, fmt, quant, group, monkeypatch):
    """From 256 rows on an unsplit w1|w3 GEMM finishes silu(a) * b in its epilogue (prefill_kernels.hip pf_gemm_big_body, EPI 3:
    neighbouring lanes hold the pair) and mc_pf_act_mul never launches.  Same roundings in the same order: every tap is the
    same bits with MC_PF_ACT_EPI=0, and both agree with the oracle like any prompt (test_prompt_lengths_around_the_tile_edges
    runs through this epilogue)."""
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(BF16, dim=256, n_heads=4, n_kv_heads=2, head_dim=64, ffn_dim=768, n_layers=2, vocab=384, max_seq_len=320)
    weights = mg.make_model(cfg, seed=89, quant=quant, group=group or 32)
    tokens = np.random.default_rng(n).integers(0, cfg["vocab"], n).tolist()
    out = {}
    for epi in ("1", "0"):
        monkeypatch.setenv("MC_PF_ACT_EPI", epi)
        dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=fmt, group_size=group))
        dec.load_model(weights)
        dec.set_taps(True)
        dec.launch_log(True)
        tok = dec.prefill(tokens, 0)
        names = set(dec.launched())
        fused = [x for x in names if x.endswith("_e3")]
        assert (len(fused) == 1 and not [x for x in names if "act_mul" in x]) if epi == "1" else (not fused and [x for x in names if "act_mul" in x]), sorted(names)
        out[epi] = (tok, dec.logits().copy(), [dec.hidden(layer).copy() for layer in range(cfg["n_layers"])])
        dec.release()
    assert out["1"][0] == out["0"][0]
    parity.exact(out["1"][1], out["0"][1], "logits, activation in the epilogue vs its own launch")
    for layer in range(cfg["n_layers"]):
        parity.exact(out["1"][2][layer], out["0"][2][layer], f"hidden[{layer}]")


# ---- the two-head prompt attention (mc_pf_attn2_bfloat_hd*: transposed scores, P in registers, exp from the LDS window).  The
# decoder takes it when the grid fills the chip (>= 2 workgroups per CU: 512 rows of a 32-head model); MC_PF_ATTN_HEADS=2 takes it
# on the small models the oracle can follow.
@pytest.mark.parametrize("n", [2, 17, 100, 300])
def test_two_head_prompt_attention_matches_oracle(acc, n, monkeypatch):
    monkeypatch.setenv("MC_PF_ATTN_HEADS", "2")
    cfg = mg.tiny_cfg(BF16, max_seq_len=320)  # head_dim 32, 8 heads on 2 kv heads
    weights = mg.make_model(cfg, seed=91, quant="i4", group=32)
    tokens = np.random.default_rng(n).integers(0, cfg["vocab"], n).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, follow=1, expect_kernel="mc_pf_attn2_bfloat_hd32")


@pytest.mark.parametrize("hd,heads,kv,nh", [(64, 4, 2, 2), (128, 8, 2, 2), (128, 8, 2, 4), (128, 4, 1, 4)])
def test_two_head_prompt_attention_head_dims(acc, hd, heads, kv, nh, monkeypatch):
    """... and the four-head form of head_dim 128 (mc_pf_attn4_bfloat_hd128: the decoder's choice from 512 rows of Llama-3-8B on)"""
    monkeypatch.setenv("MC_PF_ATTN_HEADS", str(nh))
    cfg = mg.tiny_cfg(BF16, dim=256, n_heads=heads, n_kv_heads=kv, head_dim=hd, max_seq_len=192)
    weights = mg.make_model(cfg, seed=92, quant="i4", group=32)
    tokens = np.random.default_rng(hd).integers(0, cfg["vocab"], 150).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, follow=1, expect_kernel=f"mc_pf_attn{nh}_bfloat_hd{hd}")


@pytest.mark.parametrize("n", [2, 31, 33, 64, 65, 130, 300, 448])
def test_prompt_attention_through_lds_tiles_matches_oracle(acc, n, monkeypatch):
    """mc_pf_attn8_bfloat_hd128 (round 5: K / V tiles of 64 keys through LDS by LDS-DMA, 32 rows x 4 heads per workgroup, row tiles in pairs -- the prompt
    attention of 1024 rows and more; MC_PF_ATTN_HEADS=8 takes it on the models the oracle can follow): rows below, at and above one
    row tile and one key tile, a ragged last tile, GQA 4 and 8; row tiles one per workgroup and in pairs (tile x with tile last - x: an
    odd count leaves the middle one alone)."""
    monkeypatch.setenv("MC_PF_ATTN_HEADS", "8")
    for heads, kvh in ((8, 2), (8, 1)):
        monkeypatch.setenv("MC_PF_ATTN8_PAIR", "1" if kvh == 2 else "0")
        cfg = mg.tiny_cfg(BF16, dim=256, n_heads=heads, n_kv_heads=kvh, head_dim=128, ffn_dim=512, n_layers=2, vocab=384, max_seq_len=448)
        weights = mg.make_model(cfg, seed=131 + kvh, quant="i4", group=128)
        tokens = np.random.default_rng(n + kvh).integers(0, cfg["vocab"], n).tolist()
        check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=128), tokens, follow=1, expect_kernel="mc_pf_attn8_bfloat_hd128")


def test_prompt_attention_through_lds_tiles_sliding_window_and_second_chunk(acc, monkeypatch):
    """... under gemma3's sliding window (whole key tiles below the window are skipped) and for a chunk behind an earlier context (the
    reference's mask leaves the earlier columns at -inf: nn/attention.h:283-321)"""
    monkeypatch.setenv("MC_PF_ATTN_HEADS", "8")
    monkeypatch.setenv("MC_PF_ATTN8_PAIR", "1")
    cfg = mg.tiny_cfg(BF16, family=1, dim=256, n_heads=4, n_kv_heads=1, head_dim=128, ffn_dim=512, n_layers=2, rope_sliding_theta=10000.0,
                      sliding_stride=2, attn_scale=float(1.0 / np.sqrt(48.0)), max_seq_len=320)
    weights = mg.make_model(cfg, seed=133, quant="i4", group=32)
    tokens = np.random.default_rng(14).integers(0, cfg["vocab"], 270).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, window=37, follow=1, expect_kernel="mc_pf_attn8_bfloat_hd128")
    cfg = mg.tiny_cfg(BF16, dim=256, n_heads=4, n_kv_heads=1, head_dim=128, ffn_dim=512, n_layers=1, vocab=384, max_seq_len=320)
    weights = mg.make_model(cfg, seed=134, quant="i4", group=32)
    warm = np.random.default_rng(15).integers(0, cfg["vocab"], 21).tolist()
    tokens = np.random.default_rng(16).integers(0, cfg["vocab"], 150).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, start_pos=21, warm=warm, expect_kernel="mc_pf_attn8_bfloat_hd128")


def test_four_head_prompt_attention_sliding_window_and_second_chunk(acc, monkeypatch):
    monkeypatch.setenv("MC_PF_ATTN_HEADS", "4")
    cfg = mg.tiny_cfg(BF16, family=1, dim=256, n_heads=4, n_kv_heads=1, head_dim=128, n_layers=2, rope_sliding_theta=10000.0,
                      sliding_stride=2, max_seq_len=320)
    weights = mg.make_model(cfg, seed=93, quant="i4", group=32)
    tokens = np.random.default_rng(12).integers(0, cfg["vocab"], 270).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, window=37, follow=1, expect_kernel="mc_pf_attn4_bfloat_hd128")
    cfg = mg.tiny_cfg(BF16, dim=256, n_heads=8, n_kv_heads=2, head_dim=128, max_seq_len=96)
    weights = mg.make_model(cfg, seed=94, quant="i4", group=32)
    rng = np.random.default_rng(13)
    warm = rng.integers(0, cfg["vocab"], 21).tolist()
    tokens = rng.integers(0, cfg["vocab"], 40).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, start_pos=21, warm=warm, expect_kernel="mc_pf_attn4_bfloat_hd128")


def test_two_head_prompt_attention_sliding_window_and_second_chunk(acc, monkeypatch):
    monkeypatch.setenv("MC_PF_ATTN_HEADS", "2")
    # gemma3 block, window 37 over 270 rows: tiles left of the window, whole tiles, the two diagonals
    cfg = mg.tiny_cfg(BF16, family=1, n_layers=2, rope_sliding_theta=10000.0, sliding_stride=2,
                      attn_scale=float(1.0 / np.sqrt(48.0)), max_seq_len=320)
    weights = mg.make_model(cfg, seed=80, quant="i4", group=32)
    tokens = np.random.default_rng(10).integers(0, cfg["vocab"], 270).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, window=37, follow=1, expect_kernel="mc_pf_attn2_bfloat_hd32")
    # a chunk at start_pos > 0 attends to its own rows only (nn/attention.h:283-299): the columns of the earlier context are masked
    cfg = mg.tiny_cfg(BF16, max_seq_len=96)
    weights = mg.make_model(cfg, seed=75, quant="i4", group=32)
    rng = np.random.default_rng(7)
    warm = rng.integers(0, cfg["vocab"], 21).tolist()
    tokens = rng.integers(0, cfg["vocab"], 40).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, start_pos=21, warm=warm, expect_kernel="mc_pf_attn2_bfloat_hd32")


# ---- the 256 x 256 ping-pong GEMM of long prompts (kernels/pf_gemm8.h, decoder.cc g8_ok: from 257 rows on, two row tiles; MC_PF_GEMM8_ROWS lowers the
# gate so that shorter prompts of the models the oracle can follow reach it too).  Kernel-level parity: tests/test_gemm8_gpu.py.
@pytest.mark.parametrize("quant,fmt,group,copy", [("i4", 2, 32, "0"), ("i4", 2, 128, "0"), ("i8", 1, 32, "0"), (None, 0, 0, None),
                                                  ("i4", 2, 128, None), ("i8", 1, 32, None)])
@pytest.mark.parametrize("n,gate", [(400, None), (300, None), (257, None), (200, "192")])
def test_long_prompts_take_the_ping_pong_gemm_and_match_the_oracle(acc, n, gate, quant, fmt, group, copy, monkeypatch):
    """Every linear of the block through mc_pf_gemm8_*: wq|wk|wv / wo / w2 with a plain store or residual (these models are too
    narrow to split K), w1|w3 with silu * mul in its epilogue; the library is never called.  Quantised matrices multiply from their
    dequantised bfloat16 copies (round 6, decoder.cc plain_copy_ok: the default where the copies fit an eighth of the device's memory --
    mc_pf_gemm8_w_* on linear_w::wd) or, with MC_PF_PLAIN_COPY=0, from the quantised rows (no copy: mc_decoder_derived_weight_bytes stays 0)."""
    import metalchat_amd as mc

    if gate:
        monkeypatch.setenv("MC_PF_GEMM8_ROWS", gate)
    if copy is not None:
        monkeypatch.setenv("MC_PF_PLAIN_COPY", copy)
    else:
        monkeypatch.delenv("MC_PF_PLAIN_COPY", raising=False)
    cfg = mg.tiny_cfg(BF16, dim=256, n_heads=4, n_kv_heads=2, head_dim=64, ffn_dim=768, n_layers=2, vocab=384, max_seq_len=448)
    weights = mg.make_model(cfg, seed=101, quant=quant, group=group or 32)
    tokens = np.random.default_rng(n + fmt).integers(0, cfg["vocab"], n).tolist()
    copied = fmt != 0 and copy is None
    f = "w" if copied else {2: "i4", 1: "i8", 0: "w"}[fmt]
    check_against_oracle(acc, cfg, weights, dict(weight_format=fmt, group_size=group), tokens, follow=2, expect_kernel=f"mc_pf_gemm8_{f}_bfloat_e3")
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=fmt, group_size=group))
    dec.load_model(weights)
    dec.launch_log(True)
    dec.prefill(tokens, 0)
    names = dec.launched()
    assert not [x for x in names if x.startswith("mc_pf_gemm") and not x.startswith("mc_pf_gemm8_")], sorted(set(names))
    assert "hipblasLtMatmul" not in names
    lin_weights = cfg["n_layers"] * (cfg["dim"] * (cfg["n_heads"] + 2 * cfg["n_kv_heads"]) * cfg["head_dim"] + cfg["n_heads"] * cfg["head_dim"] * cfg["dim"]
                                     + 3 * cfg["dim"] * cfg["ffn_dim"])
    if copied:
        assert not [x for x in names if x.startswith("mc_pf_gemm8_i")], sorted(set(names))
        assert len([x for x in names if x.startswith("mc_pf_dequant_rows")]) == 4 * cfg["n_layers"]
        assert dec.derived_weight_bytes() == 2 * lin_weights
        dec.launch_log(True)
        dec.prefill(tokens, 0)   # the copies are built once
        assert not [x for x in dec.launched() if x.startswith("mc_pf_dequant_rows")]
    else:
        assert not [x for x in names if x.startswith("mc_pf_dequant_rows")]
        assert dec.derived_weight_bytes() == 0
    dec.release()


@pytest.mark.parametrize("family,kvh,n,window,pair", [(1, 4, 270, 37, "1"), (1, 4, 40, 0, None), (1, 2, 129, 0, "0"), (1, 4, 300, 0, "1"), (1, 4, 300, 0, "0"),
                                                       (1, 4, 65, 0, "1"), (1, 4, 193, 5, "1")])
def test_head_dim_256_prompt_attention(acc, family, kvh, n, window, pair, monkeypatch):
    """head_dim 256 (Gemma-7B: as many kv heads as query heads) against the oracle -- gemma3's sliding window, a short prompt, long ones, GQA 2.
    Round 6: mc_pf_attn8_bfloat_hd256 (pf_attn_lds_body<256, 1, 4>: 64 rows of one head share K / V tiles of 64 keys through LDS, row tiles single
    and in pairs) from 513 rows on (MC_PF_ATTN8_ROWS256 lowers the gate for these small models); below that, and with MC_PF_ATTN_HEADS=1, the round-1 kernel -- same token, same bounds."""
    over = dict(family=family, dim=256, n_heads=4, n_kv_heads=kvh, head_dim=256, ffn_dim=512, n_layers=2, vocab=384, max_seq_len=320)
    if family == 1:
        over.update(rope_sliding_theta=10000.0, sliding_stride=2)
    cfg = mg.tiny_cfg(BF16, **over)
    weights = mg.make_model(cfg, seed=151, quant="i4", group=32)
    tokens = np.random.default_rng(n).integers(0, cfg["vocab"], n).tolist()
    if pair is not None:
        monkeypatch.setenv("MC_PF_ATTN8_PAIR", pair)
        monkeypatch.setenv("MC_PF_ATTN8_ROWS256", "64")
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, window=window, follow=2,
                         expect_kernel="mc_pf_attn8_bfloat_hd256" if pair is not None else "mc_pf_attn_bfloat_hd256")
    monkeypatch.setenv("MC_PF_ATTN_HEADS", "1")
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, window=window, follow=1, expect_kernel="mc_pf_attn_bfloat_hd256")


@pytest.mark.parametrize("family,heads,kvh,n,window,pair", [(0, 8, 1, 300, 0, "1"), (0, 8, 1, 300, 0, "0"), (0, 8, 2, 300, 0, "1"), (0, 8, 2, 270, 0, "0"),
                                                             (1, 8, 1, 270, 37, "1"), (1, 8, 2, 193, 5, "1"), (0, 16, 2, 65, 0, "1"), (0, 8, 2, 129, 0, None)])
def test_head_dim_64_prompt_attention_through_lds_tiles(acc, family, heads, kvh, n, window, pair, monkeypatch):
    """Round 6: head_dim 64 with eight or four query heads per kv head (TinyLlama-1.1B, Llama-3.2-1B) -- mc_pf_attn8_bfloat_hd64_h8 / _h4
    (pf_attn_lds_body<64, 8, 8> / <64, 4, 8>: K rows of 128 bytes, eight waves = 8 heads x 16 rows or 4 heads x 32 rows on one K / V tile) against the
    oracle: single row tiles and pairs, gemma3's sliding window, ragged last tiles; MC_PF_ATTN_HEADS=2 (the two-head kernel of round 4) within the same bounds."""
    over = dict(family=family, dim=256, n_heads=heads, n_kv_heads=kvh, head_dim=64, ffn_dim=512, n_layers=2, vocab=384, max_seq_len=320)
    if family == 1:
        over.update(rope_sliding_theta=10000.0, sliding_stride=2)
    cfg = mg.tiny_cfg(BF16, **over)
    weights = mg.make_model(cfg, seed=161, quant="i4", group=32)
    tokens = np.random.default_rng(n + heads).integers(0, cfg["vocab"], n).tolist()
    monkeypatch.setenv("MC_PF_ATTN8_ROWS64", "64")   # (the product's gate sits where the kernel starts to pay on TinyLlama's widths)
    if pair is not None:
        monkeypatch.setenv("MC_PF_ATTN8_PAIR", pair)
    kern = "mc_pf_attn8_bfloat_hd64_h8" if heads // kvh % 8 == 0 else "mc_pf_attn8_bfloat_hd64_h4"
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, window=window, follow=2, expect_kernel=kern)
    monkeypatch.setenv("MC_PF_ATTN_HEADS", "2")
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, window=window, follow=1, expect_kernel="mc_pf_attn2_bfloat_hd64")


def test_gemma_post_norms_fold_into_the_consumer_of_a_split_gemm(acc, monkeypatch):
    """gemma3 blocks wide enough for Wo and w2 to split K: mc_pf_rmsnorm2_parts_bfloat sums the fp32 partials, applies the post norm with the
    residual and the next norm in one launch -- against the oracle, and bit for bit MC_PF_NORM2=0 (mc_pf_splitk_reduce_bfloat + two
    mc_pf_rmsnorm_bfloat): logits, tokens, caches."""
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(BF16, family=1, dim=2048, n_heads=8, n_kv_heads=2, head_dim=128, ffn_dim=1024, n_layers=2, vocab=512, max_seq_len=448,
                      rope_sliding_theta=10000.0, sliding_stride=2)
    weights = mg.make_model(cfg, seed=171, quant="i4", group=128)
    tokens = np.random.default_rng(71).integers(0, cfg["vocab"], 400).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=128), tokens, follow=2, expect_kernel="mc_pf_rmsnorm2_parts_bfloat")
    out = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("MC_PF_NORM2", fold)
        dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=2, group_size=128))
        dec.load_model(weights)
        dec.launch_log(True)
        toks = [dec.prefill(tokens, 0)]
        names = dec.launched()
        assert names.count("mc_pf_rmsnorm2_parts_bfloat") == (2 * cfg["n_layers"] if fold == "1" else 0), sorted(set(names))
        lg = dec.logits().copy()
        for i in range(3):
            toks.append(dec.step(toks[-1], len(tokens) + i))
        out[fold] = (toks, lg, dec.logits().copy(), [dec.export_kv(l) for l in range(cfg["n_layers"])])
        dec.release()
    assert out["1"][0] == out["0"][0]
    parity.exact(out["1"][1], out["0"][1], "logits of the prompt, folded post norms vs three launches")
    parity.exact(out["1"][2], out["0"][2], "logits three tokens later")
    for l, ((ka, va), (kb, vb)) in enumerate(zip(out["1"][3], out["0"][3])):
        parity.exact(ka, kb, f"block {l}: K cache")
        parity.exact(va, vb, f"block {l}: V cache")


def test_gelu_table_is_the_function_for_every_bfloat16(acc):
    """Round 6: mc_gelu_table_bfloat's table IS T(gelu) of every bfloat16 value -- mc_pf_act_mul_bfloat over all 65536 values as `a` (b = 1) with the
    table and with the fp64 tanh per element give the same rows (NaN inputs aside: their payloads are not compared)."""
    import metalchat_amd as mc

    a = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    pairs = np.empty(2 * 65536, np.uint16)
    pairs[0::2] = a
    pairs[1::2] = 0x3F80   # b = 1.0
    inp = acc.to_device(pairs)
    tab = acc.alloc(65536 * 4)
    mc.KernelTask(acc.load("mc_gelu_table_bfloat"), (256 * 256, 1, 1), (256, 1, 1), [tab])()
    outs = []
    for t in (tab, None):
        out = acc.to_device(np.zeros(65536, np.uint16))
        mc.KernelTask(acc.load("mc_pf_act_mul_bfloat"), ((65536 // 4 // 256 + 1) * 256, 1, 1), (256, 1, 1), [inp, out, np.uint32(65536), np.int32(1), t])()
        acc.wait()
        outs.append(out.download(np.uint16, 65536))
    finite = (a & 0x7F80) != 0x7F80
    nan_in = ((a & 0x7F80) == 0x7F80) & ((a & 0x007F) != 0)
    assert np.array_equal(outs[0][~nan_in], outs[1][~nan_in])
    assert finite.sum() == 65536 - 256 and nan_in.sum() == 254
    t32 = tab.download(np.float32, 65536)
    assert t32[0x3F80] == 0.83984375 and t32[0] == 0.0   # gelu(1) = 0.8412 -> 0.83984375 as a bfloat16; gelu(0) = 0


@pytest.mark.parametrize("quant,fmt,group,copy", [("i4", 2, 32, None), ("i4", 2, 32, "0"), (None, 0, 0, None)])
def test_gemma_long_prompt_takes_the_gelu_epilogue(acc, quant, fmt, group, copy, monkeypatch):
    """gemma3 blocks, 300 rows: w1|w3 with gelu(w1 x) * (w3 x) in the epilogue of the 256 x 256 GEMM (mc_pf_gemm8_*_e4, T(gelu) from the table) --
    against the oracle, and bit for bit MC_PF_GELU_TABLE=0 (no table: the GEMM stores both halves and mc_pf_act_mul_bfloat evaluates the fp64 tanh)."""
    import metalchat_amd as mc

    if copy is not None:
        monkeypatch.setenv("MC_PF_PLAIN_COPY", copy)
    cfg = mg.tiny_cfg(BF16, family=1, dim=256, n_heads=4, n_kv_heads=2, head_dim=64, ffn_dim=768, n_layers=2, vocab=384, max_seq_len=320,
                      rope_sliding_theta=10000.0, sliding_stride=2)
    weights = mg.make_model(cfg, seed=141, quant=quant, group=group or 32)
    tokens = np.random.default_rng(41).integers(0, cfg["vocab"], 300).tolist()
    f = "w" if (fmt == 0 or copy is None) else "i4"
    check_against_oracle(acc, cfg, weights, dict(weight_format=fmt, group_size=group), tokens, follow=2, expect_kernel=f"mc_pf_gemm8_{f}_bfloat_e4")
    out = {}
    for table in ("1", "0"):
        monkeypatch.setenv("MC_PF_GELU_TABLE", table)
        dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=fmt, group_size=group))
        dec.load_model(weights)
        dec.launch_log(True)
        tok = dec.prefill(tokens, 0)
        names = set(dec.launched())
        assert bool([x for x in names if x.endswith("_e4")]) == (table == "1"), sorted(names)
        assert ("mc_pf_act_mul_bfloat" in names) == (table == "0"), sorted(names)
        out[table] = (tok, dec.logits().copy(), [dec.export_kv(l) for l in range(cfg["n_layers"])])
        dec.release()
    assert out["1"][0] == out["0"][0]
    parity.exact(out["1"][1], out["0"][1], "logits of the prompt, gelu from the table vs the fp64 tanh")
    for l, ((ka, va), (kb, vb)) in enumerate(zip(out["1"][2], out["0"][2])):
        parity.exact(ka, kb, f"block {l}: K cache")
        parity.exact(va, vb, f"block {l}: V cache")


@pytest.mark.parametrize("family,hd,n", [(0, 128, 300), (0, 128, 37), (1, 128, 300), (1, 256, 300), (1, 256, 37), (0, 64, 300), (0, 256, 300), (0, 256, 21), (0, 32, 70)])
def test_packed_rope_launch_changes_no_bit(acc, family, hd, n, monkeypatch):
    """Round 6: mc_pf_rope_cache{,_parts}_v4_bfloat give a thread four rotation pairs (a quarter of the waves: the one-pair launch was bound by the
    rate waves start at) and write the transposed V cache 16 slots at a time.  MC_PF_ROPE_PACK=0 is the launch of rounds 1-5: logits, tokens and
    both caches are equal bit for bit -- with gemma3's q / k norms too (family 1; head_dim 128 and 256, where the sum over a head is the old
    launch's butterfly addition for addition)."""
    import metalchat_amd as mc

    over = dict(family=family, dim=256, n_heads=4, n_kv_heads=2, head_dim=hd, ffn_dim=512, n_layers=2, vocab=384, max_seq_len=320)
    if family == 1:
        over.update(rope_sliding_theta=10000.0, sliding_stride=2)
    cfg = mg.tiny_cfg(BF16, **over)
    weights = mg.make_model(cfg, seed=107, quant="i4", group=32)
    tokens = np.random.default_rng(n + hd).integers(0, cfg["vocab"], n).tolist()
    out = {}
    for pack in ("1", "0"):
        monkeypatch.setenv("MC_PF_ROPE_PACK", pack)
        dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=2, group_size=32))
        dec.load_model(weights)
        dec.launch_log(True)
        toks = [dec.prefill(tokens, 0)]
        names = set(dec.launched())
        dec.launch_log(False)
        assert bool([x for x in names if "rope_cache" in x and "_v4_" in x]) == (pack == "1"), sorted(names)
        lg = dec.logits().copy()
        for i in range(3):
            toks.append(dec.step(toks[-1], n + i))
        out[pack] = (toks, lg, dec.logits().copy(), [dec.export_kv(l) for l in range(cfg["n_layers"])])
        dec.release()
    assert out["1"][0] == out["0"][0]
    parity.exact(out["1"][1], out["0"][1], "logits of the prompt, packed vs one unit per workgroup")
    parity.exact(out["1"][2], out["0"][2], "logits three tokens later")
    for l, ((ka, va), (kb, vb)) in enumerate(zip(out["1"][3], out["0"][3])):
        parity.exact(ka, kb, f"block {l}: K cache")
        parity.exact(va, vb, f"block {l}: V cache")


@pytest.mark.parametrize("quant,fmt,group", [("i4", 2, 128), ("i4", 2, 32), ("i8", 1, 32)])
def test_the_dequantised_copy_changes_no_bit_of_a_long_prompt(acc, quant, fmt, group, monkeypatch):
    """MC_PF_PLAIN_COPY=1 against =0: the copy holds Wd = T(T(q) T(s)), the values the quantised loop stages in LDS, and mc_pf_gemm8_w_* is the
    same loop -- hidden rows of every block, logits, the next tokens and both caches are equal bit for bit."""
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(BF16, dim=2048, n_heads=4, n_kv_heads=1, head_dim=128, ffn_dim=1024, n_layers=2, vocab=512, max_seq_len=448)
    weights = mg.make_model(cfg, seed=105, quant=quant, group=group)
    tokens = np.random.default_rng(9).integers(0, cfg["vocab"], 400).tolist()
    out = {}
    for copy in ("1", "0"):
        monkeypatch.setenv("MC_PF_PLAIN_COPY", copy)
        dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=fmt, group_size=group))
        dec.load_model(weights)
        dec.set_taps(True)
        dec.launch_log(True)
        tok = dec.prefill(tokens, 0)
        names = set(dec.launched())
        assert bool([x for x in names if x.startswith("mc_pf_gemm8_w_")]) == (copy == "1"), sorted(names)
        assert bool([x for x in names if x.startswith("mc_pf_gemm8_i")]) == (copy == "0"), sorted(names)
        rows = [np.stack([dec.hidden(l) for l in range(cfg["n_layers"])]), dec.logits().copy()]
        toks = [tok]
        for i in range(4):
            toks.append(dec.step(toks[-1], len(tokens) + i))
        out[copy] = (toks, rows, dec.logits().copy(), [dec.export_kv(l) for l in range(cfg["n_layers"])])
        dec.release()
    assert out["1"][0] == out["0"][0]
    parity.exact(out["1"][1][0], out["0"][1][0], "hidden rows of the prompt, copy vs quantised rows")
    parity.exact(out["1"][1][1], out["0"][1][1], "logits of the prompt")
    parity.exact(out["1"][2], out["0"][2], "logits after four more tokens")
    for l, ((ka, va), (kb, vb)) in enumerate(zip(out["1"][3], out["0"][3])):
        parity.exact(ka, kb, f"block {l}: K cache")
        parity.exact(va, vb, f"block {l}: V cache")


@pytest.mark.parametrize("copy", ["0", None])
def test_wide_long_prompt_splits_k_in_the_ping_pong_gemm(acc, copy, monkeypatch):
    """K = 2048 with two column tiles: the 256 x 256 GEMM splits K (mc_pf_gemm8_*_e2) and the consumers add the fp32 partial sums"""
    if copy is not None:
        monkeypatch.setenv("MC_PF_PLAIN_COPY", copy)
    else:
        monkeypatch.delenv("MC_PF_PLAIN_COPY", raising=False)
    cfg = mg.tiny_cfg(BF16, dim=2048, n_heads=4, n_kv_heads=1, head_dim=128, ffn_dim=512, n_layers=1, vocab=256, max_seq_len=400)
    weights = mg.make_model(cfg, seed=102, quant="i4", group=128)
    tokens = np.random.default_rng(3).integers(0, cfg["vocab"], 390).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=128), tokens, follow=1,
                         expect_kernel="mc_pf_gemm8_i4_bfloat_e2" if copy == "0" else "mc_pf_gemm8_w_bfloat_e2")


@pytest.mark.slow   # (hipBLASLt is opt-in since round 5, MC_PF_BLASLT=1: a comparison aid, not the product path)
# ---- the library GEMM of long prompts (decoder.cc gemm_lib: hipBLASLt on the dequantised bfloat16 copy of a matrix).  The decoder
# takes it -- ONLY when asked to, MC_PF_BLASLT=1: an opt-in comparison since round 5, the default path builds no dequantised copy --
# where a launch has >= 48 tiles of 256 x 256 (w1|w3 of Llama-3-8B from 256 rows on, every matrix from 768); MC_PF_BLASLT=2 takes it for every
# prompt GEMM that can, which is how the models the oracle can follow reach it.
@pytest.mark.parametrize("quant,fmt,group", [("i4", 2, 32), ("i4", 2, 128), ("i8", 1, 32), (None, 0, 0)])
@pytest.mark.parametrize("n", [21, 300])
def test_library_gemm_of_long_prompts_matches_oracle(acc, n, quant, fmt, group, monkeypatch, request):
    """The operand is Wd = T(T(q) T(s)) (kernel/mul.metal:78-82) as a bfloat16 copy, the sums are fp32, rounded to T once
    (nn/linear.h:70-81): the same bounds as every other prompt path, through bfloat16 rows out (w1|w3) and through the fp32
    rows the consumers of a split GEMM add up (wq|wk|wv, wo, w2)."""
    if in_torch_free_child(request):
        return
    monkeypatch.setenv("MC_PF_BLASLT", "2")
    cfg = mg.tiny_cfg(BF16, dim=256, n_heads=4, n_kv_heads=2, head_dim=64, ffn_dim=768, n_layers=2, vocab=384, max_seq_len=320)
    weights = mg.make_model(cfg, seed=97, quant=quant, group=group or 32)
    tokens = np.random.default_rng(n + fmt).integers(0, cfg["vocab"], n).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=fmt, group_size=group), tokens, follow=2, expect_kernel="hipblasLtMatmul")


@pytest.mark.slow   # (hipBLASLt is opt-in since round 5, MC_PF_BLASLT=1: a comparison aid, not the product path)
def test_library_gemm_is_the_decoders_choice_only_where_a_launch_has_enough_tiles(acc, monkeypatch, request):
    """512 rows on a model with a 16384-row w1|w3: that GEMM alone goes to the library (128 tiles of 256 x 256 >= 48; the others have 2-8), the others keep
    the prompt kernels; the token and the rows agree with the kernels-only prompt (MC_PF_BLASLT=0) like two orders of the same
    fp32 sums; and new weights rebuild the dequantised copy."""
    import metalchat_amd as mc

    if in_torch_free_child(request):
        return

    cfg = mg.tiny_cfg(BF16, dim=512, n_heads=4, n_kv_heads=2, head_dim=128, ffn_dim=8192, n_layers=2, vocab=512, max_seq_len=512)
    weights = mg.make_model(cfg, seed=98, quant="i4", group=128)
    tokens = np.random.default_rng(5).integers(0, cfg["vocab"], 512).tolist()
    out = {}
    monkeypatch.setenv("MC_PF_PLAIN_COPY", "0")   # (round 6's default would build the copy of EVERY matrix: this test counts the library's own)
    for lib in ("1", "0"):
        monkeypatch.setenv("MC_PF_BLASLT", lib)
        dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=2, group_size=128))
        dec.load_model(weights)
        dec.set_taps(True)
        dec.launch_log(True)
        tok = dec.prefill(tokens, 0)
        names = dec.launched()
        n_lib = sum(1 for x in names if x == "hipblasLtMatmul")
        assert n_lib == (cfg["n_layers"] if lib == "1" else 0), sorted(set(names))
        if lib == "1":
            assert sum(1 for x in names if x.startswith("mc_pf_dequant_rows_i4")) == cfg["n_layers"]
            assert not [x for x in names if x.endswith("_e3")] and [x for x in names if "act_mul" in x]
            dec.launch_log(True)
            dec.prefill(tokens, 0)      # the copy exists now
            assert not [x for x in dec.launched() if x.startswith("mc_pf_dequant_rows")]
        out[lib] = (tok, dec.logits().copy(), [dec.hidden(layer).copy() for layer in range(cfg["n_layers"])])
        dec.release()
    rel, frac = tol(BF16)
    for layer in range(cfg["n_layers"]):
        parity.check(BF16, out["1"][2][layer], out["0"][2][layer], rel=rel, max_ulp=2, max_frac=frac, what=f"hidden[{layer}], library vs kernels")
    parity.check(BF16, out["1"][1], out["0"][1], rel=rel, max_ulp=2, max_frac=frac, what="logits, library vs kernels")


@pytest.mark.slow   # (hipBLASLt is opt-in since round 5, MC_PF_BLASLT=1: a comparison aid, not the product path)
def test_library_gemm_under_a_gemma3_block(acc, monkeypatch, request):
    """gemma3 (gelu, post-norms, sliding window: nn/gemma.h:110-137): the plain-store GEMMs of its block through the library"""
    if in_torch_free_child(request):
        return
    monkeypatch.setenv("MC_PF_BLASLT", "2")
    cfg = mg.tiny_cfg(BF16, family=1, n_layers=2, rope_sliding_theta=10000.0, sliding_stride=2,
                      attn_scale=float(1.0 / np.sqrt(48.0)), max_seq_len=320)
    weights = mg.make_model(cfg, seed=99, quant="i4", group=32)
    tokens = np.random.default_rng(12).integers(0, cfg["vocab"], 270).tolist()
    check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=32), tokens, window=37, follow=1, expect_kernel="hipblasLtMatmul")
