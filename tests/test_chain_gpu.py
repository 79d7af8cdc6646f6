"""Round 6: the attention block AND ffn_norm + w1|w3 + SiLU * mul in one launch behind a run-ahead loader wave
(mc_attn_qkv_wo_w13_i4_bfloat_hd128_k2_q2_f2, attn_block_kernels.hip; MC_CHAIN_W13=1) at BASELINE configs[1]'s own widths.

  * against the oracle: one full-width Llama-3-8B block, S = 2048, kv_len 2041 .. 2048 and eight rolls past the end -- the bounds of the
    three-launch layer's test (test_context_gpu.py::test_llama3_8b_int4_at_the_benchmark_context);
  * BIT FOR BIT the two launches it replaces (mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2 + mc_gemv_i4_bfloat_lin2_p1_e2): hidden rows of every
    block, logits, tokens, both blocks' caches, near an empty cache and across the end of a full one;
  * graph replay == eager launches over a chain of tokens that crosses the end of the cache;
  * a range of slots past kv_len (inactive workgroups: the loader joins two barriers fewer) -- the same identity at position 70.
"""
import numpy as np
import pytest

import modelgen as mg
import parity
from test_context_gpu import random_cache, run_injected
from test_full_size_gpu import FULL_WIDTH, SEED, synth_model

pytestmark = pytest.mark.gpu
BF16 = 0
CHAIN = "mc_attn_qkv_wo_w13_i4_bfloat_hd128_k2_q2_f2"
BLOCK = "mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2"
W13 = "mc_gemv_i4_bfloat_lin2_p1_e2"


def test_chained_block_against_the_oracle_at_the_benchmark_context(acc, monkeypatch):
    import metalchat_amd as mc

    monkeypatch.setenv("MC_CHAIN_W13", "1")
    cfg = dict(dtype=BF16, n_layers=1, vocab=2048, max_seq_len=2048, norm_eps=1e-5, **FULL_WIDTH["llama3-8b"])
    weights = synth_model(cfg, SEED)
    names = set()
    agree = run_injected(acc, cfg, weights, 2040, 16, dict(weight_format=mc.WFMT_I4, group_size=128),
                         rel_logits=5e-3, max_ulp=2, max_frac=0.7, what="8B int4 S=2048 chained", launched=names)
    assert agree >= 14
    assert CHAIN in names and BLOCK not in names and W13 not in names, sorted(names)
    assert "mc_gemv_i4_bfloat_lin7_p0_e1" in names, sorted(names)


def test_chained_block_equals_the_two_launches_bit_for_bit(acc, monkeypatch):
    import metalchat_amd as mc

    cfg = dict(dtype=BF16, n_layers=2, vocab=2048, norm_eps=1e-5, max_seq_len=2048, **FULL_WIDTH["llama3-8b"])
    S = cfg["max_seq_len"]
    out = {}
    for form in ("chain", "three"):
        monkeypatch.setenv("MC_CHAIN_W13", "1" if form == "chain" else "0")
        dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=128))
        dec.init_synthetic(SEED)
        dec.set_taps(True)
        dec.launch_log(True)
        rows = []
        for n_inject in (2, 70, S - 4):  # (70: the ranges past slot 127 are inactive -- their loaders join two barriers fewer)
            for layer in range(cfg["n_layers"]):
                k, v = random_cache(cfg, n_inject, 700 + layer)
                dec.import_kv(layer, k, v)
            tok = 5
            for i in range(8):
                tok = dec.step(tok, n_inject + i)
                rows.append((tok, dec.logits().copy(), np.stack([dec.hidden(l) for l in range(-1, cfg["n_layers"])])))
        caches = [dec.export_kv(l) for l in range(cfg["n_layers"])]
        names = set(dec.launched())
        if form == "chain":
            assert CHAIN in names and BLOCK not in names and W13 not in names, sorted(names)
        else:
            assert CHAIN not in names and {BLOCK, W13} <= names, sorted(names)
        assert dec.handoff_fallbacks() == 0
        out[form] = (rows, caches)
        dec.release()
    for i, ((ta, la, ha), (tb_, lb, hb)) in enumerate(zip(out["chain"][0], out["three"][0])):
        assert ta == tb_, i
        parity.exact(ha, hb, f"step {i}: hidden rows, chained launch vs the two launches")
        parity.exact(la, lb, f"step {i}: logits")
    for l, ((ka, va), (kb, vb)) in enumerate(zip(out["chain"][1], out["three"][1])):
        parity.exact(ka, kb, f"block {l}: K cache")
        parity.exact(va, vb, f"block {l}: V cache")


def test_chained_block_graph_replay_equals_eager_across_the_end_of_the_cache(acc, monkeypatch):
    import metalchat_amd as mc

    cfg = dict(dtype=BF16, n_layers=3, vocab=4096, norm_eps=1e-5, max_seq_len=2048, **FULL_WIDTH["llama3-8b"])
    S = cfg["max_seq_len"]
    toks = {}
    for form in ("chain-graph", "chain-eager", "three-graph"):
        monkeypatch.setenv("MC_CHAIN_W13", "0" if form.startswith("three") else "1")
        dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=128))
        dec.init_synthetic(SEED)
        for layer in range(cfg["n_layers"]):
            k, v = random_cache(cfg, S - 12, 900 + layer)
            dec.import_kv(layer, k, v)
        if form.endswith("graph"):
            toks[form] = list(dec.generate(9, S - 12, 40))
        else:
            t, seq = 9, []
            for i in range(40):
                t = dec.step(t, S - 12 + i)
                seq.append(t)
            toks[form] = seq
        assert dec.handoff_fallbacks() == 0
        dec.release()
    assert toks["chain-graph"] == toks["chain-eager"] == toks["three-graph"]
