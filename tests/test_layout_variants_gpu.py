"""The row layouts added in round 2 against the kernels they replace, on the same weights and tokens:
  * rows of 1.5 KiB as super rows (gemv.h LSPLIT, K = 3072)      vs the classic kernels   (MC_LIN_SPLIT=0)
  * one row per wave for 2048-row matrices (LGEN `half`)         vs whole pairs per wave  (MC_LING_HALF=0): bit-identical
  * int8 rows of 14 KiB on the linear order (mac8b_n)            vs the classic kernel    (MC_I8_LING14=0)
Different groupings of the per-lane partial sums may move a logit by one bf16 step; the picks and the caches' new rows must
agree to that (the oracle comparisons at these shapes live in test_full_size_gpu.py and test_context_gpu.py)."""
import os

import numpy as np
import pytest

import modelgen as mg

import metalchat_amd as mc


def _decoder(acc, cfg, weights, wfmt, group, env):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        d = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=wfmt, group_size=group))
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    d.load_model(weights)
    return d


def _bf16_steps(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    scale = np.maximum(np.abs(a), np.abs(b)).max() * 2.0 ** -8
    return np.abs(a - b).max() / scale


@pytest.mark.gpu
@pytest.mark.parametrize("name,knob,quant,wfmt,dims,exact", [
    ("split rows", "MC_LIN_SPLIT", "i4", mc.WFMT_I4, dict(dim=3072, n_heads=24, n_kv_heads=8, head_dim=128, ffn_dim=4096), False),
    ("row per wave", "MC_LING_HALF", None, mc.WFMT_T, dict(dim=2048, n_heads=16, n_kv_heads=4, head_dim=128, ffn_dim=5632), True),
    ("int8 14 KiB", "MC_I8_LING14", "i8", mc.WFMT_I8, dict(dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336), False),
])
def test_new_row_layouts_agree_with_the_kernels_they_replace(name, knob, quant, wfmt, dims, exact):
    acc = mc.HardwareAccelerator(ordinal=0)
    cfg = mg.tiny_cfg(0, n_layers=1, vocab=4096, max_seq_len=32, **dims)
    weights = mg.make_model(cfg, seed=21, quant=quant, group=128)
    group = 128 if quant else 0
    new = _decoder(acc, cfg, weights, wfmt, group, {knob: "1"})
    old = _decoder(acc, cfg, weights, wfmt, group, {knob: "0"})
    tok = 3
    for pos in range(4):
        a, b = new.step(tok, pos), old.step(tok, pos)
        la, lb = np.asarray(new.logits()), np.asarray(old.logits())
        if exact:
            assert np.array_equal(la, lb), f"{name}: pos {pos}"
            assert a == b
        else:
            assert _bf16_steps(la, lb) <= 3.0, f"{name}: pos {pos}: {_bf16_steps(la, lb):.2f} bf16 steps"
        tok = b
    ka, va = new.export_kv(0)
    kb, vb = old.export_kv(0)
    if exact:
        assert np.array_equal(ka, kb) and np.array_equal(va, vb)
    else:
        assert _bf16_steps(ka, kb) <= 2.0 and _bf16_steps(va, vb) <= 2.0
    new.release()
    old.release()
