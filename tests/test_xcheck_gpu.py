"""The HIP decode path (T = float) against the INDEPENDENT float64 formulation of tests/golden/xcheck.py -- stored logits,
models regenerated from their seeds: RoPE in the QKV epilogue, the zero-copy sink ring, GQA inside the MFMA tiles, the
QLoRA adaptor launches and the gemma3 block, 24 steps past max_seq_len, with no code of oracle/ in between."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import xcheck  # noqa: E402

import modelgen as mg  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(xcheck.CASES))
def test_hip_decoder_reproduces_the_independent_formulation(acc, name):
    import metalchat_amd as mc

    cfg, mk, steps = xcheck.case_cfg(name)
    ref = np.load(os.path.join(HERE, "golden", name + ".npz"))
    weights = mg.make_model(cfg, **mk)
    fmt = {"i4": mc.WFMT_I4, "i8": mc.WFMT_I8}[mk["quant"]]
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=fmt, group_size=mk["group"]))
    dec.load_model(weights)
    tok, got = int(ref["first_token"]), []
    for pos in range(steps):
        t = dec.step(tok, pos)
        got.append(dec.logits().astype(np.float32))
        assert t == int(ref["ids"][pos]), (name, pos)
        tok = int(ref["ids"][pos])
    worst = xcheck.agreement(ref["logits"], np.stack(got))
    assert worst <= 1e-4, f"{name}: HIP f32 path vs float64 formulation, max err / (|b| + rms) = {worst:.3g}"
    dec.release()
