"""The oracle against an INDEPENDENT float64 formulation of the decode step (tests/golden/xcheck.py: PyTorch, HuggingFace
rotate_half RoPE, repeat_interleave GQA, exp / sum softmax, the sink cache as list slicing, dense QLoRA), on the parts the
reference's own tests leave unpinned: kernel/rope.metal:29-63, nn/cache.h:187-204, nn/attention.h:161-206,
quantization/lora.h:94-122, nn/gemma.h:42-146.  The float64 logits were computed in the build container and are stored
in tests/golden/xcheck_*.npz; the models are regenerated here from their seeds.  24 of the steps lie past max_seq_len."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import xcheck  # noqa: E402

import modelgen as mg  # noqa: E402


@pytest.mark.parametrize("name", sorted(xcheck.CASES))
def test_oracle_reproduces_the_independent_formulation(oracle, name):
    cfg, mk, steps = xcheck.case_cfg(name)
    ref = np.load(os.path.join(HERE, "golden", name + ".npz"))
    assert ref["logits"].shape == (steps, cfg["vocab"]) and steps - cfg["max_seq_len"] >= 24
    weights = mg.make_model(cfg, **mk)
    got, ids = xcheck.oracle_decode(cfg, weights, int(ref["first_token"]), steps, follow=ref["ids"])
    worst = xcheck.agreement(ref["logits"], got)
    assert worst <= 1e-5, f"{name}: oracle vs float64 formulation, max err / (|b| + rms) = {worst:.3g}"
    assert np.array_equal(ids, ref["ids"])  # (no near-tie among these logits: the greedy paths coincide)


def test_the_stored_vectors_are_what_the_script_computes():
    """the committed .npz are outputs of tests/golden/xcheck.py (needs torch: build container only)"""
    torch = pytest.importorskip("torch")
    assert torch is not None
    name = "xcheck_llama_gqa8"
    cfg, mk, steps = xcheck.case_cfg(name)
    ref = np.load(os.path.join(HERE, "golden", name + ".npz"))
    logits, ids = xcheck.torch_decode(cfg, mg.make_model(cfg, **mk), int(ref["first_token"]), steps)
    assert np.array_equal(ids, ref["ids"])
    assert np.allclose(logits, ref["logits"], rtol=1e-12, atol=1e-12)
