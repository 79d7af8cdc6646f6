"""The committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py) must be
reproduced by the oracle bit for bit: the oracle is the checker of every GPU parity test, so it may
not drift.  PARITY UNPINNED BY THE REFERENCE for these paths (no reference test of the rope
rotation, the sink_cache roll branch or a whole decode step -- SURVEY.md s.8c); the kernels the
reference does test are pinned in test_oracle_reference_pins.py."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as G  # noqa: E402
from oracle import mc_oracle as mo  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", sorted(G.DECODE_CASES))
def test_oracle_reproduces_decode_golden(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg, weights, _, steps = G.decode_case(name)
    assert json.loads(str(z["cfg"])) == json.loads(json.dumps(cfg))
    assert str(z["digest"]) == G.weights_digest(weights), "weight generator changed: regenerate the golden files"
    om = mo.Model(cfg, weights)
    tok = int(z["first_token"])
    for pos in range(steps):
        nt, lg = om.step(tok, pos)
        assert nt == int(z["tokens"][pos])
        assert np.array_equal(np.asarray(lg), z["logits"][pos])
        for i, layer in enumerate(range(-1, cfg["n_layers"])):
            assert np.array_equal(np.asarray(om.hidden(layer)), z["hidden"][pos][i])
        tok = nt
    k, v = om.kv(cfg["n_layers"] - 1)
    assert np.array_equal(k, z["k_last"]) and np.array_equal(v, z["v_last"])
    # runs that went past max_seq_len: the logical view is exactly max_seq_len rows long
    assert k.shape[0] == min(steps, cfg["max_seq_len"])
    om.close()


def test_oracle_reproduces_rope_golden():
    z = np.load(os.path.join(GOLD, "rope.npz"))
    L = mo.layout
    fcos, fsin = np.zeros_like(z["fcos"]), np.zeros_like(z["fsin"])
    mo.rope_freqs(L(fcos.shape), fcos, L(fsin.shape), fsin, int(z["hd"]), int(z["table_start"]), float(z["theta"]))
    assert np.array_equal(fcos, z["fcos"]) and np.array_equal(fsin, z["fsin"])
    for dt, tag in ((1, "f32"), (0, "bf16")):
        x = z["x_" + tag]
        y = np.zeros_like(x)
        mo.rope(dt, L(y.shape), y, L(x.shape), x, L(fcos.shape), fcos, L(fsin.shape), fsin, 1,
                int(z["n_head"]), int(z["start_pos"]))
        assert np.array_equal(y, z["y_" + tag])


@pytest.mark.parametrize("name", sorted(G.PROMPT_CASES))
def test_oracle_reproduces_prompt_golden(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg, weights, _, tokens, window = G.prompt_case(name)
    assert str(z["digest"]) == G.weights_digest(weights) and np.array_equal(tokens, z["tokens"])
    om = mo.Model(cfg, weights)
    tok, logits = om.forward(tokens, 0, window)
    assert tok == int(z["token"]) and np.array_equal(logits, z["logits"])
    for i in range(cfg["n_layers"]):
        assert np.array_equal(np.asarray(om.hidden(i)), z["hidden"][i])
    k, v = om.kv(cfg["n_layers"] - 1)
    assert np.array_equal(k, z["k_last"]) and np.array_equal(v, z["v_last"])
    t, pos = tok, len(tokens)
    for i in range(2):
        t, lg = om.step(t, pos)
        assert t == int(z["follow_tokens"][i]) and np.array_equal(lg, z["follow_logits"][i])
        pos += 1
    om.close()


def test_oracle_reproduces_sampler_golden():
    z = np.load(os.path.join(GOLD, "sampler.npz"))
    for dt, tag in ((1, "f32"), (0, "bf16")):
        tok, taps = mo.sample_default(dt, z["logits_" + tag], top_k=int(z["top_k"]), temperature=float(z["temperature"]),
                                      top_p=float(z["top_p"]), init_state=int(z["init_state"]), init_seq=int(z["init_seq"]),
                                      taps=True)
        assert tok == int(z["token_" + tag]) and np.array_equal(taps, z["taps_" + tag])
