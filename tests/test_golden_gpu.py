"""HIP path against the committed golden vectors (bytes on disk, not the oracle library): decode
steps through the C ABI -- including 8 tokens past max_seq_len (sink ring) and the QLoRA adaptors --
and the reference-named rope kernels.  Tolerances as in test_decode_gpu.py: f32 1e-4 relative to
(|ref| + rms), bf16 2e-3 with at most 2 scaled ulps; tokens and integer state exact."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as G  # noqa: E402
import modelgen as mg  # noqa: E402
import parity  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
BF16, F32 = 0, 1


@pytest.mark.parametrize("name", sorted(G.DECODE_CASES))
def test_decode_matches_golden(acc, name):
    import metalchat_amd as mc

    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg, weights, dk, steps = G.decode_case(name)
    assert str(z["digest"]) == G.weights_digest(weights)
    dt = cfg["dtype"]
    rel = 1e-4 if dt == F32 else (5e-3 if steps > cfg["max_seq_len"] else 2e-3)
    frac = 1.0 if dt == F32 else 0.5
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, **dk))
    dec.load_model(weights)
    dec.set_taps(True)
    tok, agree = int(z["first_token"]), 0
    for pos in range(steps):
        got = dec.step(tok, pos)
        for i, layer in enumerate(range(-1, cfg["n_layers"])):
            parity.check(dt, dec.hidden(layer), z["hidden"][pos][i], rel=rel, max_ulp=2 if layer >= 0 else 0,
                         max_frac=frac if layer >= 0 else 0.0, what=f"{name} pos {pos} hidden[{layer}]")
        parity.check(dt, dec.logits(), z["logits"][pos], rel=rel, max_ulp=2, max_frac=frac,
                     what=f"{name} pos {pos} logits")
        agree += int(got == int(z["tokens"][pos]))
        tok = int(z["tokens"][pos])
    for layer, kk, vk in ((0, "k_first", "v_first"), (cfg["n_layers"] - 1, "k_last", "v_last")):
        gk, gv = dec.export_kv(layer)
        assert gk.shape == z[kk].shape
        parity.check(dt, gk, z[kk], rel=rel, max_ulp=2, max_frac=frac, what=f"{name} K[{layer}]")
        parity.check(dt, gv, z[vk], rel=rel, max_ulp=2, max_frac=frac, what=f"{name} V[{layer}]")
    assert agree >= steps - (0 if dt == F32 else 2)
    dec.release()


@pytest.mark.parametrize("dt,tag", [(F32, "f32"), (BF16, "bf16")])
def test_rope_kernels_match_golden(acc, dt, tag):
    import metalchat_amd as mc
    from metalchat_amd import layout as L

    z = np.load(os.path.join(GOLD, "rope.npz"))
    hd, n_head = int(z["hd"]), int(z["n_head"])
    rows = z["fcos"].shape[0]
    k = acc.load("rope_freqs", "float")
    cb, sb = acc.alloc(rows * hd // 2 * 4), acc.alloc(rows * hd // 2 * 4)
    grid, thread = mc.make_kernel_grid_2d(rows, hd // 2, k.max_threads_per_threadgroup())
    mc.KernelTask(k, grid, thread, [L((rows, hd // 2)), cb, L((rows, hd // 2)), sb, np.uint32(hd),
                                    np.uint32(int(z["table_start"])), np.float32(float(z["theta"]))])()
    acc.wait()
    parity.exact(cb.download(np.float32, rows * hd // 2).reshape(rows, -1), z["fcos"], "golden rope_freqs cos")
    parity.exact(sb.download(np.float32, rows * hd // 2).reshape(rows, -1), z["fsin"], "golden rope_freqs sin")
    x = z["x_" + tag]
    k = acc.load("rope", "float" if dt == F32 else "bfloat")
    grid, thread = mc.make_kernel_grid_2d(x.shape[0], hd, k.max_threads_per_threadgroup())
    out = acc.alloc(x.size * 4)
    mc.KernelTask(k, grid, thread, [L(x.shape), out, L(x.shape), acc.to_device(x), L((rows, hd // 2)), cb,
                                    L((rows, hd // 2)), sb, np.uint32(1), np.uint32(n_head),
                                    np.uint32(int(z["start_pos"]))])()
    acc.wait()
    parity.check(dt, out.download(x.dtype, x.size), z["y_" + tag].reshape(-1),
                 rel=1e-6 if dt == F32 else 1e-3, max_ulp=1, max_frac=0.01, scale_aware=False, what="golden rope")


@pytest.mark.parametrize("name", sorted(G.PROMPT_CASES))
def test_prompt_pass_matches_golden(acc, name):
    import metalchat_amd as mc

    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg, weights, dk, tokens, window = G.prompt_case(name)
    assert str(z["digest"]) == G.weights_digest(weights)
    dt = cfg["dtype"]
    rel, frac = (1e-4, 1.0) if dt == F32 else (7.8e-3, 0.7)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, **dk))
    dec.load_model(weights)
    dec.set_taps(True)
    got = dec.prefill(tokens, 0, window)
    for i in range(cfg["n_layers"]):
        parity.check(dt, dec.hidden(i), z["hidden"][i], rel=rel, max_ulp=2, max_frac=frac, what=f"{name} hidden[{i}]")
    parity.check(dt, dec.logits(), z["logits"], rel=rel, max_ulp=2, max_frac=frac, what=f"{name} logits")
    gk, gv = dec.export_kv(cfg["n_layers"] - 1)
    parity.check(dt, gk, z["k_last"], rel=rel, max_ulp=2, max_frac=frac, what=f"{name} K")
    parity.check(dt, gv, z["v_last"], rel=rel, max_ulp=2, max_frac=frac, what=f"{name} V")
    agree = int(got == int(z["token"]))
    t, pos = int(z["token"]), len(tokens)
    for i in range(2):
        g = dec.step(t, pos)
        parity.check(dt, dec.logits(), z["follow_logits"][i], rel=rel, max_ulp=2, max_frac=frac, what=f"{name} follow {i}")
        agree += int(g == int(z["follow_tokens"][i]))
        t, pos = int(z["follow_tokens"][i]), pos + 1
    assert agree >= (3 if dt == F32 else 2)
    dec.release()


@pytest.mark.parametrize("dt,tag", [(F32, "f32"), (BF16, "bf16")])
def test_fused_sampler_matches_golden(acc, dt, tag):
    from test_sampler_gpu import fused_sample

    z = np.load(os.path.join(GOLD, "sampler.npz"))
    tok, taps = fused_sample(acc, dt, z["logits_" + tag], top_k=int(z["top_k"]), temperature=float(z["temperature"]),
                             top_p=float(z["top_p"]), seed=(int(z["init_state"]), int(z["init_seq"])))
    assert tok == int(z["token_" + tag])
    parity.exact(taps, z["taps_" + tag], "sampler chain intermediates")
