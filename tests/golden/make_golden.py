"""Generates the golden vectors under tests/golden/ from the CPU oracle (oracle/mc_oracle.c).

PARITY UNPINNED BY THE REFERENCE for everything in here: the reference has no test of the rope
rotation kernel, of the sink_cache roll branch or of a whole decode step, and it
cannot be built or run in this image, so these vectors are outputs of this repository's own
restatement, frozen so that (a) the oracle cannot drift silently and (b) the HIP path is checked
against bytes on disk and not only against code that lives next to it.  The kernels the reference
does test are pinned against its own known answers in tests/test_oracle_reference_pins.py.

    python tests/golden/make_golden.py        # rewrites the .npz files (deterministic)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import modelgen as mg  # noqa: E402
from oracle import mc_oracle as mo  # noqa: E402

BF16, F32 = 0, 1

DECODE_CASES = {
    # name: (cfg overrides, make_model kwargs, decoder kwargs, steps)
    "decode_llama_f32_i4g32": (dict(dtype=F32, max_seq_len=16), dict(seed=101, quant="i4", group=32),
                               dict(weight_format=2, group_size=32), 24),
    "decode_llama_bf16_i4g32": (dict(dtype=BF16, max_seq_len=16), dict(seed=102, quant="i4", group=32),
                                dict(weight_format=2, group_size=32), 24),
    "decode_llama_f32_qlora": (dict(dtype=F32, max_seq_len=32), dict(seed=103, quant="i4", group=32, lora_rank=16),
                               dict(weight_format=2, group_size=32), 8),
    "decode_gemma3_f32_i4g32": (dict(dtype=F32, family=1, n_layers=3, rope_sliding_theta=10000.0,
                                     sliding_stride=2, attn_scale=float(1.0 / np.sqrt(48.0))),
                                dict(seed=104, quant="i4", group=32), dict(weight_format=2, group_size=32), 8),
}


def weights_digest(weights) -> str:
    h = hashlib.sha256()

    def feed(o):
        if isinstance(o, dict):
            for k in sorted(o):
                h.update(k.encode())
                feed(o[k])
        elif isinstance(o, (list, tuple)):
            for v in o:
                feed(v)
        elif isinstance(o, np.ndarray):
            h.update(np.ascontiguousarray(o).tobytes())
        elif o is not None:
            h.update(repr(o).encode())

    feed(weights)
    return h.hexdigest()


def decode_case(name):
    over, mk, dk, steps = DECODE_CASES[name]
    cfg = mg.tiny_cfg(**over)
    weights = mg.make_model(cfg, **mk)
    return cfg, weights, dk, steps


def make_decode(name):
    cfg, weights, dk, steps = decode_case(name)
    om = mo.Model(cfg, weights)
    tok = 3
    toks, logits, hidden = [], [], []
    for pos in range(steps):
        nt, lg = om.step(tok, pos)
        toks.append(nt)
        logits.append(np.array(lg, copy=True))
        hidden.append(np.stack([np.array(om.hidden(l), copy=True) for l in range(-1, cfg["n_layers"])]))
        tok = nt
    k0, v0 = om.kv(0)
    kl, vl = om.kv(cfg["n_layers"] - 1)
    om.close()
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        cfg=json.dumps(cfg), digest=weights_digest(weights), first_token=3,
                        tokens=np.array(toks, np.int32), logits=np.stack(logits), hidden=np.stack(hidden),
                        k_first=k0, v_first=v0, k_last=kl, v_last=vl)


def make_rope():
    """kernel::rope_freqs + kernel::rope (kernel/rope.metal:29-63,77-102) on [1, 3, 4, 32]."""
    rng = np.random.default_rng(7)
    out = {}
    hd, n_head, len_, start = 32, 4, 3, 5
    L = mo.layout
    fcos = np.zeros((16, hd // 2), np.float32)
    fsin = np.zeros_like(fcos)
    mo.rope_freqs(L(fcos.shape), fcos, L(fsin.shape), fsin, hd, 2, 500000.0)
    out["fcos"], out["fsin"] = fcos, fsin
    for dt, tag in ((F32, "f32"), (BF16, "bf16")):
        x = mo.encode(dt, rng.normal(0, 1, (len_ * n_head, hd)).astype(np.float32))
        y = np.zeros_like(x)
        mo.rope(dt, L(y.shape), y, L(x.shape), x, L(fcos.shape), fcos, L(fsin.shape), fsin, 1, n_head, start)
        out["x_" + tag], out["y_" + tag] = x, y
    np.savez_compressed(os.path.join(HERE, "rope.npz"), hd=hd, n_head=n_head, table_start=2, start_pos=start,
                        theta=500000.0, **out)


PROMPT_CASES = {
    # name: (cfg overrides, make_model kwargs, decoder kwargs, prompt length, sliding window)
    "prompt_llama_f32_i4g32": (dict(dtype=F32, max_seq_len=48), dict(seed=111, quant="i4", group=32),
                               dict(weight_format=2, group_size=32), 21, 0),
    "prompt_gemma3_bf16_i4g32": (dict(dtype=BF16, family=1, n_layers=3, rope_sliding_theta=10000.0, sliding_stride=2,
                                      attn_scale=float(1.0 / np.sqrt(48.0)), max_seq_len=48),
                                 dict(seed=112, quant="i4", group=32), dict(weight_format=2, group_size=32), 30, 5),
}


def prompt_case(name):
    over, mk, dk, n, window = PROMPT_CASES[name]
    cfg = mg.tiny_cfg(**over)
    weights = mg.make_model(cfg, **mk)
    tokens = np.random.default_rng(len(name)).integers(0, cfg["vocab"], n).astype(np.int32)
    return cfg, weights, dk, tokens, window


def make_prompt(name):
    """mco_model_forward (the prompt pass, nn/llama.h:113-134) + two decode steps behind it."""
    cfg, weights, dk, tokens, window = prompt_case(name)
    om = mo.Model(cfg, weights)
    tok, logits = om.forward(tokens, 0, window)
    hidden = np.stack([np.array(om.hidden(l), copy=True) for l in range(0, cfg["n_layers"])])
    k, v = om.kv(cfg["n_layers"] - 1)
    follow_tok, follow_logits = [], []
    t, pos = tok, len(tokens)
    for _ in range(2):
        t, lg = om.step(t, pos)
        follow_tok.append(t)
        follow_logits.append(np.array(lg, copy=True))
        pos += 1
    om.close()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), cfg=json.dumps(cfg), digest=weights_digest(weights),
                        tokens=tokens, window=window, token=tok, logits=logits, hidden=hidden, k_last=k, v_last=v,
                        follow_tokens=np.array(follow_tok, np.int32), follow_logits=np.stack(follow_logits))


def make_sampler():
    """make_default_sampler (nn/sampling.h:303-313) on seeded logit rows: the seven intermediates of
    the chain and the sampled id, for both dtypes, with exact ties in the bf16 row."""
    rng = np.random.default_rng(21)
    out = {}
    for dt, tag in ((F32, "f32"), (BF16, "bf16")):
        x = rng.normal(0, 2.0, 4096).astype(np.float32)
        if dt == BF16:
            x = np.round(x * 2) / 2
        logits = mo.encode(dt, x)
        tok, taps = mo.sample_default(dt, logits, top_k=50, temperature=0.6, top_p=0.9, init_state=7, init_seq=9, taps=True)
        out["logits_" + tag], out["taps_" + tag], out["token_" + tag] = logits, taps, tok
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), top_k=50, temperature=0.6, top_p=0.9, init_state=7, init_seq=9, **out)


if __name__ == "__main__":
    for n in DECODE_CASES:
        make_decode(n)
    for n in PROMPT_CASES:
        make_prompt(n)
    make_sampler()
    make_rope()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
