"""Seeded synthetic models in the REFERENCE-NATIVE formats (T weights, or int8-held quantised
weights + f32 scales), consumed by both the CPU oracle and the HIP decoder's load_* entry points.
Shapes follow SURVEY.md section 8; sizes used in tests are scaled down."""
from __future__ import annotations

import numpy as np

from oracle import mc_oracle as mo

BF16, F32 = 0, 1
WFMT_T, WFMT_I8, WFMT_I4 = 0, 1, 2


def _enc(dt, x):
    return mo.encode(dt, np.asarray(x, dtype=np.float32))


def make_linear(rng, dt, out_f, in_f, quant, group):
    """quant: None -> nn::linear (T); "i4"/"i8" -> lora_linear-style per-(row,group) scales;
    "i8row" -> quantization::linear (per-row scale)."""
    if quant is None:
        w = rng.uniform(-1.0, 1.0, size=(out_f, in_f)).astype(np.float32) / np.sqrt(in_f)
        return dict(kind=0, weight=_enc(dt, w))
    if quant == "i4":
        q = rng.integers(-8, 8, size=(out_f, in_f), dtype=np.int8)
        qmax, fmt = 8.0, WFMT_I4
    else:
        q = rng.integers(-128, 128, size=(out_f, in_f), dtype=np.int8)
        qmax, fmt = 128.0, WFMT_I8
    if quant == "i8row":
        s = (rng.uniform(0.5, 1.5, size=(out_f, 1)) / (np.sqrt(in_f) * qmax)).astype(np.float32)
        return dict(kind=2, weight=q, scales=s, group_size=0, hbm_format=WFMT_I8)
    ng = in_f // group
    s = (rng.uniform(0.5, 1.5, size=(out_f, ng)) / (np.sqrt(in_f) * qmax)).astype(np.float32)
    return dict(kind=1, weight=q, scales=s, group_size=group, hbm_format=fmt)


def add_lora(rng, dt, spec, rank, scale=2.0):
    """Attach a quantization::lora_adaptor (A [rank, in], B [out, rank], lora.h:17-53) to a
    lora_linear spec; sized so that the adaptation is comparable to the base output."""
    out_f, in_f = spec["weight"].shape
    spec["lora_a"] = _enc(dt, rng.uniform(-1.0, 1.0, size=(rank, in_f)) / np.sqrt(in_f))
    spec["lora_b"] = _enc(dt, rng.uniform(-1.0, 1.0, size=(out_f, rank)) * (0.25 / np.sqrt(rank)))
    spec["lora_scale"] = float(scale)
    return spec


def make_model(cfg: dict, seed: int = 0, quant=None, group: int = 32, emb_quant: bool = False,
               head_quant=None, lora_rank: int = 0, lora_only=None):
    """cfg keys: dtype, family, dim, n_heads, n_kv_heads, head_dim, ffn_dim, n_layers, vocab,
    max_seq_len, rope_theta, norm_eps, attn_scale [, rope_sliding_theta, sliding_stride]."""
    rng = np.random.default_rng(seed)
    dt = cfg["dtype"]
    dim, H, KV, hd, ffn = cfg["dim"], cfg["n_heads"], cfg["n_kv_heads"], cfg["head_dim"], cfg["ffn_dim"]
    gemma = cfg.get("family", 0) == 1
    layers = []
    for i in range(cfg["n_layers"]):
        lw = dict(
            wq=make_linear(rng, dt, H * hd, dim, quant, group),
            wk=make_linear(rng, dt, KV * hd, dim, quant, group),
            wv=make_linear(rng, dt, KV * hd, dim, quant, group),
            wo=make_linear(rng, dt, dim, H * hd, quant, group),
            w1=make_linear(rng, dt, ffn, dim, quant, group),
            w2=make_linear(rng, dt, dim, ffn, quant, group),
            w3=make_linear(rng, dt, ffn, dim, quant, group),
            attention_norm=_enc(dt, rng.uniform(0.5, 1.5, dim)),
            ffn_norm=_enc(dt, rng.uniform(0.5, 1.5, dim)),
        )
        if lora_rank and quant in ("i4", "i8"):
            for n in ("wq", "wk", "wv", "wo", "w1", "w2", "w3"):
                if lora_only is None or n in lora_only:
                    add_lora(rng, dt, lw[n], lora_rank)
        if gemma:
            # gemma norm weights are used as (1 + w): keep w small
            for n, sz in (("attention_norm", dim), ("ffn_norm", dim), ("q_norm", hd), ("k_norm", hd),
                          ("attention_post_norm", dim), ("ffn_post_norm", dim)):
                lw[n] = _enc(dt, rng.uniform(-0.25, 0.25, sz))
            stride = cfg.get("sliding_stride", 0)
            lw["rope_table"] = 1 if (stride and (i + 1) % stride != 0
                                     and cfg.get("rope_sliding_theta", 0.0) > 0) else 0
        layers.append(lw)
    if emb_quant:
        q = rng.integers(-128, 128, size=(cfg["vocab"], dim), dtype=np.int8)
        s = (rng.uniform(0.5, 1.5, size=(cfg["vocab"],)) * (0.02 / 64.0)).astype(np.float32)
        emb = dict(kind=2, weight=q, scales=s)
    else:
        emb = dict(kind=0, weight=_enc(dt, rng.normal(0, 1.0, size=(cfg["vocab"], dim)) * 0.5))
    hq = head_quant if head_quant is not None else quant
    output = make_linear(rng, dt, cfg["vocab"], dim, hq, group)
    fn = rng.uniform(-0.25, 0.25, dim) if gemma else rng.uniform(0.5, 1.5, dim)
    return dict(layers=layers, embedding=emb, output=output, final_norm=_enc(dt, fn))


def tiny_cfg(dtype=BF16, **over):
    c = dict(dtype=dtype, family=0, dim=256, n_heads=8, n_kv_heads=2, head_dim=32, ffn_dim=512,
             n_layers=2, vocab=512, max_seq_len=32, rope_theta=500000.0, norm_eps=1e-5,
             attn_scale=1.0 / np.sqrt(32.0))
    c.update(over)
    if "attn_scale" not in over:
        c["attn_scale"] = float(1.0 / np.sqrt(c["head_dim"]))
    return c


def decoder_kwargs(cfg: dict, **over):
    k = {key: cfg[key] for key in ("dtype", "dim", "n_heads", "n_kv_heads", "head_dim", "ffn_dim",
                                   "n_layers", "vocab", "max_seq_len", "rope_theta", "norm_eps",
                                   "attn_scale")}
    k["family"] = cfg.get("family", 0)
    k["rope_sliding_theta"] = cfg.get("rope_sliding_theta", 0.0)
    k["sliding_stride"] = cfg.get("sliding_stride", 0)
    k["sink_pre_len"] = cfg.get("sink_pre_len", -1)
    k.update(over)
    return k
