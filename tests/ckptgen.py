"""Fabricates checkpoint files in the four layouts the reference's serializers read, from a
tests/modelgen.py weight dict.  Files are written with the `safetensors` Python package (an
implementation independent of model_io.cc), bf16 through torch."""
import json
import os

import numpy as np

from oracle import ckpt_oracle as ck

BF16, F32 = 0, 1
META, HF_LLAMA, META_QLORA, HF_GEMMA = 0, 1, 2, 3


def tie_head(weights):
    """Meta / HF checkpoints carry no output matrix: the head is the embedding table."""
    weights["output"] = dict(kind=0, weight=weights["embedding"]["weight"])
    return weights


def _save(path, tensors: dict, dt: int, metadata=None):
    """tensors: name -> numpy array; uint16 arrays are bf16 payloads."""
    import torch
    from safetensors.torch import save_file

    out = {}
    for k, v in tensors.items():
        t = torch.from_numpy(np.ascontiguousarray(v))
        if v.dtype == np.uint16:
            t = t.view(torch.bfloat16)
        out[k] = t
    save_file(out, path, metadata=metadata)


def tensors_for(weights, cfg, flavour):
    """name -> array in the on-disk convention of `flavour`."""
    H, KV = cfg["n_heads"], cfg["n_kv_heads"]
    t = {}
    hf = flavour in (HF_LLAMA, HF_GEMMA)
    P = "model." if hf else ""
    lin = (dict(wq="self_attn.q_proj", wk="self_attn.k_proj", wv="self_attn.v_proj", wo="self_attn.o_proj",
                w1="mlp.gate_proj", w2="mlp.down_proj", w3="mlp.up_proj") if hf else
           dict(wq="attention.wq", wk="attention.wk", wv="attention.wv", wo="attention.wo",
                w1="feed_forward.w1", w2="feed_forward.w2", w3="feed_forward.w3"))
    if flavour == HF_LLAMA:
        vec = dict(attention_norm="input_layernorm", ffn_norm="post_attention_layernorm")
    elif flavour == HF_GEMMA:
        vec = dict(attention_norm="input_layernorm", attention_post_norm="post_attention_layernorm",
                   ffn_norm="pre_feedforward_layernorm", ffn_post_norm="post_feedforward_layernorm",
                   q_norm="self_attn.q_norm", k_norm="self_attn.k_norm")
    else:
        vec = dict(attention_norm="attention_norm", ffn_norm="ffn_norm")
    for i, lw in enumerate(weights["layers"]):
        L = f"{P}layers.{i}."
        for n, path in lin.items():
            spec = lw[n]
            w = spec["weight"]
            if flavour == META and n in ("wq", "wk"):
                w = ck.unpermute_attention_heads(w, H if n == "wq" else KV)
            t[L + path + ".weight"] = w
            if spec["kind"] != 0:
                t[L + path + ".scales"] = spec["scales"]
                if spec.get("lora_a") is not None:
                    t[L + path + ".adaptor.A.weight"] = spec["lora_a"]
                    t[L + path + ".adaptor.B.weight"] = spec["lora_b"]
        for n, path in vec.items():
            t[L + path + ".weight"] = lw[n]
    t[P + ("embed_tokens" if hf else "tok_embeddings") + ".weight"] = weights["embedding"]["weight"]
    if weights["embedding"]["kind"] != 0:
        t["tok_embeddings.scales"] = weights["embedding"]["scales"].reshape(-1, 1)
    t[P + "norm.weight"] = weights["final_norm"]
    if flavour == META_QLORA:
        t["output.weight"] = weights["output"]["weight"]
        t["output.scales"] = weights["output"]["scales"].reshape(-1, 1)
    return t


def write_checkpoint(path, weights, cfg, flavour, shards: int = 1):
    t = tensors_for(weights, cfg, flavour)
    if shards == 1:
        _save(path, t, cfg["dtype"], metadata={"format": "pt"})
        return path
    names = list(t)
    base = os.path.dirname(path)
    weight_map = {}
    for s in range(shards):
        part = {n: t[n] for n in names[s::shards]}
        fn = f"model-{s + 1:05d}-of-{shards:05d}.safetensors"
        _save(os.path.join(base, fn), part, cfg["dtype"])
        weight_map.update({n: fn for n in part})
    index = os.path.join(base, "model.safetensors.index.json")
    with open(index, "w") as f:
        json.dump({"metadata": {"total_size": int(sum(v.nbytes for v in t.values()))}, "weight_map": weight_map}, f)
    return index


def options_json(cfg, flavour) -> str:
    if flavour in (META, META_QLORA):
        return json.dumps(dict(dim=cfg["dim"], n_layers=cfg["n_layers"], n_heads=cfg["n_heads"],
                               n_kv_heads=cfg["n_kv_heads"], vocab_size=cfg["vocab"], ffn_dim_multiplier=1.5,
                               multiple_of=256, norm_eps=cfg["norm_eps"], rope_theta=cfg["rope_theta"],
                               use_scaled_rope=True))
    o = dict(head_dim=cfg["head_dim"], hidden_size=cfg["dim"], intermediate_size=cfg["ffn_dim"],
             num_attention_heads=cfg["n_heads"], num_hidden_layers=cfg["n_layers"],
             num_key_value_heads=cfg["n_kv_heads"], rms_norm_eps=cfg["norm_eps"], rope_theta=cfg["rope_theta"],
             vocab_size=cfg["vocab"], model_type="llama", rope_scaling=None, tie_word_embeddings=True)
    if flavour == HF_GEMMA:
        o.update(model_type="gemma3_text", sliding_window=512, _sliding_window_pattern=cfg.get("sliding_stride", 0),
                 query_pre_attn_scalar=1.0 / cfg["attn_scale"] ** 2, rope_local_base_freq=cfg.get("rope_sliding_theta", 0.0))
    return json.dumps(o)
