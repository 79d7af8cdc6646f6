"""CPU restatement (numpy / re) of the reference's checkpoint adaptors and option serializers.

TEST INFRASTRUCTURE ONLY, like the rest of oracle/: the product path is model_io.cc behind the C
ABI; only tests/ import this file, as the checker.

Pinned by the reference's own tests where they exist: the two option-serializer known answers
(test/test_huggingface.cc:41-86, test/test_reference.cc:17-45), the adapted-name property
`!name.starts_with("model")` and the count 147 for the 16-layer HF checkpoint
(test/test_huggingface.cc:19-38), "tensor link" and "sharded document" (test/test_safetensor.cc:
150-213).  PARITY UNPINNED: permute_attention_heads has no reference test; the restatement below
follows include/metalchat/nn/attention.h:225-254 literally.
"""
import json
import re

import numpy as np

# include/metalchat/huggingface/llama.h:88-100 -- every rule is applied to every name in turn
# (safetensor_document::rename, include/metalchat/safetensor.h:835-852)
HF_LLAMA_MAPPING = [
    (r"model\.(layers\.\d+)\.input_layernorm", r"\1.attention_norm"),
    (r"model\.(layers\.\d+)\.post_attention_layernorm", r"\1.ffn_norm"),
    (r"model\.(layers\.\d+)\.mlp\.gate_proj", r"\1.feed_forward.w1"),
    (r"model\.(layers\.\d+)\.mlp\.down_proj", r"\1.feed_forward.w2"),
    (r"model\.(layers\.\d+)\.mlp\.up_proj", r"\1.feed_forward.w3"),
    (r"model\.(layers\.\d+)\.self_attn\.q_proj", r"\1.attention.wq"),
    (r"model\.(layers\.\d+)\.self_attn\.k_proj", r"\1.attention.wk"),
    (r"model\.(layers\.\d+)\.self_attn\.v_proj", r"\1.attention.wv"),
    (r"model\.(layers\.\d+)\.self_attn\.o_proj", r"\1.attention.wo"),
    (r"model.norm", "norm"),
    (r"model.embed_tokens", "tok_embeddings"),
]

# include/metalchat/huggingface/gemma.h:59-77
HF_GEMMA_MAPPING = [
    (r"model\.(layers\.\d+)\.input_layernorm", r"\1.attention_norm"),
    (r"model\.(layers\.\d+)\.post_attention_layernorm", r"\1.attention_post_norm"),
    (r"model\.(layers\.\d+)\.pre_feedforward_layernorm", r"\1.ffn_norm"),
    (r"model\.(layers\.\d+)\.post_feedforward_layernorm", r"\1.ffn_post_norm"),
    (r"model\.(layers\.\d+)\.mlp\.gate_proj", r"\1.feed_forward.w1"),
    (r"model\.(layers\.\d+)\.mlp\.down_proj", r"\1.feed_forward.w2"),
    (r"model\.(layers\.\d+)\.mlp\.up_proj", r"\1.feed_forward.w3"),
    (r"model\.(layers\.\d+)\.self_attn\.q_proj", r"\1.attention.wq"),
    (r"model\.(layers\.\d+)\.self_attn\.q_norm", r"\1.attention.q_norm"),
    (r"model\.(layers\.\d+)\.self_attn\.k_proj", r"\1.attention.wk"),
    (r"model\.(layers\.\d+)\.self_attn\.k_norm", r"\1.attention.k_norm"),
    (r"model\.(layers\.\d+)\.self_attn\.v_proj", r"\1.attention.wv"),
    (r"model\.(layers\.\d+)\.self_attn\.o_proj", r"\1.attention.wo"),
    (r"model.norm", "norm"),
    (r"model.embed_tokens", "tok_embeddings"),
]


def adapt_names(names, mapping):
    """serializer.adapt(document): renamed entries in document order, then the linked output head
    (huggingface/llama.h:102-104)."""
    out = []
    for n in names:
        for pat, rep in mapping:
            n = re.sub(pat, rep, n)
        out.append(n)
    return out + ["output.weight"]


def permute_attention_heads(w: np.ndarray, n_heads: int) -> np.ndarray:
    """include/metalchat/nn/attention.h:225-254: input row (i, j, k) of the view
    [n_heads, hd/2, 2] goes to output row i*hd + k*(hd/2) + j."""
    size = w.shape[0]
    attention_heads = size // n_heads // 2
    out = np.empty_like(w)
    for idx in range(size):
        i, rem = divmod(idx, attention_heads * 2)
        j, k = divmod(rem, 2)
        out[i * attention_heads * 2 + k * attention_heads + j] = w[idx]
    return out


def unpermute_attention_heads(w: np.ndarray, n_heads: int) -> np.ndarray:
    """Inverse (what a Meta checkpoint holds for a half-split weight): used to fabricate inputs."""
    hd = w.shape[0] // n_heads
    return w.reshape(n_heads, 2, hd // 2, *w.shape[1:]).swapaxes(1, 2).reshape(w.shape)


def options_hf_llama(text: str) -> dict:
    """src/llama.cc:41-55"""
    o = json.loads(text)
    return dict(head_dim=o["head_dim"], n_heads=o["num_attention_heads"], n_kv_heads=o["num_key_value_heads"],
                n_layers=o["num_hidden_layers"], max_seq_len=1024, rope_theta=float(o["rope_theta"]),
                norm_eps=float(o["rms_norm_eps"]))


def options_meta_llama(text: str) -> dict:
    """src/reference.cc:52-66"""
    o = json.loads(text)
    return dict(head_dim=o["dim"] // o["n_heads"], n_heads=o["n_heads"], n_kv_heads=o["n_kv_heads"],
                n_layers=o["n_layers"], max_seq_len=1024, rope_theta=float(o["rope_theta"]),
                norm_eps=float(o["norm_eps"]))


def options_hf_gemma(text: str) -> dict:
    """src/gemma.cc:20-42 (only "_sliding_window_pattern" is consulted, as in the reference)"""
    o = json.loads(text)
    return dict(head_dim=o["head_dim"], hidden_dim=o["hidden_size"], n_heads=o["num_attention_heads"],
                n_kv_heads=o["num_key_value_heads"], n_layers=o["num_hidden_layers"], max_seq_len=1024,
                sliding_window=o.get("sliding_window", 0), sliding_stride=o.get("_sliding_window_pattern") or 0,
                attn_scale=float(o["query_pre_attn_scalar"]), rope_theta=float(o["rope_theta"]),
                rope_sliding_theta=float(o["rope_local_base_freq"]), norm_eps=float(o["rms_norm_eps"]))
