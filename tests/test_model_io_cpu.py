"""Part 3 of the C ABI on the CPU (no kernels involved): safetensor_document semantics against the
reference's own tests (test/test_safetensor.cc, test/test_huggingface.cc, test/test_reference.cc)
and against the independent `safetensors` Python package; checkpoint adaptors against the numpy
restatement in oracle/ckpt_oracle.py."""
import json
import os
import struct

import numpy as np
import pytest

import ckptgen as cg
import modelgen as mg
from oracle import ckpt_oracle as ck

BF16, F32 = 0, 1


@pytest.fixture(scope="module")
def mc():
    import metalchat_amd

    return metalchat_amd


def test_file_written_by_safetensors_package_is_read_bytewise(mc, tmp_path):
    import torch
    from safetensors.torch import save_file

    rng = np.random.default_rng(0)
    src = {"b.weight": rng.normal(size=(10, 20)).astype(np.float32),
           "a.weight": rng.integers(0, 65535, size=(3, 4)).astype(np.uint16),    # bf16 payload
           "c.q": rng.integers(-128, 127, size=(7, 32)).astype(np.int8),
           "scalar": np.array(3, np.int32), "empty": np.zeros((0, 4), np.float32)}
    tt = {k: (torch.from_numpy(v).view(torch.bfloat16) if v.dtype == np.uint16 else torch.from_numpy(v))
          for k, v in src.items()}
    p = tmp_path / "m.safetensors"
    save_file(tt, str(p), metadata={"format": "pt", "note": "héllo \"quoted\""})
    d = mc.Document(p)
    assert len(d) == len(src)
    assert d.metadata("format") == "pt" and d.metadata("note") == "héllo \"quoted\"" and d.metadata("nope") is None
    exp_dtype = {"b.weight": "F32", "a.weight": "BF16", "c.q": "I8", "scalar": "I32", "empty": "F32"}
    prev = -1
    for i in range(len(d)):
        t = d.tensor(i)
        assert t["dtype"] == exp_dtype[t["name"]] and t["shape"] == src[t["name"]].shape
        assert np.array_equal(t["data"], src[t["name"]])
        if t["nbytes"]:
            assert t["address"] > prev   # ascending file offsets: src/safetensor.cc:111-115
            prev = t["address"]
    assert np.array_equal(d.find("c.q")["data"], src["c.q"])
    with pytest.raises(mc.McError, match="is not in the document"):
        d.find("missing")


def test_write_and_read_small_model(mc, tmp_path):
    # test/test_safetensor.cc:94-147: linear1.weight f32 [10,20], linear2.weight bf16 [3,4]
    from safetensors import safe_open

    rng = np.random.default_rng(1)
    w1 = rng.uniform(size=(10, 20)).astype(np.float32)
    w2 = (rng.uniform(size=(3, 4)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    out = mc.Document()
    out.insert("linear1.weight", w1)
    out.insert("linear2.weight", w2, dtype="BF16")
    p = tmp_path / "model.st"
    out.save(p)
    assert p.exists()
    back = mc.Document(p)
    assert np.array_equal(back.find("linear1.weight")["data"], w1)
    assert np.array_equal(back.find("linear2.weight")["data"], w2) and back.find("linear2.weight")["dtype"] == "BF16"
    # and the independent reader agrees with what was written
    with safe_open(str(p), framework="pt") as f:
        import torch

        assert set(f.keys()) == {"linear1.weight", "linear2.weight"}
        assert np.array_equal(f.get_tensor("linear1.weight").numpy(), w1)
        assert np.array_equal(f.get_tensor("linear2.weight").view(torch.uint16).numpy(), w2)


def test_tensor_link(mc):
    # test/test_safetensor.cc:150-164
    a = np.random.default_rng(2).uniform(size=(3, 4)).astype(np.float32)
    d = mc.Document()
    d.insert("input.weight", a)
    d.link("output.weight", "input.weight")
    o, i = d.find("output.weight"), d.find("input.weight")
    assert o["shape"] == (3, 4) and o["address"] == i["address"]      # same container
    with pytest.raises(mc.McError):
        d.link("x", "absent")


def test_sharded_document(mc, tmp_path):
    # test/test_safetensor.cc:167-213: two files, one tensor each, absolute paths in weight_map
    rng = np.random.default_rng(3)
    t1, t2 = rng.uniform(size=(4, 3)).astype(np.float32), rng.uniform(size=(10, 6)).astype(np.float32)
    p1, p2 = tmp_path / "tensors-0001-of-0002.safetensors", tmp_path / "tensors-0002-of-0002.safetensors"
    for p, n, t in ((p1, "tensor1", t1), (p2, "tensor2", t2)):
        d = mc.Document()
        d.insert(n, t)
        d.save(p)
    idx = tmp_path / "tensors.safetensors.index.json"
    idx.write_text(json.dumps({"metadata": {}, "weight_map": {"tensor1": str(p1), "tensor2": str(p2)}}))
    doc = mc.Document(idx, sharded=True)
    assert len(doc) == 2
    assert np.array_equal(doc.find("tensor2")["data"], t2)
    # relative file names (what HF writes) resolve against the index's directory; a file that
    # holds several tensors is opened once
    idx.write_text(json.dumps({"metadata": {}, "weight_map": {"tensor1": p1.name, "tensor2": p2.name, "again": p1.name}}))
    assert len(mc.Document(idx, sharded=True)) == 2


def test_corrupt_files_are_runtime_errors(mc, tmp_path):
    # src/safetensor.cc:88-109
    p = tmp_path / "short.safetensors"
    p.write_bytes(b"\x01\x02\x03")
    with pytest.raises(mc.McError, match="header size is corrupted") as e:
        mc.Document(p)
    assert e.value.status == 2
    p.write_bytes(struct.pack("<Q", 1000) + b"{}")
    with pytest.raises(mc.McError, match="header is corrupted"):
        mc.Document(p)
    hdr = json.dumps({"t": {"dtype": "F32", "shape": [4], "data_offsets": [0, 16]}}).encode()
    p.write_bytes(struct.pack("<Q", len(hdr)) + hdr + b"\0" * 8)     # data shorter than declared
    with pytest.raises(mc.McError, match="unable to read tensor of size 16"):
        mc.Document(p)
    hdr = json.dumps({"t": {"dtype": "F32", "shape": [3], "data_offsets": [0, 16]}}).encode()
    p.write_bytes(struct.pack("<Q", len(hdr)) + hdr + b"\0" * 16)
    with pytest.raises(mc.McError, match="disagree"):
        mc.Document(p)
    with pytest.raises(mc.McError, match="unable to open"):
        mc.Document(tmp_path / "nope.safetensors")


def test_hf_llama_adaptor_names_and_count(mc, tmp_path):
    # test/test_huggingface.cc:19-38: no adapted name starts with "model"; 16 layers -> 147 entries
    names = ["model.embed_tokens.weight", "model.norm.weight"]
    for i in range(16):
        for n in ("input_layernorm", "post_attention_layernorm", "mlp.gate_proj", "mlp.down_proj", "mlp.up_proj",
                  "self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj"):
            names.append(f"model.layers.{i}.{n}.weight")
    d = mc.Document()
    for k, n in enumerate(names):
        d.insert(n, np.full((2, 2), k, np.float32))
    d.adapt(mc.CKPT_HF_LLAMA3)
    got = d.names()
    assert len(got) == 147 and not any(n.startswith("model") for n in got)
    assert got == ck.adapt_names(names, ck.HF_LLAMA_MAPPING)
    assert d.find("output.weight")["address"] == d.find("tok_embeddings.weight")["address"]
    assert d.find("layers.7.feed_forward.w2.weight")["data"][0, 0] == names.index("model.layers.7.mlp.down_proj.weight")


def test_hf_gemma_adaptor_names(mc):
    names = ["model.embed_tokens.weight", "model.norm.weight"]
    for i in range(3):
        for n in ("input_layernorm", "post_attention_layernorm", "pre_feedforward_layernorm",
                  "post_feedforward_layernorm", "mlp.gate_proj", "mlp.down_proj", "mlp.up_proj", "self_attn.q_proj",
                  "self_attn.q_norm", "self_attn.k_proj", "self_attn.k_norm", "self_attn.v_proj", "self_attn.o_proj"):
            names.append(f"model.layers.{i}.{n}.weight")
    d = mc.Document()
    for n in names:
        d.insert(n, np.zeros((1,), np.float32))
    d.adapt(mc.CKPT_HF_GEMMA3)
    assert d.names() == ck.adapt_names(names, ck.HF_GEMMA_MAPPING)
    assert "layers.1.attention_post_norm.weight" in d.names() and "layers.2.attention.k_norm.weight" in d.names()


HF_LLAMA_JSON = """{
  "attention_bias": false, "attention_dropout": 0.0, "head_dim": 64, "hidden_act": "silu", "hidden_size": 2048,
  "initializer_range": 0.02, "intermediate_size": 8192, "max_position_embeddings": 131072, "mlp_bias": false,
  "model_type": "llama", "num_attention_heads": 32, "num_hidden_layers": 16, "num_key_value_heads": 8,
  "pretraining_tp": 1, "rms_norm_eps": 1e-05,
  "rope_scaling": {"factor": 32.0, "high_freq_factor": 4.0, "low_freq_factor": 1.0,
                   "original_max_position_embeddings": 8192, "rope_type": "llama3"},
  "rope_theta": 500000.0, "use_cache": true, "vocab_size": 128256
}"""

META_JSON = """{"dim": 2048, "n_layers": 16, "n_heads": 32, "n_kv_heads": 8, "vocab_size": 128256,
  "ffn_dim_multiplier": 1.5, "multiple_of": 256, "norm_eps": 1e-05, "rope_theta": 500000.0, "use_scaled_rope": true}"""


def test_llama3_options_serializers_known_answers(mc):
    # test/test_huggingface.cc:41-86 and test/test_reference.cc:17-45 (same expected values)
    for text, flavour, oracle in ((HF_LLAMA_JSON, mc.CKPT_HF_LLAMA3, ck.options_hf_llama),
                                  (META_JSON, mc.CKPT_META_LLAMA3, ck.options_meta_llama)):
        c = mc.config_from_json(text, flavour)
        assert (c.head_dim, c.n_layers, c.n_heads, c.n_kv_heads, c.max_seq_len) == (64, 16, 32, 8, 1024)
        assert abs(c.rope_theta - 500000.0) <= 0.01 * 500000.0 and abs(c.norm_eps - 1e-5) <= 0.01 * 1e-5
        o = oracle(text)
        assert (c.head_dim, c.n_layers, c.n_heads, c.n_kv_heads, c.max_seq_len) == \
            (o["head_dim"], o["n_layers"], o["n_heads"], o["n_kv_heads"], o["max_seq_len"])
        assert c.vocab == 128256 and c.family == mc.FAMILY_LLAMA3
        assert abs(c.attn_scale - 0.125) < 1e-7                        # 1/sqrt(head_dim), nn/llama.h:88
    with pytest.raises(mc.McError) as e:
        mc.config_from_json("{not json", mc.CKPT_HF_LLAMA3)
    assert e.value.status == 1


def test_gemma3_options_serializer(mc):
    text = json.dumps(dict(head_dim=256, hidden_size=1152, num_attention_heads=4, num_key_value_heads=1,
                           num_hidden_layers=26, sliding_window=512, _sliding_window_pattern=6,
                           sliding_window_pattern=3, query_pre_attn_scalar=256, rms_norm_eps=1e-6,
                           rope_theta=1000000.0, rope_local_base_freq=10000.0, intermediate_size=6912,
                           vocab_size=262144))
    c = mc.config_from_json(text, mc.CKPT_HF_GEMMA3)
    o = ck.options_hf_gemma(text)
    assert (c.head_dim, c.dim, c.n_heads, c.n_kv_heads, c.n_layers, c.sliding_stride, c.max_seq_len) == \
        (o["head_dim"], o["hidden_dim"], o["n_heads"], o["n_kv_heads"], o["n_layers"], o["sliding_stride"], 1024)
    assert c.sliding_stride == 6 and c.family == mc.FAMILY_GEMMA3
    assert abs(c.attn_scale - 1.0 / 16.0) < 1e-7 and c.rope_sliding_theta == 10000.0   # nn/gemma.h:99


def test_permute_attention_heads_restatement_is_an_involution_pair():
    # unpinned by the reference; the loop restatement and the reshape form must agree
    rng = np.random.default_rng(4)
    w = rng.normal(size=(4 * 8, 5)).astype(np.float32)
    p = ck.permute_attention_heads(w, 4)
    assert np.array_equal(p, w.reshape(4, 4, 2, 5).swapaxes(1, 2).reshape(32, 5))
    assert np.array_equal(ck.unpermute_attention_heads(p, 4), w)


def test_config_from_document_reads_widths_from_tensors(mc, tmp_path):
    cfg = mg.tiny_cfg(F32, n_layers=2)
    w = cg.tie_head(mg.make_model(cfg, seed=5))
    p = cg.write_checkpoint(str(tmp_path / "model.safetensors"), w, cfg, cg.HF_LLAMA)
    d = mc.Document(p)
    d.adapt(mc.CKPT_HF_LLAMA3)
    c = mc.config_from_json(cg.options_json(cfg, cg.HF_LLAMA), mc.CKPT_HF_LLAMA3)
    c.dim = c.ffn_dim = c.vocab = 0
    mc.config_from_document(d, c)
    assert (c.dim, c.ffn_dim, c.vocab, c.n_layers, c.layer_end) == (cfg["dim"], cfg["ffn_dim"], cfg["vocab"], 2, 2)
