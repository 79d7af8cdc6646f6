"""serializer.load(document) on the GPU: a decoder filled from a checkpoint FILE (safetensors
written by the independent `safetensors` package, config from the options JSON + tensor shapes)
must produce bit-identical logits and tokens to a decoder filled from the same arrays through
mc_decoder_load_linear, and match the CPU oracle (which is handed the arrays the numpy
restatement of the adaptor yields).  All four serializers of the reference:
reference::llama3 (wq / wk head permutation), huggingface::llama3, huggingface::llama3_qlora
(int8-held int4 g32 + adaptors + quantised table and head), huggingface::gemma3."""
import numpy as np
import pytest

import ckptgen as cg
import modelgen as mg
import parity
from oracle import ckpt_oracle as ck
from oracle import mc_oracle as mo

pytestmark = pytest.mark.gpu
BF16, F32 = 0, 1


def decoder_from_file(acc, mc, path, cfg, flavour, sharded=False, **over):
    doc = mc.Document(path, sharded=sharded)
    doc.adapt(flavour)
    c = mc.config_from_json(cg.options_json(cfg, flavour), flavour)
    c.dtype = cfg["dtype"]
    c.max_seq_len = cfg["max_seq_len"]   # the serializers pin 1024; the tests use small caches
    for k, v in over.items():
        setattr(c, k, v)
    mc.config_from_document(doc, c)
    dec = mc.Decoder.from_config(acc, c)
    dec.load_document(doc, flavour)
    doc.release()                         # weights were repacked into HBM: the file can go
    return dec, c


def same_tokens_and_logits(a, b, steps=6):
    tok = 3
    for pos in range(steps):
        ta, tb = a.step(tok, pos), b.step(tok, pos)
        assert ta == tb, f"pos {pos}"
        assert np.array_equal(a.logits(), b.logits()), f"pos {pos} logits differ"
        tok = ta


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("flavour", [cg.META, cg.HF_LLAMA])
def test_llama_checkpoints_load_like_arrays(acc, tmp_path, dt, flavour):
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(dt, max_seq_len=16)
    w = cg.tie_head(mg.make_model(cfg, seed=41))
    p = cg.write_checkpoint(str(tmp_path / "model.safetensors"), w, cfg, flavour)
    dec, c = decoder_from_file(acc, mc, p, cfg, flavour)
    assert (c.dim, c.ffn_dim, c.vocab, c.n_layers) == (cfg["dim"], cfg["ffn_dim"], cfg["vocab"], cfg["n_layers"])
    ref = mc.Decoder(acc, **mg.decoder_kwargs(cfg))
    ref.load_model(w)
    same_tokens_and_logits(dec, ref)
    # and the oracle, fed what the numpy restatement of the adaptor produces from the file tensors
    t = cg.tensors_for(w, cfg, flavour)
    if flavour == cg.META:
        wq = ck.permute_attention_heads(t["layers.0.attention.wq.weight"], cfg["n_heads"])
        assert np.array_equal(wq, w["layers"][0]["wq"]["weight"])
    om = mo.Model(cfg, w)
    otok, ologits = om.step(3, 0)
    d2, _ = decoder_from_file(acc, mc, p, cfg, flavour)
    assert d2.step(3, 0) == otok
    parity.check(dt, d2.logits(), ologits, rel=1e-4 if dt == F32 else 2e-3, max_ulp=2,
                 max_frac=1.0 if dt == F32 else 0.5, what="logits vs oracle")
    om.close()
    for d in (dec, ref, d2):
        d.release()


def test_qlora_checkpoint(acc, tmp_path):
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(BF16, max_seq_len=16)
    w = mg.make_model(cfg, seed=42, quant="i4", group=32, lora_rank=16, emb_quant=True, head_quant="i8row")
    p = cg.write_checkpoint(str(tmp_path / "model.safetensors"), w, cfg, cg.META_QLORA)
    dec, c = decoder_from_file(acc, mc, p, cfg, cg.META_QLORA, weight_format=mc.WFMT_I4)
    assert c.group_size == 32
    ref = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=2, group_size=32))
    ref.load_model(w)
    same_tokens_and_logits(dec, ref)
    dec.release()
    ref.release()


def test_gemma3_checkpoint_sharded(acc, tmp_path):
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(F32, family=1, n_layers=3, rope_sliding_theta=10000.0, sliding_stride=2,
                      attn_scale=float(1.0 / np.sqrt(48.0)), max_seq_len=16)
    w = cg.tie_head(mg.make_model(cfg, seed=43))
    idx = cg.write_checkpoint(str(tmp_path / "model.safetensors"), w, cfg, cg.HF_GEMMA, shards=3)
    dec, c = decoder_from_file(acc, mc, idx, cfg, cg.HF_GEMMA, sharded=True)
    assert c.sliding_stride == 2 and abs(c.attn_scale - cfg["attn_scale"]) < 1e-6
    ref = mc.Decoder(acc, **mg.decoder_kwargs(cfg))
    ref.load_model(w)
    same_tokens_and_logits(dec, ref)
    dec.release()
    ref.release()


def test_pipeline_stage_loads_only_its_layers_and_unknown_names_fail(acc, tmp_path):
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(F32, n_layers=4, max_seq_len=16)
    w = cg.tie_head(mg.make_model(cfg, seed=44))
    p = cg.write_checkpoint(str(tmp_path / "model.safetensors"), w, cfg, cg.HF_LLAMA)
    # stage 1 of 2 owns layers [2, 4) + norm + head; chained after stage 0 it reproduces one decoder
    s0, _ = decoder_from_file(acc, mc, p, cfg, cg.HF_LLAMA, layer_begin=0, layer_end=2)
    s1, _ = decoder_from_file(acc, mc, p, cfg, cg.HF_LLAMA, layer_begin=2, layer_end=4)
    whole, _ = decoder_from_file(acc, mc, p, cfg, cg.HF_LLAMA)
    tok = 3
    for pos in range(4):
        s0.step(tok, pos, sync=False)
        acc.wait()
        got = s1.step(tok, pos, hidden_in=s0.hidden_out_ptr())
        assert got == whole.step(tok, pos)
        assert np.array_equal(s1.logits(), whole.logits())
        tok = got
    for d in (s0, s1, whole):
        d.release()
    # a tensor the model does not register: layer.parameter(name) throws in the reference
    doc = mc.Document(p)
    doc.adapt(mc.CKPT_HF_LLAMA3)
    doc.insert("layers.1.attention.wq.bias", np.zeros(4, np.float32))
    c = mc.config_from_json(cg.options_json(cfg, cg.HF_LLAMA), mc.CKPT_HF_LLAMA3)
    c.dtype, c.max_seq_len = F32, 16
    mc.config_from_document(doc, c)
    dec = mc.Decoder.from_config(acc, c)
    with pytest.raises(mc.McError, match="is not registered"):
        dec.load_document(doc, mc.CKPT_HF_LLAMA3)
    dec.release()
