"""The greedy pick inside the output head's launch (gemv.h EPI_STORE_PICK) against the argmax launch behind it
(mc_argmax_T, MC_HEAD_PICK=0): same tokens, same logits -- and, with every logit equal, the FIRST index, whichever
workgroup finishes last (mc_argmax_T's rule: first index of the maximum)."""
import os

import numpy as np
import pytest

import modelgen as mg

import metalchat_amd as mc


def _decoder(acc, cfg, weights, pick, wfmt, group):
    old = os.environ.get("MC_HEAD_PICK")
    os.environ["MC_HEAD_PICK"] = str(int(pick))
    try:
        d = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=wfmt, group_size=group))
    finally:
        if old is None:
            os.environ.pop("MC_HEAD_PICK", None)
        else:
            os.environ["MC_HEAD_PICK"] = old
    d.load_model(weights)
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("quant,wfmt,dim,vocab", [("i4", mc.WFMT_I4, 2048, 6000), ("i4", mc.WFMT_I4, 4096, 33000),
                                                  (None, mc.WFMT_T, 2048, 4100), ("i8", mc.WFMT_I8, 4096, 2052)])
def test_pick_in_the_head_launch_equals_the_argmax_launch(quant, wfmt, dim, vocab):
    acc = mc.HardwareAccelerator(ordinal=0)
    cfg = mg.tiny_cfg(0, dim=dim, n_heads=dim // 128, n_kv_heads=dim // 512, head_dim=128, ffn_dim=2048, n_layers=1,
                      vocab=vocab, max_seq_len=64)
    weights = mg.make_model(cfg, seed=11, quant=quant, group=128)
    b = _decoder(acc, cfg, weights, 0, wfmt, 128 if quant else 0)
    tb = list(b.generate(5, 0, 40))
    for mode in (1, 2):  # 1: atomic key + ticket inside the head's launch; 2: a key per workgroup + mc_argmax_keys
        a = _decoder(acc, cfg, weights, mode, wfmt, 128 if quant else 0)
        assert list(a.generate(5, 0, 40)) == tb, mode
        assert np.array_equal(a.logits(), b.logits())
        # single steps (host-visible token) and a second generate on the same decoder (key and ticket were left clean)
        assert a.step(7, 0) == b.step(7, 0)
        assert list(a.generate(9, 3, 17)) == list(b.generate(9, 3, 17))
        a.release()
        b.generate(5, 0, 40)
    b.release()


@pytest.mark.gpu
def test_equal_logits_pick_the_first_index():
    acc = mc.HardwareAccelerator(ordinal=0)
    cfg = mg.tiny_cfg(0, dim=2048, n_heads=16, n_kv_heads=4, head_dim=128, ffn_dim=2048, n_layers=1, vocab=8192, max_seq_len=32)
    weights = mg.make_model(cfg, seed=3, quant="i4", group=128)
    out = weights["output"]
    out["weight"][:] = out["weight"][0]   # every row of the head the same: every logit the same
    out["scales"][:] = out["scales"][0]
    for pick in (2, 1, 0):
        d = _decoder(acc, cfg, weights, pick, mc.WFMT_I4, 128)
        toks = list(d.generate(5, 0, 6))
        lg = np.asarray(d.logits())
        assert np.all(lg == lg[0])
        assert toks[1:] == [0] * 5, (pick, toks)
        d.release()
    # ... and with the maximum in the LAST row only, that row wins wherever its workgroup finishes
    out["weight"][-1] = np.clip(-out["weight"][0].astype(np.int32), -8, 7).astype(np.int8)
    db = _decoder(acc, cfg, weights, 0, mc.WFMT_I4, 128)
    want = list(db.generate(5, 0, 6))
    for mode in (1, 2):
        da = _decoder(acc, cfg, weights, mode, mc.WFMT_I4, 128)
        assert list(da.generate(5, 0, 6)) == want
        da.release()
    db.release()
