"""Child process of tests/test_torch_interop_gpu.py: torch is imported FIRST (as in bench.py), so
its bundled HIP runtime and the library's resolve to one."""
import sys

import numpy as np

sys.path.insert(0, sys.argv[1])          # tests/
sys.path.insert(0, sys.argv[2])          # repo root
import torch  # noqa: E402

import modelgen as mg  # noqa: E402

BF16, F32 = 0, 1


class _Raw:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr=typestr, data=(ptr, False), version=3)


def run(dt):
    import metalchat_amd as mc

    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    ts = torch.cuda.Stream(device=0)
    torch.cuda.set_stream(ts)
    acc = mc.HardwareAccelerator(ordinal=0, stream=ts.cuda_stream)
    cfg = mg.tiny_cfg(dt, n_layers=4, max_seq_len=32)
    w = mg.make_model(cfg, seed=91, quant="i4", group=32)
    kw = dict(weight_format=2, group_size=32)
    whole = mc.Decoder(acc, **mg.decoder_kwargs(cfg, **kw))
    whole.load_model(w)
    s0 = mc.Decoder(acc, **mg.decoder_kwargs(cfg, layer_begin=0, layer_end=2, **kw))
    s1 = mc.Decoder(acc, **mg.decoder_kwargs(cfg, layer_begin=2, layer_end=4, **kw))
    s0.load_model(w)
    s1.load_model(w)
    s0.set_taps(True)
    tt = "<u2" if dt == BF16 else "<f4"
    h_out = torch.as_tensor(_Raw(s0.hidden_out_ptr(), cfg["dim"], tt), device="cuda:0")
    h_in = torch.as_tensor(_Raw(s1.hidden_in_ptr(), cfg["dim"], tt), device="cuda:0")
    tok = 3
    for pos in range(6):
        s0.step(tok, pos, sync=False)
        acc.wait()
        # the "hop": a torch copy on the adopted stream between the two stages' buffers
        h_in.copy_(h_out)
        torch.cuda.current_stream().synchronize()
        got = s1.step(-1, pos, hidden_in=s1.hidden_in_ptr())
        ref = whole.step(tok, pos)
        assert got == ref, f"pos {pos}"
        assert np.array_equal(s1.logits(), whole.logits())
        # torch sees exactly the bytes the decoder wrote
        host = h_out.cpu().numpy().view(np.uint16 if dt == BF16 else np.float32)
        assert np.array_equal(host, s0.hidden(1))
        tok = ref
    for d in (whole, s0, s1):
        d.release()
    torch.cuda.set_stream(torch.cuda.default_stream(0))


if __name__ == "__main__":
    for dt in (BF16, F32):
        run(dt)
    print("interop ok")
