"""Child process of tests/test_trace_gpu.py: one short generation; prints the tokens and whether launches are wrapped in
named ranges (MC_TRACE_RANGES is read once per process, so each setting gets a process of its own)."""
import sys

sys.path.insert(0, sys.argv[1])  # tests/
sys.path.insert(0, sys.argv[2])  # repo root
import modelgen as mg  # noqa: E402

import metalchat_amd as mc  # noqa: E402
from metalchat_amd import runtime  # noqa: E402

acc = mc.HardwareAccelerator(ordinal=0)
cfg = mg.tiny_cfg(0, n_layers=2)
weights = mg.make_model(cfg, seed=1, quant="i4", group=32)
d = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=32))
d.load_model(weights)
toks = [int(t) for t in d.generate(3, 0, 8)]
on = int(runtime.capi().mc_trace_ranges_enabled())
d.release()
print("tokens", toks, "ranges", on)
