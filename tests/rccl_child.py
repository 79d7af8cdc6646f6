"""Child process of tests/test_pipeline_gpu.py::test_rccl_transport_comes_up_with_one_rank.

A process of its own WITHOUT torch, as the workers of bench.py are: the torch wheel carries private copies of
libamdhip64 / libhsa-runtime64 / librccl, and once they are in a process (pytest imports tests/ckptgen.py, hence torch,
while collecting) `dlopen("librccl.so.1")` resolves to torch's copy, whose own `dlopen("libhsa-runtime64.so")` then opens
the system file as a SECOND, uninitialised HSA runtime (hsa_system_get_info -> 0x100B, "no ROCm-capable device")."""
import sys

sys.path.insert(0, sys.argv[1])  # tests/
sys.path.insert(0, sys.argv[2])  # repo root
assert "torch" not in sys.modules
import modelgen as mg  # noqa: E402

import metalchat_amd as mc  # noqa: E402

acc = mc.HardwareAccelerator(ordinal=0)
cfg = mg.tiny_cfg(0, n_layers=2)
weights = mg.make_model(cfg, seed=1, quant="i4", group=32)
d = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=32))
d.load_model(weights)
want = list(d.generate(3, 0, 10))
uid = mc.pipeline_unique_id()
assert len(uid) == 128 and any(uid)
pipe = mc.Pipeline.rccl(d, 0, 1, uid)
assert list(pipe.generate(3, 0, 10)) == want
assert pipe.allreduce_max(1.5) == 1.5
pipe.release()
d.release()
assert "torch" not in sys.modules
print("rccl child ok")
