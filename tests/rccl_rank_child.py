"""One rank of tests/test_pipeline_gpu.py::test_rccl_two_ranks_equal_single_stage: a process of its own WITHOUT torch
(tests/rccl_child.py says why), started before anything in it has touched the GPU, one GPU per rank -- exactly how
bench.py --gpus N runs its workers.  argv: tests dir, repo root, rank, world, uid file.

Every rank builds the SAME seeded model, owns mc_pipeline_layer_range(rank, world) of its layers and decodes 44 chained
tokens (past max_seq_len = 32: the sink ring turns on every stage) through mc_pipeline_generate over RCCL; rank 0 also
decodes them with a single-stage decoder on its own GPU and requires the very same ids, then runs a prompt pass through
the pipeline (mc_pipeline_prefill) against the single stage's."""
import os
import sys
import time

sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[2])
rank, world, uid_file = int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
assert "torch" not in sys.modules
import numpy as np  # noqa: E402

import modelgen as mg  # noqa: E402

import metalchat_amd as mc  # noqa: E402

if mc.device_count() < world:
    print(f"rccl rank {rank}: only {mc.device_count()} device(s)")
    sys.exit(77)
acc = mc.HardwareAccelerator(ordinal=rank)
cfg = mg.tiny_cfg(0, n_layers=5, max_seq_len=32)
weights = mg.make_model(cfg, seed=4, quant="i4", group=32)
lb, le = mc.pipeline_layer_range(rank, world, cfg["n_layers"])
stage = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=32, layer_begin=lb, layer_end=le))
stage.load_model(weights)
if rank == 0:
    uid = mc.pipeline_unique_id()
    with open(uid_file + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(uid_file + ".tmp", uid_file)
else:
    t0 = time.time()
    while not os.path.exists(uid_file):
        if time.time() - t0 > 120:
            sys.exit(3)
        time.sleep(0.05)
    uid = open(uid_file, "rb").read()
pipe = mc.Pipeline.rccl(stage, rank, world, uid)
n = 44
got = list(pipe.generate(3, 0, n))
prompt = np.random.default_rng(5).integers(0, cfg["vocab"], 9)
nxt = pipe.prefill(prompt, 0)
follow = list(pipe.generate(nxt if rank in (0, world - 1) else 0, len(prompt), 6))
assert pipe.allreduce_max(float(rank)) == float(world - 1)
pipe.release()
stage.release()
if rank == 0:
    single = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=32))
    single.load_model(weights)
    want = list(single.generate(3, 0, n))
    assert got == want, (got, want)
    wnxt = single.prefill(prompt, 0)
    assert nxt == wnxt, (nxt, wnxt)
    assert follow == list(single.generate(wnxt, len(prompt), 6))
    single.release()
    print(f"rccl {world} ranks ok: {n} tokens, prompt pass and continuation equal the single stage")
assert "torch" not in sys.modules
