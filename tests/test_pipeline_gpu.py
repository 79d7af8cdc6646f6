"""The layer pipeline behind the C ABI (mc_pipeline_*) on the GPU.

One box has one GPU, so the N stages run in ONE process here (mc_pipeline_create_local: device-to-device hops
behind events -- the same launches per stage as the RCCL transport, whose hop is ncclSend / ncclRecv on the same
stream).  The split must not change a single bit: tokens, logits and every stage's caches equal the single-stage
decoder's for N = 2, 4, 8, llama3 and gemma3, greedy and the device sampler, past max_seq_len (sink ring), and
against the oracle's own stage split (mco_model_step_range)."""
import numpy as np
import pytest

import modelgen as mg
import parity
from oracle import mc_oracle as mo

pytestmark = pytest.mark.gpu
BF16, F32 = 0, 1


def build(acc, cfg, weights, world, fmt, group, **over):
    import metalchat_amd as mc

    stages = []
    for r in range(world):
        lb, le = mc.pipeline_layer_range(r, world, cfg["n_layers"])
        d = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=fmt, group_size=group, layer_begin=lb, layer_end=le, **over))
        d.load_model(weights)
        stages.append(d)
    return stages


@pytest.mark.parametrize("family,dtype", [(0, BF16), (0, F32), (1, BF16)])
def test_local_pipeline_equals_the_single_stage_bit_for_bit(acc, family, dtype):
    import metalchat_amd as mc

    over = dict(family=family, n_layers=8, max_seq_len=24)
    if family == 1:
        over.update(rope_sliding_theta=10000.0, sliding_stride=2)
    cfg = mg.tiny_cfg(dtype, **over)
    weights = mg.make_model(cfg, seed=11, quant="i4", group=32)
    single = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=32))
    single.load_model(weights)
    n = 40  # 16 tokens past max_seq_len: every stage's ring turns
    want = list(single.generate(5, 0, n))
    want_logits = single.logits().copy()
    want_kv = {layer: tuple(a.copy() for a in single.export_kv(layer)) for layer in range(cfg["n_layers"])}
    want_more = list(single.generate(want[-1], n, 6))  # a second call continues from the caches (start_pos > 0)
    single.release()
    # (eager launches, and one hipGraph per stage -- the launches of a token between two hops, captured at the second token)
    for world, graph in ((2, 0), (4, 0), (8, 0), (2, 1), (4, 1), (8, 1)):
        stages = build(acc, cfg, weights, world, mc.WFMT_I4, 32, use_graph=graph)
        pipe = mc.Pipeline.local(stages)
        got = list(pipe.generate(5, 0, n))
        assert got == want, f"world {world} graph {graph}: tokens differ"
        parity.exact(stages[-1].logits(), want_logits, f"world {world}: logits of the last token")
        for r, d in enumerate(stages):
            for layer in range(d.cfg["layer_begin"], d.cfg["layer_end"]):
                gk, gv = d.export_kv(layer)
                parity.exact(gk, want_kv[layer][0], f"world {world} stage {r} K[{layer}]")
                parity.exact(gv, want_kv[layer][1], f"world {world} stage {r} V[{layer}]")
        assert list(pipe.generate(got[-1], n, 6)) == want_more, f"world {world}: continuation differs"
        pipe.release()
        for d in stages:
            d.release()


def test_local_pipeline_against_the_oracle_stage_split(acc):
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(F32, n_layers=4, max_seq_len=32)
    weights = mg.make_model(cfg, seed=2, quant="i8", group=32)
    om = mo.Model(cfg, weights)
    otoks, tok = [], 3
    for pos in range(12):
        tok, _ = om.step(tok, pos)
        otoks.append(tok)
    stages = build(acc, cfg, weights, 2, mc.WFMT_I8, 32)
    pipe = mc.Pipeline.local(stages)
    assert list(pipe.generate(3, 0, 12)) == otoks
    pipe.release()
    for d in stages:
        d.release()
    om.close()


def test_local_pipeline_with_the_device_sampler(acc):
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(BF16, n_layers=4, max_seq_len=32)
    weights = mg.make_model(cfg, seed=4, quant="i4", group=32)
    single = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=32))
    single.load_model(weights)
    single.set_sampler(mc.SAMPLER_DEFAULT, 50, 0.6, 0.9)
    single.set_seeds([(1, 2), (3, 4)])
    want = list(single.generate(7, 0, 20))
    stages = build(acc, cfg, weights, 4, mc.WFMT_I4, 32)
    stages[-1].set_sampler(mc.SAMPLER_DEFAULT, 50, 0.6, 0.9)
    stages[-1].set_seeds([(1, 2), (3, 4)])
    pipe = mc.Pipeline.local(stages)
    assert list(pipe.generate(7, 0, 20)) == want
    pipe.release()
    for d in stages + [single]:
        d.release()


def test_pipeline_rejects_a_wrong_layer_split(acc):
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(BF16, n_layers=4)
    weights = mg.make_model(cfg, seed=1, quant="i4", group=32)
    a = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=32, layer_begin=0, layer_end=3))
    b = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=32, layer_begin=3, layer_end=4))
    a.load_model(weights)
    b.load_model(weights)
    with pytest.raises(mc.McError, match="must own layers"):
        mc.Pipeline.local([a, b])
    a.release()
    b.release()


def test_rccl_transport_comes_up_with_one_rank():
    # the RCCL path needs one GPU per rank; what a one-GPU box can check is that librccl loads, a unique id is made, a
    # communicator of one rank initialises on the decoder's device and generate() degenerates to the single stage.
    # In a child process without torch (tests/rccl_child.py says why), which is how bench.py runs its ranks.
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "rccl_child.py"), here, os.path.dirname(here)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl child ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("world", [2, 4])
def test_rccl_two_ranks_equal_single_stage(world):
    # The RCCL hop for real: `world` torch-free child processes, one GPU each, started BEFORE any of them touches the
    # GPU (tests/rccl_rank_child.py); tokens past max_seq_len, the prompt pass and its continuation must equal the
    # single-stage decoder's.  Skipped where fewer devices are visible (the one-GPU test box).
    import os
    import subprocess
    import sys
    import tempfile

    import metalchat_amd as mc

    if mc.device_count() < world:
        pytest.skip(f"{world} GPUs needed, {mc.device_count()} visible")
    here = os.path.dirname(os.path.abspath(__file__))
    with tempfile.TemporaryDirectory(prefix="mc_rccl_") as tmp:
        uid_file = os.path.join(tmp, "uid")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs = [subprocess.Popen([sys.executable, os.path.join(here, "rccl_rank_child.py"), here, os.path.dirname(here), str(r), str(world),
                                   uid_file], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
        outs = []
        try:
            for p in procs:
                outs.append(p.communicate(timeout=600))
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: rc {p.returncode}\n{o[-2000:]}\n{e[-4000:]}"
    assert f"rccl {world} ranks ok" in outs[0][0]


def test_local_pipeline_of_gemma_7b_shaped_blocks_equals_the_single_stage_bit_for_bit(acc):
    """The gemma3 block in one launch (round 5, mc_attn_qkv_wo_qkn_*) at the seams of a layer pipeline: the first block of a later stage has no
    post-norm in front of it (`_p1_` on the row the hop delivered), the last block of a stage leaves its ffn post-norm to mc_rmsnorm_row so that the
    row can travel -- two stages of two Gemma-7B-wide blocks each against the single stage: tokens, logits and every cache bit for bit, across the
    end of the cache."""
    import metalchat_amd as mc
    from test_context_gpu import SEED, random_cache

    cfg = dict(dtype=BF16, n_layers=4, vocab=2048, max_seq_len=2048, norm_eps=1e-5, dim=3072, n_heads=16, n_kv_heads=16, head_dim=256,
               ffn_dim=4096, family=1, rope_theta=10000.0, rope_sliding_theta=10000.0, sliding_stride=2, attn_scale=256 ** -0.5)
    kw = mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=128)
    caches = [random_cache(cfg, 2040, 1300 + layer) for layer in range(cfg["n_layers"])]
    single = mc.Decoder(acc, **kw)
    single.init_synthetic(SEED)
    for layer, (k, v) in enumerate(caches):
        single.import_kv(layer, k, v)
    single.launch_log(True)
    want = list(single.generate(5, 2040, 20))
    assert {"mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p1_t2", "mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p2_t2"} <= set(single.launched())
    want_logits = single.logits().copy()
    want_kv = [tuple(a.copy() for a in single.export_kv(layer)) for layer in range(cfg["n_layers"])]
    single.release()
    stages = []
    for r in range(2):
        lb, le = mc.pipeline_layer_range(r, 2, cfg["n_layers"])
        d = mc.Decoder(acc, **dict(kw, layer_begin=lb, layer_end=le))
        d.init_synthetic(SEED)
        for layer in range(lb, le):
            d.import_kv(layer, *caches[layer])
        stages.append(d)
    pipe = mc.Pipeline.local(stages)
    got = list(pipe.generate(5, 2040, 20))
    assert got == want
    parity.exact(stages[-1].logits(), want_logits, "logits of the last token")
    for r, d in enumerate(stages):
        for layer in range(d.cfg["layer_begin"], d.cfg["layer_end"]):
            gk, gv = d.export_kv(layer)
            parity.exact(gk, want_kv[layer][0], f"stage {r} K[{layer}]")
            parity.exact(gv, want_kv[layer][1], f"stage {r} V[{layer}]")
    pipe.release()
    for d in stages:
        d.release()
