"""The N > 1 layer-pipeline protocol (tests/pipeline_schedule.py) over gloo on CPU, world_size 2 and
3, with the CPU oracle as the stage compute: the pipelined greedy tokens must equal the
single-process oracle's (bit-exact integer path), and the layer split must cover every layer once."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layer_range_covers_all_layers():
    from pipeline_schedule import layer_range

    for n_layers in (1, 2, 22, 32, 80):
        for world in (1, 2, 3, 4, 8):
            if world > n_layers:
                continue
            got = [layer_range(r, world, n_layers) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n_layers
            for a, b in zip(got, got[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in got]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_tokens, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import modelgen as mg
    from pipeline_schedule import layer_range, pipelined_decode
    from oracle import mc_oracle as mo

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mo.set_num_threads(2)
    cfg = mg.tiny_cfg(mg.BF16, n_layers=3, max_seq_len=16)
    weights = mg.make_model(cfg, seed=31, quant="i4", group=32)
    om = mo.Model(cfg, weights)  # every rank holds the weights; it only runs its own layers
    lb, le = layer_range(rank, world, cfg["n_layers"])
    h_in = torch.zeros(cfg["dim"], dtype=torch.int16)
    h_out = torch.zeros(cfg["dim"], dtype=torch.int16)
    tok_buf = torch.zeros(1, dtype=torch.int32)

    def step(token, pos):
        hin = h_in.numpy().view(np.uint16) if rank > 0 else None
        tok, hout = om.step_range(token, pos, lb, le, hidden_in=hin)
        if hout is not None:
            h_out.copy_(torch.from_numpy(hout.view(np.int16)))
        return tok if le == cfg["n_layers"] else None

    toks = pipelined_decode(dist, rank, world, h_in, h_out, tok_buf, step, 5, 0, n_tokens)
    if rank == 0:
        q.put(toks)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_tokens_equal_single_process(world):
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import modelgen as mg
    from oracle import mc_oracle as mo

    n_tokens = 20  # crosses max_seq_len 16: the sink-cache roll happens on every stage
    cfg = mg.tiny_cfg(mg.BF16, n_layers=3, max_seq_len=16)
    weights = mg.make_model(cfg, seed=31, quant="i4", group=32)
    om = mo.Model(cfg, weights)
    tok, ref = 5, []
    for pos in range(n_tokens):
        tok, _ = om.step(tok, pos, want_logits=False)
        ref.append(tok)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_tokens, q)) for r in range(world)]
    for p in procs:
        p.start()
    toks = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert toks == ref


def test_the_native_pipeline_ships_the_hidden_row_as_bytes_and_bench_needs_no_torch():
    # RCCL has no 16-bit unsigned element type (its process group rejects UInt16 tensors, tools/dtype_probe.py on the
    # MI355X box): the bf16 hidden row travels as bytes through ncclSend / ncclRecv (csrc/decoder.cc, mc_pipeline_generate);
    # bench.py drives that C ABI and must not import torch at any N
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = open(os.path.join(root, "metalchat_amd", "csrc", "decoder.cc")).read()
    assert re.search(r"api\.Send\(d->hidden, row, ncclUint8", host) and re.search(r"api\.Recv\(d->hidden_in, row, ncclUint8", host)
    bench = open(os.path.join(root, "bench.py")).read()
    assert not re.search(r"^\s*(import|from)\s+torch", bench, re.M)


def test_bench_gpus_n_starts_its_own_workers_and_does_not_hang_when_they_die():
    # `python bench.py --gpus 2` without a launcher spawns two fresh workers; in this container there is no GPU, so both
    # fail in mc_device_create ("no HIP device") -- the parent must come back with their exit code instead of waiting
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "HIP" in r.stderr or "hip" in r.stderr
