"""Layer pipeline over several devices, one process per device.

The reference runs on one Metal device (src/metal.cc:51-55); the only state that crosses a layer
boundary is the hidden row [1,1,dim] (include/metalchat/nn/llama.h:123-126), so contiguous layer
ranges shard naturally with ONE point-to-point hop per stage boundary and one 4-byte hop that
brings the greedy token back to the first stage.  No all-reduce, no all-gather.

This module is TEST INFRASTRUCTURE -- the schedule only (round 4: moved out of the product package): tests/test_pipeline_cpu.py drives it over gloo
(world 2 and 3) with the CPU oracle as the stage, so the hop protocol -- who sends what to whom, in which
order, past max_seq_len -- is covered without a GPU.  The product path does not use it: bench.py and the
C++ shim go through mc_pipeline_* (metalchat_amd/csrc/decoder.cc: ncclSend / ncclRecv on the decoder's
stream, or device-to-device copies behind events), which implement the same schedule and are checked on the
GPU by tests/test_pipeline_gpu.py (local transport == single stage bit for bit; RCCL with 2 and 4 ranks
where that many devices are visible).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple


def layer_range(rank: int, world: int, n_layers: int) -> Tuple[int, int]:
    """Contiguous split; the first n_layers % world stages get one extra layer (== mc_pipeline_layer_range)."""
    base, extra = divmod(n_layers, world)
    lb = rank * base + min(rank, extra)
    return lb, lb + base + (1 if rank < extra else 0)


def pipelined_decode(dist, rank: int, world: int, h_in, h_out, tok_buf,
                     step: Callable[[int, int], Optional[int]], first_token: int, start_pos: int,
                     n: int) -> List[int]:
    """Greedy decode of n tokens through the pipeline.

    h_in / h_out: torch tensors (on the stage's device) the stage reads its inbound hidden row from
    and writes its outbound hidden row to; tok_buf: a 1-element int32 tensor on the same device.
    step(token, pos) runs the stage's layers (token is meaningful on rank 0 only) and returns the
    greedy token on the last stage, None elsewhere.  Returns the generated tokens on rank 0 (and on
    the last rank), [] on the others.
    """
    out: List[int] = []
    tok = first_token
    last = world - 1
    for i in range(n):
        pos = start_pos + i
        if rank > 0:
            dist.recv(h_in, src=rank - 1)
        nt = step(tok if rank == 0 else -1, pos)
        if rank < last:
            dist.send(h_out, dst=rank + 1)
        if world > 1:
            # token hop back to the first stage
            if rank == last:
                tok_buf.fill_(int(nt))
                dist.send(tok_buf, dst=0)
                out.append(int(nt))
            elif rank == 0:
                dist.recv(tok_buf, src=last)
                tok = int(tok_buf.item())
                out.append(tok)
        else:
            tok = int(nt)
            out.append(tok)
    return out
