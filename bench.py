#!/usr/bin/env python3
"""Headline benchmark: single-stream (batch = 1) greedy decode tokens/s of int4 Llama-3-8B on
MI355X, with the achieved HBM rate of the dominant kernel (the fused int4 GEMV) against the
8 TB/s roofline and the CPU restatement of the reference timed beside it.

  python bench.py --gpus N --steps K --warmup W          (N = 1 default)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one generated token: embedding -> 32 x (rmsnorm, QKV GEMV, RoPE, sink-cache write,
QK^T, softmax, PV, Wo, residual, rmsnorm, w1|w3 GEMV, SiLU*mul, w2 GEMV, residual) -> final norm
-> output-head GEMV -> greedy argmax, exactly what the reference's transform(token, start_pos)
does per token (include/metalchat/transformer.h:357-364).  Weights are synthetic (counter-based
hash, generated on the device), the context is filled by really decoding up to
seq_len - steps before the timed region, and the timed tokens end at position seq_len so the KV
traffic is at its configured maximum.

N > 1: the layers are pipelined over N GPUs (contiguous layer ranges, one RCCL send/recv of the
hidden row per stage boundary and one 4-byte token hop back), one process per GPU.  At batch 1
only one stage is busy at a time, so this buys capacity, not speed: scaling is "strong".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

MODELS = {
    # SURVEY.md section 8 shape table
    "llama3-8b": dict(dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=32,
                      vocab=128256, rope_theta=500000.0, norm_eps=1e-5),
    "tinyllama-1.1b": dict(dim=2048, n_heads=32, n_kv_heads=4, head_dim=64, ffn_dim=5632,
                           n_layers=22, vocab=32000, rope_theta=10000.0, norm_eps=1e-5),
    "llama3-70b": dict(dim=8192, n_heads=64, n_kv_heads=8, head_dim=128, ffn_dim=28672,
                       n_layers=80, vocab=128256, rope_theta=500000.0, norm_eps=1e-5),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=256)
    p.add_argument("--warmup", type=int, default=32)
    p.add_argument("--model", default="llama3-8b", choices=sorted(MODELS))
    p.add_argument("--wbits", type=int, default=4, choices=[4, 8, 16])
    p.add_argument("--group", type=int, default=128)
    p.add_argument("--seq-len", type=int, default=2048)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--qmode", default="exact", choices=["exact", "fast"])
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--seed", type=int, default=0x5EED)
    p.add_argument("--share-device", action="store_true",
                   help="validation aid: every rank uses GPU 0 and the hops go over gloo (RCCL refuses two "
                        "ranks on one device), so the N > 1 path can be exercised end to end on a 1-GPU box")
    p.add_argument("--force-pipeline", action="store_true",
                   help="run the N > 1 code path (torch stream, RCCL group, stage loop) with one rank: a "
                        "smoke check of the multi-GPU plumbing on a 1-GPU box")
    return p.parse_args()


def algorithmic_bytes(m, wbits, group, seq_len, tbytes):
    """weights + scales of every linear + KV read + KV write + embedding row."""
    L, dim, H, KV, hd, ffn, V = (m["n_layers"], m["dim"], m["n_heads"], m["n_kv_heads"],
                                 m["head_dim"], m["ffn_dim"], m["vocab"])
    p_mm = L * (dim * H * hd + 2 * dim * KV * hd + H * hd * dim + 3 * dim * ffn) + V * dim
    sb = 0 if wbits == 16 else (2 if tbytes == 2 else 4)
    w = p_mm * (wbits / 8.0 + (sb / group if sb else 0.0))
    kv_read = 2 * L * KV * hd * seq_len * tbytes
    kv_write = 2 * L * KV * hd * tbytes
    return dict(weights=w, kv_read=kv_read, kv_write=kv_write, embed=dim * tbytes,
                total=w + kv_read + kv_write + dim * tbytes, p_mm=p_mm)


def dec_kernel_name(args):
    """Host name of the dominant kernel for this configuration (gemv_kernels.hip)."""
    fmt = {4: "i4", 8: "i8", 16: "w"}[args.wbits]
    t = "bfloat" if args.dtype == "bf16" else "float"
    i4bf = args.wbits == 4 and args.dtype == "bf16"
    variant = ("_fast" if args.qmode == "fast" else "_m4d") if i4bf else ""   # decoder.cc gemv(): name selection
    return f"mc_gemv_{fmt}_{t}{variant}_p1_e2"


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary
    (profiles/rNN_pmc_traffic.json, produced by tools/profile_round.sh with separate rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE passes and the gfx950 correction).  Counters cannot be read
    from inside this process, so this is the figure of the last profiled run, or null."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"].get(kernel)
        return k["hbm_bytes_per_launch"] if k else None
    except Exception:
        return None


def cpu_baseline(args, m, tbytes):
    """The oracle (CPU restatement of the reference op sequence) timed on this box's host cores on
    a bounded sample: a 2-layer slice of the same shapes + the full output head, a few tokens,
    scaled to the full depth.  kind = "port" (there is no reference CPU path to run)."""
    import numpy as np

    import modelgen as mg
    from oracle import mc_oracle as mo

    cores = os.cpu_count() or 1
    mo.set_num_threads(cores)
    sample_layers = 2
    cfg = dict(m)
    cfg.update(dtype=0 if args.dtype == "bf16" else 1, family=0, n_layers=sample_layers,
               max_seq_len=64, attn_scale=float(1.0 / np.sqrt(m["head_dim"])))
    quant = {4: "i4", 8: "i8", 16: None}[args.wbits]
    t0 = time.time()
    weights = mg.make_model(cfg, seed=1, quant=quant, group=args.group)
    gen_s = time.time() - t0
    om = mo.Model(cfg, weights)
    tok = 1
    om.step(tok, 0, want_logits=False)  # touch pages
    n = 3
    t0 = time.time()
    for pos in range(1, 1 + n):
        tok, _ = om.step(tok, pos, want_logits=False)
    per_tok = (time.time() - t0) / n
    om.close()
    # time of the head alone ~ vocab*dim MACs vs layer MACs: split by parameter count
    L = m["n_layers"]
    p_layer = m["dim"] * (m["n_heads"] + 2 * m["n_kv_heads"]) * m["head_dim"] + \
        m["n_heads"] * m["head_dim"] * m["dim"] + 3 * m["dim"] * m["ffn_dim"]
    p_head = m["vocab"] * m["dim"]
    t_layer = per_tok * p_layer / (sample_layers * p_layer + p_head)
    t_head = per_tok * p_head / (sample_layers * p_layer + p_head)
    full = L * t_layer + t_head
    return dict(value=1.0 / full, unit="tokens/s", cores=cores, kind="port",
                sample=f"{sample_layers}-layer slice of {args.model} + full output head, {n} tokens at "
                       f"context <= 4, {per_tok:.2f} s/token measured, scaled to {L} layers "
                       f"(weights generated in {gen_s:.0f} s, not timed)")


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with "
                  f"--nproc-per-node {args.gpus}", file=sys.stderr)
            sys.exit(2)
    import numpy as np

    import metalchat_amd as mc

    # N = 1 needs no torch at all: the C ABI owns device memory, stream and events (torch bundles
    # its own ROCm runtime, which also keeps rocprofv3 --pmc from attaching).  N > 1 uses
    # torch.distributed over RCCL for the stage hops.
    torch = None
    dist = None
    piped = world > 1 or args.force_pipeline
    if piped:
        import torch
        import torch.distributed as dist_mod

        if not torch.cuda.is_available():
            print("bench.py: no GPU visible (there is no CPU fallback for the product path)", file=sys.stderr)
            sys.exit(2)
        if args.share_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")  # torch.distributed.run sets both; --force-pipeline alone does not
        if args.share_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    m = MODELS[args.model]
    tbytes = 2 if args.dtype == "bf16" else 4
    dtype = mc.BF16 if args.dtype == "bf16" else mc.F32
    wfmt = {4: mc.WFMT_I4, 8: mc.WFMT_I8, 16: mc.WFMT_T}[args.wbits]
    S, K, W = args.seq_len, args.steps, args.warmup
    if K + W > S:
        print("bench.py: steps + warmup must not exceed seq-len", file=sys.stderr)
        sys.exit(2)

    # N > 1: the decoder's queue adopts a torch-owned (non-default) stream that is also the current
    # stream for the RCCL hops, so torch orders send/recv against the kernels
    tstream = torch.cuda.Stream(device=local_rank) if piped else None
    if tstream is not None:
        torch.cuda.set_stream(tstream)
    acc = mc.HardwareAccelerator(ordinal=local_rank,
                                 stream=tstream.cuda_stream if tstream is not None else None)
    L = m["n_layers"]
    from metalchat_amd.pipeline import layer_range

    lb, le = layer_range(rank, world, L)
    dec = mc.Decoder(acc, dtype=dtype, family=mc.FAMILY_LLAMA3, max_seq_len=S,
                     attn_scale=float(1.0 / np.sqrt(m["head_dim"])), layer_begin=lb, layer_end=le,
                     weight_format=wfmt, group_size=(args.group if args.wbits != 16 else 0),
                     qmode=(mc.QMODE_FAST if args.qmode == "fast" else mc.QMODE_EXACT),
                     use_graph=0 if (args.no_graph or piped) else 1, **m)
    dec.init_synthetic(args.seed)

    fill = S - K - W  # context decoded (untimed) before warm-up so the timed tokens end at S

    def sync_all():
        # barrier + device synchronize on both sides of the timed region
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        acc.wait()

    if not piped:
        tok = 1
        if fill > 0:
            tok = int(dec.generate(tok, 0, fill)[-1])
        if W > 0:
            tok = int(dec.generate(tok, fill, W)[-1])
        sync_all()
        t0 = time.perf_counter()
        toks = dec.generate(tok, fill + W, K)
        sync_all()
        dt = time.perf_counter() - t0
        tmax = dt
    else:
        class _Raw:  # expose a decoder-owned device buffer to torch (CUDA array interface)
            def __init__(self, ptr, n, typestr):
                self.__cuda_array_interface__ = dict(shape=(n,), typestr=typestr, data=(ptr, False),
                                                     version=3)
        # the hidden row travels as raw bytes: uint8 is a type every process-group backend carries
        # (RCCL has no 16-bit unsigned integer type)
        nb = m["dim"] * tbytes
        h_out = torch.as_tensor(_Raw(dec.hidden_out_ptr(), nb, "|u1"), device=f"cuda:{local_rank}")
        h_in = torch.as_tensor(_Raw(dec.hidden_in_ptr(), nb, "|u1"), device=f"cuda:{local_rank}")
        tok_t = torch.zeros(1, dtype=torch.int32, device=f"cuda:{local_rank}")
        from metalchat_amd.pipeline import pipelined_decode

        def stage(token, pos):
            if rank > 0:
                torch.cuda.current_stream().synchronize()  # inbound row landed
            nt = dec.step(token, pos, hidden_in=dec.hidden_in_ptr() if rank > 0 else None,
                          sync=(rank == world - 1))
            acc.wait()  # outbound row complete before the hop is enqueued
            return nt if rank == world - 1 else None

        def run(start, n, tok):
            toks = pipelined_decode(dist, rank, world, h_in, h_out, tok_t, stage, tok, start, n)
            return toks[-1] if toks else tok

        tok = 1
        if fill > 0:
            tok = run(0, fill, tok)
        if W > 0:
            tok = run(fill, W, tok)
        sync_all()
        t0 = time.perf_counter()
        tok = run(fill + W, K, tok)
        sync_all()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax = float(t.item())

    ab = algorithmic_bytes(m, args.wbits, args.group, S, tbytes)
    tok_s = K / tmax
    out = {
        "metric": "decode tokens/s (batch=1) + achieved HBM GB/s vs roofline, int4 Llama-3-8B",
        "value": tok_s, "unit": "tokens/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * tmax / K, "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.model} int{args.wbits} (group={args.group}) batch=1 greedy decode, "
                               f"seq_len={S}, {args.dtype} activations/KV, qmode={args.qmode}",
                   "parallelism": "single GPU" if world == 1 else
                   f"layer pipeline pp{world} ({'gloo hops, ranks sharing GPU 0' if args.share_device else 'RCCL send/recv'})",
                   "hipgraph": bool(not piped and not args.no_graph)},
        "whole_token": {"algorithmic_bytes": ab["total"], "achieved_GBs": ab["total"] * tok_s / 1e9,
                        "frac_of_hbm_peak": ab["total"] * tok_s / 1e9 / HBM_PEAK_GBS},
    }

    if rank == 0 and world == 1 and not args.no_roofline:
        # dominant kernel: the fused rmsnorm + w1|w3 int4 GEMV + SiLU*mul (81 % of the layer bytes
        # are the three FFN matrices).  Launched for all layers back to back (different weights
        # each launch, 1.9 GB >> the 256 MiB Infinity Cache) between two HIP events recorded on
        # the queue's stream.
        reps = 5
        ms, by, ln = dec.time_gemv("w13", reps)
        per = ms / (reps * ln)
        achieved = by / ln / (per * 1e-3) / 1e9
        ms_all, by_all, ln_all = dec.time_gemv("all", reps)
        out["roofline"] = {
            "bound": "hbm", "kernel": "mc_gemv (w1|w3 fused, per launch)", "achieved": achieved,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(dec_kernel_name(args)),
            "bytes_per_launch": by / ln, "avg_launch_us": per * 1e3,
            "all_gemv": {"achieved": by_all * reps / (ms_all * 1e-3) / 1e9,
                         "frac": by_all * reps / (ms_all * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "bytes_per_token": by_all, "ms_per_token": ms_all / reps,
                         "launches_per_token": ln_all},
        }
    if rank == 0 and world == 1 and not args.no_roofline:
        # informational (not `value`): the prompt pass that precedes decoding -- time to first token
        # for a 512-token prompt through mc_decoder_prefill (dequant-once MFMA GEMMs), measured
        # after the timed region on the same decoder; the first call allocates and is not timed
        plen = min(512, S)
        ptoks = np.random.default_rng(1).integers(0, m["vocab"], plen)
        dec.prefill(ptoks, 0)
        acc.wait()
        t0 = time.perf_counter()
        for _ in range(3):
            dec.prefill(ptoks, 0)
        acc.wait()
        pms = (time.perf_counter() - t0) / 3 * 1e3
        lin_params = m["n_layers"] * (m["dim"] * m["n_heads"] * m["head_dim"] * 2
                                      + 2 * m["dim"] * m["n_kv_heads"] * m["head_dim"]
                                      + 3 * m["dim"] * m["ffn_dim"])
        out["prompt_pass"] = {"tokens": plen, "ms": pms, "tokens_per_s": plen / (pms * 1e-3),
                              "linear_TFLOPs": 2.0 * lin_params * plen / (pms * 1e-3) / 1e12,
                              "mfma_peak_TFLOPs": 2500.0}
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, m, tbytes)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
