#!/usr/bin/env python3
"""Looks for performance cliffs off the headline configuration: sampler chain, T = float, short and
mid contexts (Llama-3-8B shapes, int4 g128, hipGraph chaining)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import metalchat_amd as mc

M = dict(dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=32, vocab=128256, rope_theta=500000.0)
acc = mc.HardwareAccelerator()


def run(name, dtype=mc.BF16, S=2048, sampler=None, fill=None, K=64, layers=32):
    m = dict(M, n_layers=layers)
    dec = mc.Decoder(acc, dtype=dtype, family=mc.FAMILY_LLAMA3, max_seq_len=S, norm_eps=1e-5, attn_scale=128 ** -0.5,
                     weight_format=mc.WFMT_I4, group_size=128, use_graph=1, **m)
    dec.init_synthetic(3)
    if sampler:
        dec.set_sampler(mc.SAMPLER_DEFAULT, 50, 0.6, 0.9)
        dec.set_seeds(np.arange(64, dtype=np.uint64))
    fill = S - K - 8 if fill is None else fill
    tok = 1
    if fill:
        tok = int(dec.generate(tok, 0, fill)[-1])
    tok = int(dec.generate(tok, fill, 8)[-1])
    acc.wait()
    t0 = time.perf_counter()
    dec.generate(tok, fill + 8, K)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(case=name, tokens_per_s=round(K / dt * (32 / layers), 1), ms_per_token=round(dt / K * 1e3 * (32 / layers), 3))), flush=True)
    dec.release()


run("greedy S=2048 (headline)")
run("default sampler (top-k 50, nucleus) S=2048", sampler=True)
run("context 64..136 of max 2048", fill=64)
run("context 512 of max 2048", fill=512)
run("max_seq 4096 full", S=4096)
run("T=float, 8 layers scaled to 32", dtype=mc.F32, layers=8)


def run_steps(name, use_graph):
    dec = mc.Decoder(acc, dtype=mc.BF16, family=mc.FAMILY_LLAMA3, max_seq_len=2048, norm_eps=1e-5, attn_scale=128 ** -0.5,
                     weight_format=mc.WFMT_I4, group_size=128, use_graph=use_graph, **M)
    dec.init_synthetic(3)
    tok = int(dec.generate(1, 0, 1900)[-1])
    for i in range(8):
        tok = dec.step(tok, 1900 + i)
    t0 = time.perf_counter()
    for i in range(64):
        tok = dec.step(tok, 1908 + i)   # one host round trip per token: what interpreter::read_until does
    dt = time.perf_counter() - t0
    print(json.dumps(dict(case=name, tokens_per_s=round(64 / dt, 1), ms_per_token=round(dt / 64 * 1e3, 3))), flush=True)
    dec.release()


run_steps("mc_decoder_step per token, use_graph=1", 1)
run_steps("mc_decoder_step per token, use_graph=0", 0)
