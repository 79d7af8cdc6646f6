#!/usr/bin/env python3
"""Headline benchmark: single-stream (batch = 1) greedy decode tokens/s of int4 Llama-3-8B on
MI355X, with the achieved HBM rate of the dominant kernel (the fused int4 GEMV) against the
8 TB/s roofline and the CPU restatement of the reference timed beside it.

  python bench.py --gpus N --steps K --warmup W                         (N = 1 default)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (also accepted)

A "step" is one generated token: embedding -> 32 x (rmsnorm, QKV GEMV, RoPE, sink-cache write,
QK^T, softmax, PV, Wo, residual, rmsnorm, w1|w3 GEMV, SiLU*mul, w2 GEMV, residual) -> final norm
-> output-head GEMV -> greedy argmax, exactly what the reference's transform(token, start_pos)
does per token (include/metalchat/transformer.h:357-364).  Weights are synthetic (counter-based
hash, generated on the device), the context is filled by really decoding up to
seq_len - steps before the timed region, and the timed tokens end at position seq_len so the KV
traffic is at its configured maximum.

N > 1: the layers are pipelined over N GPUs (contiguous layer ranges, one RCCL send/recv of the
hidden row per stage boundary and one 4-byte token hop back -- mc_pipeline_* of the C ABI, every
hop enqueued on the decoder's stream), one process per GPU.  Launched without a launcher,
`python bench.py --gpus N` starts its N workers itself (fresh child processes, before anything
touches the GPU) and hands the RCCL id over through a file; under torch.distributed.run it reads
RANK / LOCAL_RANK / WORLD_SIZE from the environment.  torch is never imported.  At batch 1 only
one stage is busy at a time, so this buys capacity (70B fits), not speed: the same model is
split over more GPUs, i.e. scaling is "strong".
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
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
    # the reference's default model (src/llama.cc:19-31) and the decode layer MI355X_MICROARCH.md prices
    # ("launches-baseline": 121.6 MB of bf16 weights per layer in 30.6 us as five captured launches)
    "llama3.2-1b": dict(dim=2048, n_heads=32, n_kv_heads=8, head_dim=64, ffn_dim=8192, n_layers=16,
                        vocab=128256, rope_theta=500000.0, norm_eps=1e-5),
    "gemma-7b": dict(dim=3072, n_heads=16, n_kv_heads=16, head_dim=256, ffn_dim=24576, n_layers=28,
                     vocab=256000, rope_theta=10000.0, norm_eps=1e-5),
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
    p.add_argument("--stage-graph", action="store_true",
                   help="N > 1: replay one hipGraph per pipeline stage instead of launching eagerly (measured round 4 on one "
                        "device: 788 vs 796 tokens/s at N = 2, 763 vs 775 at N = 4 -- the eager path is the default)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-other-configs", action="store_true")
    p.add_argument("--seed", type=int, default=0x5EED)
    p.add_argument("--share-device", action="store_true",
                   help="validation aid for a 1-GPU box: the N stages run in THIS process on GPU 0 "
                        "(mc_pipeline_create_local: device-to-device hops), the same launches per stage "
                        "as the RCCL transport")
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


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/rNN_pmc_traffic.json,
    produced by tools/profile_round.sh with separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes and the gfx950
    correction).  Counters cannot be read from inside this process, so this is the figure of the last PROFILED
    build: the entry names its file and the git HEAD it was taken at, and is null when the file has no such kernel."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        doc = json.load(open(files[-1]))
        k = doc["kernels"].get(kernel)
        if not k:
            return None
        return dict(hbm_bytes_per_launch=k["hbm_bytes_per_launch"], file=os.path.relpath(files[-1], ROOT),
                    git_head=doc.get("git_head"), kernel=kernel)
    except Exception:
        return None


def gemv_ablations(kernel):
    """What bounds `kernel` as a launch of its own, from the newest committed ablation summary (profiles/rNN_gemv_ablations.json: the product build
    next to a build without weight traffic and one without arithmetic, tools/mall_probe.py): informational, measured on another box than this run."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemv_ablations.json")))
    if not files:
        return None
    try:
        doc = json.load(open(files[-1]))
        k = doc["kernels"].get(kernel)
        if not k:
            return None
        b = k["bytes_per_launch"]
        return dict(full_us=k["full_us"], compute_only_us=k["compute_only_us"], stream_only_us=k["stream_only_us"],
                    stream_only_frac=b / (k["stream_only_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, file=os.path.relpath(files[-1], ROOT), git_head=doc.get("git_head"))
    except Exception:
        return None


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container on
    a 256-core host is often given a fraction of it; an OpenMP team of one thread per HOST core then spends its
    time being descheduled inside barriers)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, q // per))
        except Exception:
            pass
    return n


def native_oracle():
    """The oracle built -O3 -march=native ON THIS HOST for the cpu_baseline leg (the .so that ships with the
    repository is a portable -O2 build: it must run on whatever CPU the box has).  Falls back to the portable
    build when there is no compiler."""
    odir = os.path.join(ROOT, "oracle")
    so = os.path.join(odir, "_build", "libmc_oracle_native.so")
    flags = "-O3 -march=native -fPIC -std=c11 -fno-fast-math -ffp-contract=off -fopenmp"
    try:
        src = os.path.join(odir, "mc_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(so), exist_ok=True)
            subprocess.check_call(["gcc"] + flags.split() + ["-shared", "-o", so, src, "-lm"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.environ["MC_ORACLE_SO"] = so
        return flags
    except Exception:
        return "-O2 (portable build shipped with the repository; no compiler on this host)"


def cpu_baseline(args, m, tbytes):
    """The oracle (CPU restatement of the reference op sequence) timed on this box's host cores on
    bounded samples.  kind = "port" (there is no reference CPU path to run).
      * the selected model: a 2-layer slice of the same shapes + the full output head, a few tokens at context
        <= 4 (no KV traffic to speak of), scaled to the full depth;
      * BASELINE configs[0] in FULL: TinyLlama-1.1B, 22 layers, bf16 weights, vocabulary 32000,
        random weights, a few tokens from position 0."""
    import numpy as np

    flags = native_oracle()
    import modelgen as mg
    from oracle import mc_oracle as mo

    cores = usable_cores()
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
    del weights
    # time of the head alone ~ vocab*dim MACs vs layer MACs: split by parameter count
    L = m["n_layers"]
    p_layer = m["dim"] * (m["n_heads"] + 2 * m["n_kv_heads"]) * m["head_dim"] + \
        m["n_heads"] * m["head_dim"] * m["dim"] + 3 * m["dim"] * m["ffn_dim"]
    p_head = m["vocab"] * m["dim"]
    t_layer = per_tok * p_layer / (sample_layers * p_layer + p_head)
    t_head = per_tok * p_head / (sample_layers * p_layer + p_head)
    full = L * t_layer + t_head
    out = dict(value=1.0 / full, unit="tokens/s", cores=cores, kind="port", compiler_flags=flags,
               sample=f"{sample_layers}-layer slice of {args.model} + full output head, {n} tokens at context <= 4 "
                      f"(KV traffic negligible), {per_tok:.2f} s/token measured, scaled to {L} layers "
                      f"(weights generated in {gen_s:.0f} s, not timed)")
    # ---- TinyLlama-1.1B in full
    try:
        tm = MODELS["tinyllama-1.1b"]
        tcfg = dict(tm)
        tcfg.update(dtype=0, family=0, max_seq_len=64, attn_scale=float(1.0 / np.sqrt(tm["head_dim"])))
        rng = np.random.default_rng(7)

        def lin(o, i):  # bf16 bits of N(0, 1/sqrt(in)) weights: the truncated upper half of the float
            w = rng.standard_normal((o, i), dtype=np.float32) * np.float32(1.0 / np.sqrt(i))
            return dict(kind=0, weight=(w.view(np.uint32) >> 16).astype(np.uint16))

        def vec(k):
            return mo.encode(0, rng.uniform(0.5, 1.5, k).astype(np.float32))

        t0 = time.time()
        dim, H, KV, hd, ffn = tm["dim"], tm["n_heads"], tm["n_kv_heads"], tm["head_dim"], tm["ffn_dim"]
        layers = [dict(wq=lin(H * hd, dim), wk=lin(KV * hd, dim), wv=lin(KV * hd, dim), wo=lin(dim, H * hd),
                       w1=lin(ffn, dim), w2=lin(dim, ffn), w3=lin(ffn, dim), attention_norm=vec(dim), ffn_norm=vec(dim))
                  for _ in range(tm["n_layers"])]
        tw = dict(layers=layers, embedding=lin(tm["vocab"], dim), output=lin(tm["vocab"], dim), final_norm=vec(dim))
        tgen = time.time() - t0
        om = mo.Model(tcfg, tw)
        tok = 1
        om.step(tok, 0, want_logits=False)
        nt = 2
        t0 = time.time()
        for pos in range(1, 1 + nt):
            tok, _ = om.step(tok, pos, want_logits=False)
        tper = (time.time() - t0) / nt
        om.close()
        ab = algorithmic_bytes(tm, 16, 0, nt, 2)
        out["tinyllama_1_1b_full"] = dict(value=1.0 / tper, unit="tokens/s", cores=cores, kind="port",
                                          effective_GBs=ab["total"] / tper / 1e9,
                                          sample=f"all 22 layers + head, bf16 weights (random, generated in {tgen:.0f} s, not timed), "
                                                 f"{nt} tokens at context <= {nt + 1}, {tper:.3f} s/token")
    except MemoryError:
        out["tinyllama_1_1b_full"] = None
    return out


def spawn_workers(args):
    """`python bench.py --gpus N` without a launcher: N fresh child processes, one per GPU, started BEFORE this
    process touches the GPU; rank 0 prints the JSON line.  The RCCL id travels through a file."""
    import shutil

    n = args.gpus
    uid_dir = tempfile.mkdtemp(prefix="mc_bench_")
    uid_file = os.path.join(uid_dir, "rccl_uid")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MC_UID_FILE=uid_file,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # a worker that dies (no such device, RCCL error) would leave its peers waiting in the communicator: watch
    # them and take the rest down
    rc = 0
    live = list(procs)
    t0 = time.time()
    while live:
        time.sleep(0.2)
        if time.time() - t0 > float(os.environ.get("MC_BENCH_DEADLINE_S", "3000")):
            # (a rank that waits in a collective for a peer that will never come: end the job instead of hanging)
            print("bench.py: the ranks did not finish inside the deadline", file=sys.stderr)
            rc = 4
            for q in live:
                q.kill()
            break
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0:
                rc = r
                for q in live:
                    q.kill()
    shutil.rmtree(uid_dir, ignore_errors=True)
    sys.exit(rc)


_UID_FILES = []


def exchange_uid(mc, rank, world, tag=""):
    """Rank 0 makes the 128-byte RCCL id; the others read it from a file rank 0 wrote atomically.  `tag` names a
    second communicator of the same job (the 70B stages).  The file's NAME is unique per launch -- spawn_workers makes a
    directory of its own; under torch.distributed.run the run id, the restart count, MASTER_PORT and the launcher's pid -- so a
    file that exists IS this launch's (no age heuristic: a rank that starts late must not reject a valid id); rank 0 removes
    it at the end of the job (cleanup_uid_files)."""
    path = os.environ.get("MC_UID_FILE")
    if not path:
        key = "_".join((os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"),
                        os.environ.get("MASTER_PORT", "29500"), str(os.getppid())))
        path = os.path.join(tempfile.gettempdir(), f"mc_bench_uid_{os.getuid()}_{key}")
    path += tag
    if rank == 0:
        uid = mc.pipeline_unique_id()
        tmp = path + f".{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        _UID_FILES.append(path)
        return uid
    t0 = time.time()
    while time.time() - t0 < 120:
        try:
            if os.path.getsize(path) == 128:
                with open(path, "rb") as f:
                    return f.read()
        except OSError:
            pass
        time.sleep(0.05)
    print("bench.py: no RCCL id from rank 0 after 120 s", file=sys.stderr)
    sys.exit(2)


def cleanup_uid_files():
    for p in _UID_FILES:
        try:
            os.unlink(p)
        except OSError:
            pass


def run_other_configs(mc, acc, np):
    """The other BASELINE.json configurations, briefly, on this one GPU (informational, driver-observed): each
    builds its own decoder, STARTS at position seq_len - tokens - 8 (the earlier cache rows are zero: the context
    is not decoded, the KV / weight traffic per token is what the position implies) and times `tokens` chained
    greedy tokens after 64 warm-up tokens (8 read 2 % low against tools/configs_run.py, which decodes the whole context
    first: the chip has not reached its clocks behind a weight fill)."""
    cases = [
        ("TinyLlama-1.1B bf16 weights, S=2048 (configs[0] on the GPU)", "tinyllama-1.1b", 16, 0, 2048, 128, 0, mc.FAMILY_LLAMA3),
        ("Llama-3-8B int8 g128, S=8192, 64 of the tokens past max_seq_len (configs[2])", "llama3-8b", 8, 128, 8192, 64, 64, mc.FAMILY_LLAMA3),
        ("Gemma-7B shapes int4 g128, gemma3 block, S=2048 (configs[3])", "gemma-7b", 4, 128, 2048, 64, 0, mc.FAMILY_GEMMA3),
        ("Llama-3-70B int4 g128 on ONE GPU, S=2048 (configs[4] without the pipeline)", "llama3-70b", 4, 128, 2048, 32, 0, mc.FAMILY_LLAMA3),
    ]
    out = []
    for name, model, wbits, group, S, K, past, fam in cases:
        dec = None
        try:
            m = MODELS[model]
            extra = dict(rope_sliding_theta=10000.0, sliding_stride=6) if fam == mc.FAMILY_GEMMA3 else {}
            dec = mc.Decoder(acc, dtype=mc.BF16, family=fam, max_seq_len=S, attn_scale=float(1 / np.sqrt(m["head_dim"])),
                             weight_format={4: mc.WFMT_I4, 8: mc.WFMT_I8, 16: mc.WFMT_T}[wbits], group_size=group,
                             use_graph=1, **extra, **m)
            dec.init_synthetic(7)
            WARM = 64
            start = S - K - WARM
            tok = int(dec.generate(1, start, WARM)[-1])
            acc.wait()
            t0 = time.perf_counter()
            dec.generate(tok, start + WARM, K + past)
            acc.wait()
            dt = time.perf_counter() - t0
            ab = algorithmic_bytes(m, wbits, group or 1, S, 2)
            tps = (K + past) / dt
            # ("cache": the rows in front of the timed tokens were never decoded -- the traffic is the position's, the softmax runs over zero rows)
            out.append(dict(config=name, tokens_per_s=tps, ms_per_token=dt / (K + past) * 1e3, tokens=K + past, cache="zero-filled",
                            algorithmic_bytes=ab["total"], frac_of_hbm_peak=ab["total"] * tps / 1e9 / HBM_PEAK_GBS))
        except Exception as e:  # an informational leg must not take the headline down
            out.append(dict(config=name, error=str(e)[:200]))
        finally:
            if dec is not None:
                try:
                    dec.release()
                except Exception:
                    pass
    out.append(run_local_pipeline_70b(mc, acc, np))
    return out


def run_local_pipeline_70b(mc, acc, np, world=8):
    """BASELINE configs[4] as far as ONE GPU can show it: Llama-3-70B int4 g128 cut into `world` stages of
    80 / world layers (include/metalchat/nn/llama.h:123-126), all of them on this device in this process -- mc_pipeline_create_local:
    the launches of every stage are the RCCL transport's, a hop is a device-to-device copy of the hidden row behind an event instead
    of ncclSend / ncclRecv.  Next to the single-stage 70B number above it prices the pipeline's bookkeeping (7 row hops + the token
    hop per token); it says nothing about xGMI."""
    name = f"Llama-3-70B int4 g128 layer-pipelined pp{world}, all stages sharing this GPU (device-to-device hops), S=2048 (configs[4])"
    stages, pipe = [], None
    try:
        m = MODELS["llama3-70b"]
        S, K = 2048, 32
        for r in range(world):
            b, e = mc.pipeline_layer_range(r, world, m["n_layers"])
            d = mc.Decoder(acc, dtype=mc.BF16, family=mc.FAMILY_LLAMA3, max_seq_len=S, attn_scale=float(1.0 / np.sqrt(m["head_dim"])),
                           layer_begin=b, layer_end=e, weight_format=mc.WFMT_I4, group_size=128, use_graph=0, **m)
            d.init_synthetic(7)
            stages.append(d)
        pipe = mc.Pipeline.local(stages)
        WARM = 32
        start = S - K - WARM
        tok = int(pipe.generate(1, start, WARM)[-1])
        acc.wait()
        t0 = time.perf_counter()
        pipe.generate(tok, start + WARM, K)
        acc.wait()
        dt = time.perf_counter() - t0
        ab = algorithmic_bytes(m, 4, 128, S, 2)
        tps = K / dt
        return dict(config=name, tokens_per_s=tps, ms_per_token=dt / K * 1e3, tokens=K, layers_per_stage=m["n_layers"] // world,
                    algorithmic_bytes=ab["total"], frac_of_hbm_peak=ab["total"] * tps / 1e9 / HBM_PEAK_GBS,
                    hop_transport="device-to-device copy behind an event", hipgraph=False, cache="zero-filled")
    except Exception as e:
        return dict(config=name, error=str(e)[:200])
    finally:   # (whatever happened: the stages built so far give their HBM back)
        try:
            if pipe is not None:
                pipe.release()
        except Exception:
            pass
        for d in stages:
            try:
                d.release()
            except Exception:
                pass


def build_70b_stage(mc, acc, np, args, rank, world):
    """This rank's stage(s) of BASELINE configs[4] (everything that can fail locally before the first collective)."""
    m = MODELS["llama3-70b"]

    def stage(r):
        b, e = mc.pipeline_layer_range(r, world, m["n_layers"])
        d = mc.Decoder(acc, dtype=mc.BF16, family=mc.FAMILY_LLAMA3, max_seq_len=2048, attn_scale=float(1.0 / np.sqrt(m["head_dim"])),
                       layer_begin=b, layer_end=e, weight_format=mc.WFMT_I4, group_size=128, use_graph=1 if args.stage_graph else 0, **m)
        d.init_synthetic(7)
        return d

    return [stage(r) for r in range(world)] if args.share_device else [stage(rank)]


def run_pipelined_70b(mc, acc, np, args, rank, world, stages):
    """BASELINE configs[4]: Llama-3-70B int4 g128, layers pipelined over the `world` stages of this job (80 / world
    layers and their caches per stage, include/metalchat/nn/llama.h:123-126 cut into stages), S = 2048, batch 1 greedy.
    Runs after the headline timing with a communicator of its own; the value is absolute tokens/s and the fraction of the
    219 tokens/s a single 8 TB/s stream of the model's 36.5 GB allows -- at batch 1 one stage is busy at a time, so the
    aggregate bandwidth of the N GPUs is not the ceiling."""
    m = MODELS["llama3-70b"]
    S, K = 2048, 32
    lb, le = mc.pipeline_layer_range(rank, world, m["n_layers"])
    if args.share_device:
        pipe = mc.Pipeline.local(stages)
    else:
        pipe = mc.Pipeline.rccl(stages[0], rank, world, exchange_uid(mc, rank, world, tag="_70b"))
    ranks_seen = pipe.comm_info()
    start = S - K - 8
    tok = int(pipe.generate(1, start, 8)[-1])
    pipe.allreduce_max(0.0)
    t0 = time.perf_counter()
    pipe.generate(tok, start + 8, K)
    pipe.allreduce_max(0.0)
    dt = pipe.allreduce_max(time.perf_counter() - t0)
    pipe.release()
    for d in stages:
        d.release()
    ab = algorithmic_bytes(m, 4, 128, S, 2)
    tps = K / dt
    return dict(config=f"Llama-3-70B int4 g128 layer-pipelined pp{world}, S=2048 (configs[4]), "
                       f"{'stages sharing GPU 0, device-to-device hops' if args.share_device else 'one process per GPU, RCCL send/recv per hop'}",
                tokens_per_s=tps, ms_per_token=dt / K * 1e3, tokens=K, layers_per_stage=le - lb, algorithmic_bytes=ab["total"],
                single_stream_ceiling_tokens_per_s=HBM_PEAK_GBS * 1e9 / ab["total"],
                frac_of_hbm_peak=ab["total"] * tps / 1e9 / HBM_PEAK_GBS, hipgraph=False,
                rccl={"ranks": ranks_seen[0], "rank": ranks_seen[1]})


def main():
    args = parse()
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and env_world != args.gpus and not args.share_device:
        spawn_workers(args)  # does not return
    rank = int(os.environ.get("RANK", "0")) if not args.share_device else 0
    world = args.gpus
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if not args.share_device else 0
    import numpy as np

    import metalchat_amd as mc

    m = MODELS[args.model]
    tbytes = 2 if args.dtype == "bf16" else 4
    dtype = mc.BF16 if args.dtype == "bf16" else mc.F32
    wfmt = {4: mc.WFMT_I4, 8: mc.WFMT_I8, 16: mc.WFMT_T}[args.wbits]
    S, K, W = args.seq_len, args.steps, args.warmup
    if K + W > S:
        print("bench.py: steps + warmup must not exceed seq-len", file=sys.stderr)
        sys.exit(2)

    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    acc = mc.HardwareAccelerator(ordinal=local_rank)
    L = m["n_layers"]
    piped = world > 1

    def stage(r):
        lb, le = mc.pipeline_layer_range(r, world, L)
        d = mc.Decoder(acc, dtype=dtype, family=mc.FAMILY_LLAMA3, max_seq_len=S,
                       attn_scale=float(1.0 / np.sqrt(m["head_dim"])), layer_begin=lb, layer_end=le,
                       weight_format=wfmt, group_size=(args.group if args.wbits != 16 else 0),
                       qmode=(mc.QMODE_FAST if args.qmode == "fast" else mc.QMODE_EXACT),
                       use_graph=(1 if args.stage_graph else 0) if piped else (0 if args.no_graph else 1), **m)
        d.init_synthetic(args.seed)
        return d

    pipe = None
    if not piped:
        dec = stage(0)
        gen = dec.generate
    elif args.share_device:
        stages = [stage(r) for r in range(world)]
        dec = stages[0]
        pipe = mc.Pipeline.local(stages)
        gen = pipe.generate
    else:
        dec = stage(rank)
        pipe = mc.Pipeline.rccl(dec, rank, world, exchange_uid(mc, rank, world))
        gen = pipe.generate

    def sync_all(v=0.0):
        # barrier over the ranks + device synchronize; returns max(v) over the ranks
        if pipe is not None:
            return pipe.allreduce_max(v)
        acc.wait()
        return v

    fill = S - K - W  # context decoded (untimed) before warm-up so the timed tokens end at S
    tok = 1
    if fill > 0:
        tok = int(gen(tok, 0, fill)[-1])
    if W > 0:
        tok = int(gen(tok, fill, W)[-1])
    sync_all()
    t0 = time.perf_counter()
    tok = int(gen(tok, fill + W, K)[-1])
    sync_all()
    dt = time.perf_counter() - t0
    tmax = sync_all(dt)
    # Two more chains of the same K tokens, continuing past position seq_len (the sink ring turns: every token still attends to
    # max_seq_len rows, the same traffic): `value` stays the FIRST chain -- steps x ms_per_step is the region the driver can check --
    # and `value_spread` says how far apart three identical chains of this process land
    chain_tps = [K / tmax]
    for c in (1, 2):
        sync_all()
        t0 = time.perf_counter()
        tok = int(gen(tok, fill + W + c * K, K)[-1])
        sync_all()
        chain_tps.append(K / sync_all(time.perf_counter() - t0))

    ab = algorithmic_bytes(m, args.wbits, args.group, S, tbytes)
    tok_s = K / tmax
    out = {
        "metric": "decode tokens/s (batch=1) + achieved HBM GB/s vs roofline, int4 Llama-3-8B",
        "value": tok_s, "unit": "tokens/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * tmax / K, "higher_is_better": True,
        "value_spread": [min(chain_tps), max(chain_tps)], "chains_tokens_per_s": chain_tps,
        "scaling": "strong", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.model} int{args.wbits} (group={args.group}) batch=1 greedy decode, "
                               f"seq_len={S}, {args.dtype} activations/KV, qmode={args.qmode}",
                   "parallelism": "single GPU" if world == 1 else
                   f"layer pipeline pp{world} ({'stages sharing GPU 0 in one process, device-to-device hops' if args.share_device else 'one process per GPU, RCCL send/recv on the decoder stream'})",
                   "hipgraph": bool(args.stage_graph if piped else not args.no_graph),
                   # (N > 1: --stage-graph replays one graph per stage -- the launches of a token between two hops; with three
                   #  launches per layer the eager path keeps the queue full and measured 1-1.5 % faster: the default)
                   "hop_transport": None if not piped else ("device-to-device copy behind an event" if args.share_device else "ncclSend / ncclRecv (RCCL) on the decoder stream")},
        "whole_token": {"algorithmic_bytes": ab["total"], "achieved_GBs": ab["total"] * tok_s / 1e9,
                        "frac_of_hbm_peak": ab["total"] * tok_s / 1e9 / HBM_PEAK_GBS},
    }
    if piped:
        # what the transport itself says (ncclCommCount / ncclCommUserRank; (-1, -1): stages sharing one process): the driver
        # can check that RCCL saw N ranks
        n_seen, r_seen = pipe.comm_info()
        out["rccl"] = {"ranks": n_seen, "rank": r_seen}

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
        per_kind = {}
        for which in ("qkv", "wo", "w2", "head"):
            msk, byk, lnk = dec.time_gemv(which, reps)
            pk = msk / (reps * lnk)
            per_kind[which] = {"kernel": dec.gemv_kernel_name(which), "avg_launch_us": pk * 1e3, "bytes_per_launch": byk / lnk,
                               "frac": byk / lnk / (pk * 1e-3) / 1e9 / HBM_PEAK_GBS}
        attn_kernel = dec.gemv_kernel_name("attn")
        inside = ("qkv", "wo") if attn_kernel.startswith("mc_attn_qkv_wo_") else ("wo",) if attn_kernel.startswith("mc_attn_wo_") else ()
        for which in inside:
            per_kind[which]["note"] = f"stand-alone launch, timed for reference: in the token this GEMV runs inside {attn_kernel}"
        kname = dec.gemv_kernel_name("w13")  # the name decoder.cc gemv() selects, asked of the decoder itself
        out["roofline"] = {
            "bound": "hbm", "kernel": f"{kname} (w1|w3 fused, per launch)", "achieved": achieved,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(kname),
            # (what bounds this kernel as a launch of its own: the same launch without arithmetic / without weight traffic, profiles/rNN_gemv_ablations.json)
            "ablations": gemv_ablations(kname),
            "bytes_per_launch": by / ln, "avg_launch_us": per * 1e3,
            "other_gemvs": per_kind,
            "attention_kernel": attn_kernel,
            "all_gemv": {"achieved": by_all * reps / (ms_all * 1e-3) / 1e9,
                         "frac": by_all * reps / (ms_all * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "bytes_per_token": by_all, "ms_per_token": ms_all / reps,
                         "launches_per_token": ln_all},
        }
        # informational (not `value`): the prompt pass that precedes decoding -- time to first token
        # for a 512-token prompt through mc_decoder_prefill (hand-written MFMA GEMMs; since round 6 on a dequantised bfloat16 copy of each quantised
        # matrix, built in the first call -- `extra_hbm_bytes` -- where it fits an eighth of the device: DESIGN.md section 7), measured after the timed region on the same decoder; `first_call_ms` is the first prompt of the process
        # (allocations included), `ms` the mean of the next three, `extra_hbm_bytes` what derived weight copies hold afterwards
        plen = min(512, S)
        ptoks = np.random.default_rng(1).integers(0, m["vocab"], plen)
        acc.wait()
        t0 = time.perf_counter()
        dec.prefill(ptoks, 0)   # the FIRST prompt of the process: row buffers allocated, the table of exponentials built
        acc.wait()
        first_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        for _ in range(3):
            dec.prefill(ptoks, 0)
        acc.wait()
        pms = (time.perf_counter() - t0) / 3 * 1e3
        lin_params = m["n_layers"] * (m["dim"] * m["n_heads"] * m["head_dim"] * 2
                                      + 2 * m["dim"] * m["n_kv_heads"] * m["head_dim"]
                                      + 3 * m["dim"] * m["ffn_dim"])
        # which of its GEMMs went to the ROCm library (decoder.cc gemm_lib: launches with >= 48 tiles of 256 x 256 multiply a
        # dequantised bfloat16 copy of the matrix in hipBLASLt; the others are the hand-written prompt kernels) -- from the launch log
        dec.launch_log(True)
        dec.prefill(ptoks, 0)
        acc.wait()
        pnames = dec.launched()
        dec.launch_log(False)
        out["prompt_pass"] = {"tokens": plen, "ms": pms, "first_call_ms": first_ms, "extra_hbm_bytes": dec.derived_weight_bytes(),
                              "tokens_per_s": plen / (pms * 1e-3),
                              "linear_TFLOPs": 2.0 * lin_params * plen / (pms * 1e-3) / 1e12,
                              "mfma_peak_TFLOPs": 2500.0,
                              "gemm_launches": {"hipBLASLt": pnames.count("hipblasLtMatmul"),
                                                "hand_written": sum(1 for x in pnames if x.startswith("mc_pf_gemm") or x.startswith("mc_pf2_gemm"))}}
    if rank == 0 and world == 1 and not args.no_other_configs and args.model == "llama3-8b":
        if pipe is None:
            dec.release()
        out["other_configs"] = run_other_configs(mc, acc, np)
    if piped and not args.no_other_configs and args.model == "llama3-8b":
        # every rank takes part; rank 0 reports.  A rank that fails ALONE would leave its peers waiting in the new communicator
        # for ever (the launcher only ends a job whose worker exits non-zero), so: (1) what can fail locally before the first
        # collective -- building the 70B stage: allocations -- is agreed on over the headline communicator, which is still
        # alive: one failure and every rank skips the leg; (2) a failure behind that point (RCCL bring-up, a hop) ends the JOB
        # with a non-zero exit -- rank 0 prints the headline line first, also when it is the launcher that takes it down (SIGTERM).
        import signal

        label = f"Llama-3-70B int4 g128 layer-pipelined pp{world}"
        build_err = None
        stage70 = None
        try:
            stage70 = build_70b_stage(mc, acc, np, args, rank, world)
        except Exception as e:
            build_err = str(e)[:300]
        failed = pipe.allreduce_max(1.0 if build_err else 0.0)
        pipe.release()
        for d in (stages if args.share_device else [dec]):
            d.release()
        pipe = None
        if failed:
            out["other_configs"] = [dict(config=label, error=build_err or "another rank could not build its stages")]
        else:
            def headline_then_die(*_):
                out["other_configs"] = [dict(config=label, error="the leg failed on another rank; the job was ended")]
                if rank == 0:
                    print(json.dumps(out), flush=True)
                os._exit(3)

            signal.signal(signal.SIGTERM, headline_then_die)
            try:
                out["other_configs"] = [run_pipelined_70b(mc, acc, np, args, rank, world, stage70)]
            except Exception as e:
                out["other_configs"] = [dict(config=label, error=str(e)[:300])]
                if rank == 0:
                    print(json.dumps(out), flush=True)
                sys.stdout.flush()
                os._exit(3)  # (non-zero: spawn_workers / torchrun end the peers that wait for this rank)
            signal.signal(signal.SIGTERM, signal.SIG_DFL)
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, m, tbytes)
    if rank == 0:
        print(json.dumps(out), flush=True)
    sync_all()
    cleanup_uid_files()


if __name__ == "__main__":
    main()
