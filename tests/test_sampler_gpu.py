"""Sampler chain on the GPU.

Part A: the reference-named kernels sort / cumsum / multinomial / gt / le / scatter / gather / sub
driven through the encoder seam with the reference's launch geometry
(include/metalchat/kernel/{sort,sum,multinomial,logical,copy}.h), against the reference's own test
properties (test/test_kernel_sort.cc, test_kernel_sum.cc, test_kernel_multinomial.cc) AND
bit-exactly against the oracle.
Part B: the fused default sampler (topk -> nucleus -> multinomial in two launches) against
mco_sample_default: every intermediate of the chain and the token, bit-exact; then through the
decoder (step and chained generate)."""
import numpy as np
import pytest

import modelgen as mg
import parity
from oracle import mc_oracle as mo

pytestmark = pytest.mark.gpu
BF16, F32 = 0, 1
TN = {BF16: "bfloat", F32: "float"}


def L(*a, **k):
    import metalchat_amd as mc

    return mc.layout(*a, **k)


def ceil_pow2(n):
    p = 1
    while p < n:
        p *= 2
    return p


def gpu_sort(acc, dt, x):
    """kernel::sort::operator() (include/metalchat/kernel/sort.h:27-62)"""
    import metalchat_amd as mc

    rows, dim = x.shape
    aligned = ceil_pow2(dim)
    k = acc.load("sort", TN[dt])
    block = -(-aligned // k.max_threads_per_threadgroup())
    threads = -(-aligned // block)
    vb = acc.alloc(rows * aligned * x.itemsize)
    ib = acc.alloc(rows * aligned * 4)
    mc.KernelTask(k, (threads * rows, 1, 1), (threads, 1, 1),
                  [L((rows, aligned)), vb, L((rows, aligned)), ib, L(x.shape), acc.to_device(x), np.uint32(block)])()
    acc.wait()
    return (vb.download(x.dtype, rows * aligned).reshape(rows, aligned),
            ib.download(np.int32, rows * aligned).reshape(rows, aligned))


@pytest.mark.parametrize("dt,dim", [(F32, 100000), (BF16, 128256), (F32, 50), (BF16, 64)])
def test_sort_matches_reference_properties_and_oracle(acc, dt, dim):
    # test/test_kernel_sort.cc:17-50: descending, and indices map the input onto the output
    x = mo.encode(dt, np.random.default_rng(1).uniform(size=(1, dim)).astype(np.float32))
    v, idx = gpu_sort(acc, dt, x)
    vs = mo.decode(dt, v[:, :dim])
    assert np.all(vs[0, :-1] >= vs[0, 1:])
    assert np.array_equal(x[0, idx[0, :dim]], v[0, :dim])
    al = ceil_pow2(dim)
    rv, ri = np.zeros((1, al), x.dtype), np.zeros((1, al), np.int32)
    mo.sort(dt, mo.layout(rv.shape), rv, mo.layout(ri.shape), ri, mo.layout(x.shape), x)
    parity.exact(v, rv, "sort values")
    parity.exact(idx, ri, "sort indices (same network, same tie handling)")


@pytest.mark.parametrize("dt,dim", [(F32, 400), (BF16, 400), (F32, 5000), (BF16, 50)])
def test_cumsum_matches_reference_property_and_oracle(acc, dt, dim):
    # test/test_kernel_sum.cc:17-40; launch maths of include/metalchat/kernel/sum.h:28-58
    import metalchat_amd as mc

    x = mo.encode(dt, np.random.default_rng(2).uniform(size=(2, dim)).astype(np.float32))
    B = max(2, ceil_pow2(-(-dim // 1024)))
    threads = -(-dim // B)
    k = acc.load(f"cumsum_{B}", TN[dt])
    out = acc.alloc(x.size * x.itemsize)
    mc.KernelTask(k, (threads * 2, 1, 1), (threads, 1, 1), [L(x.shape), out, L(x.shape), acc.to_device(x)])()
    acc.wait()
    got = out.download(x.dtype, x.size).reshape(x.shape)
    ref = np.zeros_like(x)
    mo.cumsum(dt, mo.layout(x.shape), ref, mo.layout(x.shape), x)
    parity.exact(got, ref, "cumsum")
    if dt == F32:
        expect = np.zeros_like(x)
        for r in range(2):
            acc_ = np.float32(0)
            for j in range(dim):
                acc_ = np.float32(acc_ + x[r, j])
                expect[r, j] = acc_
        assert np.allclose(got, expect, rtol=1.2e-5 * 100, atol=1e-4)   # Catch Approx(...).margin(1e-4)


def test_multinomial_reference_experiment_and_oracle(acc):
    # test/test_kernel_multinomial.cc:17-55
    import metalchat_amd as mc

    cdf_rev = np.tile(np.array([1.0, 0.8, 0.4, 0.3, 0.1], np.float32), (4, 1))
    ns = 8192
    k = acc.load("multinomial", "float")
    out = acc.alloc(4 * ns * 4)
    grid, thread = mc.make_kernel_grid_2d(4, ns, k.max_threads_per_threadgroup())
    st, sq = 0x1234567890ABCDEF, 0x0FEDCBA987654321
    mc.KernelTask(k, grid, thread, [L((4, ns)), out, L(cdf_rev.shape), acc.to_device(cdf_rev),
                                    np.uint64(st), np.uint64(sq)])()
    acc.wait()
    got = out.download(np.int32, 4 * ns).reshape(4, ns)
    experiment = np.array([0.2, 0.4, 0.1, 0.2, 0.1])                     # reversed experiment_probs
    for r in range(4):
        freq = np.bincount(got[r], minlength=5) / ns
        assert np.all(np.abs(freq - experiment) <= 0.02), freq
    ref = np.zeros((4, ns), np.int32)
    mo.multinomial(F32, mo.layout(ref.shape), ref, mo.layout(cdf_rev.shape), cdf_rev, st, sq)
    parity.exact(got, ref, "multinomial draws (PCG32 streams)")


@pytest.mark.parametrize("dt", [F32, BF16])
def test_logical_scatter_gather_sub_exact(acc, dt):
    import metalchat_amd as mc

    rng = np.random.default_rng(3)
    shp = (3, 700)
    a = mo.encode(dt, rng.uniform(size=shp).astype(np.float32))
    b = mo.encode(dt, rng.uniform(size=shp).astype(np.float32))
    ab, bb = acc.to_device(a), acc.to_device(b)
    thr = mo.encode(dt, np.array([0.5], np.float32))[0]

    def run(name, out_dtype, args, rows=shp[0], dim=shp[1], out_shape=shp):
        k = acc.load(name)
        grid, thread = mc.make_kernel_grid_2d(rows, dim, k.max_threads_per_threadgroup())
        out = acc.alloc(int(np.prod(out_shape)) * np.dtype(out_dtype).itemsize)
        mc.KernelTask(k, grid, thread, [L(out_shape), out] + args)()
        acc.wait()
        return out, out.download(out_dtype, int(np.prod(out_shape))).reshape(out_shape)

    _, d = run(f"sub_{TN[dt]}", a.dtype, [L(shp), ab, L(shp), bb])
    ref = np.zeros_like(a)
    mo.sub(dt, mo.layout(shp), ref, mo.layout(shp), a, mo.layout(shp), b)
    parity.exact(d, ref, "sub")
    for name, fn in (("gt", mo.gt), ("le", mo.le)):
        mb, m = run(f"{name}_{TN[dt]}", np.uint8, [L(shp), ab, thr])
        rm = np.zeros(shp, np.uint8)
        fn(dt, mo.layout(shp), rm, mo.layout(shp), a, 0.5)
        parity.exact(m, rm, name)
    # scatter writes in place where the mask is set (doc example of include/metalchat/kernel/copy.h:106-118)
    mask = (rng.uniform(size=shp) > 0.7).astype(np.uint8)
    tgt = acc.to_device(a)
    k = acc.load(f"scatter_{TN[dt]}")
    grid, thread = mc.make_kernel_grid_2d(*shp, k.max_threads_per_threadgroup())
    nine = mo.encode(dt, np.array([9.0], np.float32))[0]
    mc.KernelTask(k, grid, thread, [L(shp), tgt, L(shp), acc.to_device(mask), nine])()
    acc.wait()
    ref = a.copy()
    mo.scatter(dt, mo.layout(shp), ref, mo.layout(shp), mask, 9.0)
    parity.exact(tgt.download(a.dtype, a.size).reshape(shp), ref, "scatter")
    # gather along the row
    index = rng.integers(0, shp[1], size=(3, 40)).astype(np.int32)
    for name, src, odt, code in ((f"gather_{TN[dt]}", a, a.dtype, dt),
                                 ("gather_int32_t", rng.integers(0, 1 << 30, size=shp).astype(np.int32), np.int32, 2)):
        _, g = run(name, odt, [L(shp), acc.to_device(src), L(index.shape), acc.to_device(index)],
                   rows=3, dim=40, out_shape=(3, 40))
        ref = np.zeros((3, 40), odt)
        mo.gather(code, mo.layout((3, 40)), ref, mo.layout(shp), src, mo.layout(index.shape), index)
        parity.exact(g, ref, name)
        assert np.array_equal(g, np.take_along_axis(src, index, axis=1))


@pytest.mark.parametrize("dt", [F32, BF16])
def test_div_and_row_sum(acc, dt):
    # div: kernel/arithmetic.metal:124-157; sum: test/test_kernel_sum.cc:43-64 ([4, 64, 4098] rows,
    # within 0.01 of the sequential float sum) -- and both bit-exact against the oracle
    import metalchat_amd as mc

    rng = np.random.default_rng(4)
    shp = (5, 300)
    a = mo.encode(dt, rng.uniform(0.5, 2.0, shp).astype(np.float32))
    b = mo.encode(dt, rng.uniform(0.5, 2.0, shp).astype(np.float32))
    k = acc.load(f"div_{TN[dt]}")
    grid, thread = mc.make_kernel_grid_2d(*shp, k.max_threads_per_threadgroup())
    out = acc.alloc(a.size * a.itemsize)
    mc.KernelTask(k, grid, thread, [L(shp), out, L(shp), acc.to_device(a), L(shp), acc.to_device(b)])()
    acc.wait()
    ref = np.zeros_like(a)
    mo.div(dt, mo.layout(shp), ref, mo.layout(shp), a, mo.layout(shp), b)
    parity.exact(out.download(a.dtype, a.size).reshape(shp), ref, "div")
    rows, dim = 4 * 64, 4098
    x = mo.encode(dt, rng.uniform(size=(rows, dim)).astype(np.float32))
    k = acc.load(f"sum_{TN[dt]}")
    block = -(-dim // k.max_threads_per_threadgroup())
    threads = -(-dim // block)
    so = acc.alloc(rows * x.itemsize)
    mc.KernelTask(k, (threads * rows, 1, 1), (threads, 1, 1), [L((rows,)), so, L(x.shape), acc.to_device(x), np.uint32(block)])()
    acc.wait()
    got = so.download(x.dtype, rows)
    if dt == F32:
        seq = np.zeros(rows, np.float32)
        for r in range(rows):
            acc_ = np.float32(0)
            for v in x[r]:
                acc_ = np.float32(acc_ + v)
            seq[r] = acc_
        assert np.all(np.abs(got - seq) <= 0.01)
    ref = np.zeros(rows, x.dtype)
    mo.row_sum(dt, mo.layout((rows,)), ref, mo.layout(x.shape), x)
    # fp32 summation order differs (64-lane shuffle tree vs the oracle's 32-lane butterfly)
    parity.check(dt, got, ref, rel=1e-5 if dt == F32 else 1e-2, max_ulp=1, max_frac=0.1, scale_aware=False, what="sum")


# ------------------------------------------------------------------------------------------ fused
SP = np.dtype([("k", np.uint32), ("ncand", np.uint32), ("cap", np.uint32), ("inv_temp", np.float32),
               ("top_p", np.float32), ("nlists", np.uint32), ("kpad", np.uint32)])


def fused_sample(acc, dt, logits, top_k=50, temperature=0.6, top_p=0.9, seed=(0, 0), cap=4096):
    """The two launches of decoder.cc run_head() with its geometry (sampler_chunk): 512-logit chunks, one wave and one sorted list of
    kpad keys per chunk, then one workgroup of two waves."""
    import metalchat_amd as mc

    n = logits.size
    k = min(top_k, n)
    kpad = ceil_pow2(top_k)
    chunk = max(512, kpad)
    while chunk < 2048 and -(-n // chunk) > 1024:
        chunk *= 2
    lists = -(-n // chunk)
    cand = acc.alloc(lists * kpad * 8)
    lb = acc.to_device(logits)
    mc.KernelTask(acc.load("mc_topk_candidates", TN[dt]), (lists * 64, 1, 1), (64, 1, 1),
                  [lb, np.uint32(n), np.uint32(kpad), cand, np.uint32(chunk)])()
    rt = (lambda v: float(mo.decode(dt, mo.encode(dt, np.array([v], np.float32)))[0]))
    p = np.zeros(1, SP)
    p["k"], p["ncand"], p["cap"], p["nlists"], p["kpad"] = k, lists * kpad, cap, lists, kpad
    p["inv_temp"], p["top_p"] = rt(1.0 / rt(temperature)), rt(top_p)
    state = acc.to_device(np.zeros(8, np.int32))
    toks = acc.to_device(np.full(4, -1, np.int32))
    taps = acc.alloc(7 * k * 4)
    seeds = acc.to_device(np.array(seed, np.uint64))
    mc.KernelTask(acc.load("mc_sample", TN[dt]), (128, 1, 1), (128, 1, 1),
                  [cand, p, seeds, np.uint32(1), state, toks, taps], lds_bytes=cap * 8)()
    acc.wait()
    return int(toks.download(np.int32, 4)[0]), taps.download(np.float32, 7 * k).reshape(7, k)


# ("flat": one value everywhere -- with top_k 128 more keys tie at the bound than the second launch holds: its bisection path;
#  "coarse" at top_k 128 on the big row: long runs of equal logits)
CASES = [(F32, 128256, 50, "normal"), (BF16, 128256, 50, "normal"), (BF16, 128256, 50, "coarse"),
         (F32, 40, 50, "normal"), (BF16, 5000, 128, "coarse"), (F32, 256000, 64, "peaked"), (BF16, 2048, 1, "normal"),
         (BF16, 128256, 128, "flat"), (F32, 128256, 128, "coarse"), (BF16, 700000, 50, "normal")]


@pytest.mark.parametrize("dt,vocab,top_k,kind", CASES)
def test_fused_default_sampler_matches_oracle_chain(acc, dt, vocab, top_k, kind):
    rng = np.random.default_rng(vocab + top_k)
    x = rng.normal(0, 2.0, vocab).astype(np.float32)
    if kind == "coarse":
        x = np.round(x * 2) / 2            # many exactly equal logits: exercises the tie rule
    if kind == "peaked":
        x[rng.integers(0, vocab, 3)] += 12.0
    if kind == "flat":
        x[:] = 0.5
    logits = mo.encode(dt, x)
    for seed in ((0, 0), (123456789, 987654321)):
        tok, taps = fused_sample(acc, dt, logits, top_k=top_k, seed=seed)
        otok, otaps = mo.sample_default(dt, logits, top_k=top_k, init_state=seed[0], init_seq=seed[1], taps=True)
        names = ["scaled", "probs", "sorted", "cumsum", "diff", "masked", "ids"]
        for r in range(7):
            parity.exact(taps[r], otaps[r], f"sampler tap '{names[r]}'")
        assert tok == otok
        # nn/sampling.h + kernel/multinomial.metal:112 with sample_size 1: the draw interval is
        # empty, so the chain returns the head of the sorted nucleus
        assert tok == int(otaps[6][0])


@pytest.mark.parametrize("dt,vocab,top_k,kind", [(BF16, 128256, 50, "coarse"), (F32, 128256, 50, "normal")])
def test_fused_default_sampler_bisection_path(acc, dt, vocab, top_k, kind):
    """The second launch with room for only 2 * kpad keys: more keys lie above the bound than it holds, so the k-th largest key
    is found by bisection over the key bits -- the same tokens and taps as the usual path."""
    rng = np.random.default_rng(vocab + top_k)
    x = rng.normal(0, 2.0, vocab).astype(np.float32)
    if kind == "coarse":
        x = np.round(x * 2) / 2
    logits = mo.encode(dt, x)
    tok, taps = fused_sample(acc, dt, logits, top_k=top_k, seed=(5, 6), cap=128)
    otok, otaps = mo.sample_default(dt, logits, top_k=top_k, init_state=5, init_seq=6, taps=True)
    for r in range(7):
        parity.exact(taps[r], otaps[r], f"sampler tap {r} (bisection path)")
    assert tok == otok


def test_sampler_argument_errors(acc):
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(F32, n_layers=1)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg))
    for args, msg in (((mc.SAMPLER_DEFAULT, 50, 0.0, 0.9), "temperature must be positive"),
                      ((mc.SAMPLER_DEFAULT, 50, 0.6, 1.5), r"probability must be in \[0.0, 1.0\]"),
                      ((mc.SAMPLER_DEFAULT, 500, 0.6, 0.9), "1..128"), ((7, 50, 0.6, 0.9), "unknown sampler")):
        with pytest.raises(mc.McError, match=msg) as e:
            dec.set_sampler(*args)
        assert e.value.status == 1
    dec.release()


@pytest.mark.parametrize("dt", [F32, BF16])
def test_decoder_with_default_sampler(acc, dt):
    import metalchat_amd as mc

    cfg = mg.tiny_cfg(dt, max_seq_len=32)
    w = mg.make_model(cfg, seed=61, quant="i4", group=32)
    dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=2, group_size=32))
    dec.load_model(w)
    dec.set_taps(True)
    dec.set_sampler(mc.SAMPLER_DEFAULT, 50, 0.6, 0.9)
    dec.set_seeds([(11, 22)])
    tok, stepped = 3, []
    for pos in range(8):
        got = dec.step(tok, pos)
        otok, otaps = mo.sample_default(dt, dec.logits(), init_state=11, init_seq=22, taps=True)
        assert got == otok
        parity.exact(dec.sampler_taps(), otaps, f"pos {pos} sampler taps")
        stepped.append(got)
        tok = got
    # chained on the device (and through the captured graph): same tokens
    for graph in (0, 1):
        d2 = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=2, group_size=32, use_graph=graph))
        d2.load_model(w)
        d2.set_sampler(mc.SAMPLER_DEFAULT, 50, 0.6, 0.9)
        d2.set_seeds([(11, 22), (33, 44), (55, 66)])
        assert list(d2.generate(3, 0, 8)) == stepped
        d2.set_sampler(mc.SAMPLER_GREEDY)
        d2.release()
    dec.release()
