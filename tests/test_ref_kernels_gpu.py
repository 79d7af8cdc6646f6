"""Parity of the reference-named kernels (metalchat_amd/csrc/kernels/ref_kernels.hip) driven
through the encoder seam exactly as the reference's wrappers drive Metal: kernel looked up by its
mangled host name, tensor = tensor_layout<N> by value then buffer, scalars by value, dispatchThreads
geometry from make_kernel_grid_2d (include/metalchat/kernel/*.h, src/kernel.cc:13-37).

Each test restates one reference kernel test with seeded inputs and checks
the GPU result against the CPU oracle: bit-exact for copy / roll / embedding, tolerance for fp."""
import numpy as np
import pytest

import parity
from oracle import mc_oracle as mo

pytestmark = pytest.mark.gpu
BF16, F32 = 0, 1
TN = {BF16: "bfloat", F32: "float"}


def L(*a, **k):
    import metalchat_amd as mc

    return mc.layout(*a, **k)


def rnd(seed, shape, dt, lo=0.0, hi=1.0):
    x = np.random.default_rng(seed).uniform(lo, hi, size=shape).astype(np.float32)
    return mo.encode(dt, x)


def run2d(acc, name, dt, out_shape, args, rows, dim):
    """Launch a 2-D elementwise kernel with make_kernel_grid_2d geometry; args excludes output."""
    import metalchat_amd as mc

    k = acc.load(name)
    grid, thread = mc.make_kernel_grid_2d(rows, dim, k.max_threads_per_threadgroup())
    out = acc.alloc(int(np.prod(out_shape)) * (2 if dt == BF16 else 4))
    mc.KernelTask(k, grid, thread, [L(out_shape), out] + args)()
    acc.wait()
    return out.download(mo.np_dtype(dt), int(np.prod(out_shape))).reshape(out_shape)


@pytest.mark.parametrize("dt", [F32, BF16])
def test_hadamard_add_scalar_mul(acc, dt):
    # test/test_kernel_mul.cc:16-38,68-93 ; test/test_kernel_arithmetic.cc:18-42
    shp = (15, 8192)
    a, b = rnd(1, shp, dt), rnd(2, shp, dt)
    ab, bb = acc.to_device(a), acc.to_device(b)
    for name, ofn in (("hadamard", mo.hadamard), ("add", mo.add)):
        got = run2d(acc, f"{name}_{TN[dt]}", dt, shp, [L(shp), ab, L(shp), bb], *shp)
        ref = np.zeros_like(a)
        ofn(dt, mo.layout(shp), ref, mo.layout(shp), a, mo.layout(shp), b)
        parity.exact(got, ref, name)
    m = mo.encode(dt, np.array([8.0], np.float32))
    got = run2d(acc, f"scalar_mul_{TN[dt]}", dt, shp, [L(shp), ab, m[0]], *shp)
    ref = np.zeros_like(a)
    mo.scalar_mul(dt, mo.layout(shp), ref, mo.layout(shp), a, 8.0)
    parity.exact(got, ref, "scalar_mul")


@pytest.mark.parametrize("odt,sdt", [(F32, F32), (BF16, F32), (BF16, BF16), (F32, BF16)])
def test_hadamard_broadcast_dequantizer(acc, odt, sdt):
    # test/test_kernel_mul.cc:41-65: int8 [512,64,32] x scales [512,64,1]
    r = np.random.default_rng(3)
    w = r.integers(-128, 128, size=(512 * 64, 32), dtype=np.int8)
    s = mo.encode(sdt, r.random(512 * 64, dtype=np.float32))
    shp = w.shape
    got = run2d(acc, f"hadamard_broadcast_{TN[odt]}_int8_t_{TN[sdt]}", odt, shp,
                [L(shp), acc.to_device(w), L((shp[0],)), acc.to_device(s)], *shp)
    ref = np.zeros(shp, mo.np_dtype(odt))
    mo.hadamard_broadcast(odt, sdt, mo.layout(shp), ref, mo.layout(shp), w, mo.layout((shp[0],)), s)
    parity.exact(got, ref, "hadamard_broadcast")


@pytest.mark.parametrize("dt", [F32, BF16])
def test_bmm_linear_shape(acc, dt):
    # test/test_kernel_bmm.cc:33-61: [1,5,2048] x ([1024,2048])^T, strided B = transposed weight
    import metalchat_amd as mc

    a, w = rnd(4, (1, 5, 2048), dt, -1, 1), rnd(5, (1024, 2048), dt, -1, 1)
    k = acc.load("bmm", 8, TN[dt])
    out = acc.alloc(5 * 1024 * (2 if dt == BF16 else 4))
    bl = L((1, 2048, 1024), strides=(2048 * 1024, 1, 2048))
    mc.KernelTask(k, (8, 1024, 1), (8, 8, 1),
                  [L((1, 5, 1024)), out, L((1, 5, 2048)), acc.to_device(a), bl, acc.to_device(w)])()
    acc.wait()
    got = out.download(mo.np_dtype(dt), 5 * 1024)
    ref = np.zeros((1, 5, 1024), mo.np_dtype(dt))
    mo.bmm(dt, mo.layout((1, 5, 1024)), ref, mo.layout((1, 5, 2048)), a,
           mo.layout((1, 2048, 1024), strides=(2048 * 1024, 1, 2048)), w)
    # same k order as the reference (sequential); fp32 fma contraction may differ from the host
    parity.check(dt, got, ref.reshape(-1), rel=1e-5 if dt == F32 else 1e-3, max_ulp=1, max_frac=0.01,
                 scale_aware=False, what="bmm")


@pytest.mark.parametrize("dt", [F32, BF16])
def test_rmsnorm_and_softmax(acc, dt):
    # test/test_kernel_rmsnorm.cc:18-70 ; test/test_kernel_softmax.cc:19-72
    import metalchat_amd as mc

    rows, dim = 15, 2048
    x, w = rnd(6, (rows, dim), dt), rnd(7, (dim,), dt)
    k = acc.load("rmsnorm", TN[dt])
    mt = k.max_threads_per_threadgroup()
    block = (dim + mt - 1) // mt
    threads = (dim + block - 1) // block
    out = acc.alloc(rows * dim * (2 if dt == BF16 else 4))
    mc.KernelTask(k, (threads * rows, 1, 1), (threads, 1, 1),
                  [L((rows, dim)), out, L((rows, dim)), acc.to_device(x), L((dim,)), acc.to_device(w),
                   np.float32(1e-5), np.float32(0.0), np.uint32(block)])()
    acc.wait()
    got = out.download(mo.np_dtype(dt), rows * dim)
    ref = np.zeros((rows, dim), mo.np_dtype(dt))
    mo.rmsnorm(dt, mo.layout((rows, dim)), ref, mo.layout((rows, dim)), x, mo.layout((dim,)), w, 1e-5, 0.0)
    parity.check(dt, got, ref.reshape(-1), rel=1e-5 if dt == F32 else 1e-3, max_ulp=1, max_frac=0.01,
                 scale_aware=False, what="rmsnorm")
    # rmsnorm of ones with w = 3: exactly 3 in bf16 (test_kernel_rmsnorm.cc:18-37); in f32 the
    # eps shows: 3 / sqrt(1 + 1e-5)
    ones, w3 = mo.encode(dt, np.ones((60, 7), np.float32)), mo.encode(dt, np.full(7, 3.0, np.float32))
    out = acc.alloc(60 * 7 * 4)
    mc.KernelTask(k, (7 * 60, 1, 1), (7, 1, 1),
                  [L((60, 7)), out, L((60, 7)), acc.to_device(ones), L((7,)), acc.to_device(w3),
                   np.float32(1e-5), np.float32(0.0), np.uint32(1)])()
    acc.wait()
    got3 = mo.decode(dt, out.download(mo.np_dtype(dt), 420))
    assert np.all(got3 == (3.0 if dt == BF16 else np.float32(3.0) * (np.float32(1.0) / np.sqrt(np.float32(1.0) + np.float32(1e-5)))))

    k = acc.load("softmax", TN[dt])
    rows, dim = 32, 2048
    x = rnd(8, (rows, dim), dt, -3, 3)
    block = (dim + mt - 1) // mt
    threads = (dim + block - 1) // block
    out = acc.alloc(rows * dim * 4)
    mc.KernelTask(k, (threads * rows, 1, 1), (threads, 1, 1),
                  [L((rows, dim)), out, L((rows, dim)), acc.to_device(x), np.uint32(block)])()
    acc.wait()
    got = out.download(mo.np_dtype(dt), rows * dim)
    ref = np.zeros((rows, dim), mo.np_dtype(dt))
    mo.softmax(dt, mo.layout((rows, dim)), ref, mo.layout((rows, dim)), x)
    parity.check(dt, got, ref.reshape(-1), rel=1e-5 if dt == F32 else 1e-3, max_ulp=1, max_frac=0.01,
                 scale_aware=False, what="softmax")


def test_softmax_bf16_known_answer(acc):
    # test/test_kernel_softmax.cc:19-39
    import metalchat_amd as mc

    x = mo.to_bf16(np.arange(5, dtype=np.float32)).reshape(1, 5)
    k = acc.load("softmax", "bfloat")
    out = acc.alloc(16)
    mc.KernelTask(k, (5, 1, 1), (5, 1, 1), [L((1, 5)), out, L((1, 5)), acc.to_device(x), np.uint32(1)])()
    acc.wait()
    got = mo.from_bf16(out.download(np.uint16, 5))
    expect = mo.round_bf16(np.array([0.0116577, 0.0317383, 0.0859375, 0.234375, 0.636719], np.float32))
    np.testing.assert_allclose(got, expect, atol=1e-5, rtol=0)


@pytest.mark.parametrize("dt", [F32, BF16])
def test_rope_and_rope_freqs(acc, dt):
    # rope_freqs: test/test_kernel_embedding.cc:72-137; rope rotation: no reference test (unpinned)
    import metalchat_amd as mc

    dim, seq, theta, start = 64, 1024, 500000.0, 100
    k = acc.load("rope_freqs", "float")
    cb, sb = acc.alloc(seq * 32 * 4), acc.alloc(seq * 32 * 4)
    grid, thread = mc.make_kernel_grid_2d(seq, 32, k.max_threads_per_threadgroup())
    mc.KernelTask(k, grid, thread, [L((seq, 32)), cb, L((seq, 32)), sb, np.uint32(dim), np.uint32(start),
                                    np.float32(theta)])()
    acc.wait()
    c, s = np.zeros((seq, 32), np.float32), np.zeros((seq, 32), np.float32)
    mo.rope_freqs(mo.layout((seq, 32)), c, mo.layout((seq, 32)), s, dim, start, theta)
    parity.exact(cb.download(np.float32, seq * 32).reshape(seq, 32), c, "rope_freqs cos")
    parity.exact(sb.download(np.float32, seq * 32).reshape(seq, 32), s, "rope_freqs sin")
    # rotation of [bs=1, len=3, heads=8, hd=64] at start_pos 5 (table row = start_pos + pos)
    n_head, ln = 8, 3
    x = rnd(9, (ln * n_head, dim), dt, -1, 1)
    k = acc.load("rope", TN[dt])
    grid, thread = mc.make_kernel_grid_2d(ln * n_head, dim, k.max_threads_per_threadgroup())
    out = acc.alloc(x.size * 4)
    mc.KernelTask(k, grid, thread, [L(x.shape), out, L(x.shape), acc.to_device(x), L((seq, 32)), cb,
                                    L((seq, 32)), sb, np.uint32(1), np.uint32(n_head), np.uint32(5)])()
    acc.wait()
    ref = np.zeros_like(x)
    mo.rope(dt, mo.layout(x.shape), ref, mo.layout(x.shape), x, mo.layout((seq, 32)), c,
            mo.layout((seq, 32)), s, 1, n_head, 5)
    parity.check(dt, out.download(mo.np_dtype(dt), x.size), ref.reshape(-1),
                 rel=1e-6 if dt == F32 else 1e-3, max_ulp=1, max_frac=0.01, scale_aware=False, what="rope")


@pytest.mark.parametrize("dt", [F32, BF16])
def test_embedding_copy_roll_exact(acc, dt):
    # test/test_kernel_embedding.cc:19-53 ; test_kernel_copy.cc:14-70 ; test_kernel_roll.cc:16-71
    import metalchat_amd as mc

    tb = 2 if dt == BF16 else 4
    w = rnd(10, (1000, 256), dt)
    ids = np.array([[0, 1, 2, 3], [2, 4, 1, 0], [4, 3, 3, 999]], np.int32)
    k = acc.load("embedding", TN[dt])
    out = acc.alloc(3 * 4 * 256 * tb)
    # geometry of include/metalchat/kernel/embedding.h:41-63 for dim_size 4, emb 256, 1024 threads
    mc.KernelTask(k, (1, 256, 3), (1, 256, 1),
                  [L((3, 4, 256)), out, L((3, 4)), acc.to_device(ids), L((1000, 256)), acc.to_device(w),
                   np.uint32(4)])()
    acc.wait()
    parity.exact(out.download(mo.np_dtype(dt), 3 * 4 * 256).reshape(3, 4, 256), w[ids], "embedding")

    # copy into a narrowed slice: [48,64] -> narrow(dim 3, offset 2) of zeros [1,6,8,4,64]
    src = rnd(11, (48, 64), dt)
    dst = acc.to_device(np.zeros(6 * 8 * 4 * 64, mo.np_dtype(dt)))
    k = acc.load("copy", TN[dt])
    grid, thread = mc.make_kernel_grid_2d(48, 64, k.max_threads_per_threadgroup())
    mc.KernelTask(k, grid, thread, [L((48, 64), strides=(256, 1), offsets=(128, 0)), dst, L((48, 64)),
                                    acc.to_device(src)])()
    acc.wait()
    got = dst.download(mo.np_dtype(dt), 6 * 8 * 4 * 64).reshape(48, 4, 64)
    parity.exact(got[:, 2, :], src, "copy into slice")
    assert np.all(got[:, [0, 1, 3], :] == 0)

    # roll left by one along dim 1 of [2,128,8,64]
    x = rnd(12, (2, 128, 8, 64), dt)
    n = x.size
    k = acc.load("roll", TN[dt])
    out = acc.alloc(n * tb)
    mt = k.max_threads_per_threadgroup()
    mc.KernelTask(k, ((n + mt - 1) // mt * mt, 1, 1), (mt, 1, 1),
                  [L((n,)), out, L((n,)), acc.to_device(x), np.uint32(1), np.uint32(128), np.uint32(8 * 64)])()
    acc.wait()
    parity.exact(out.download(mo.np_dtype(dt), n).reshape(x.shape), np.roll(x, -1, axis=1), "roll")


@pytest.mark.parametrize("dt", [F32, BF16])
def test_activations_and_add_broadcast(acc, dt):
    # test/test_kernel_activation.cc:19-83 ; test/test_kernel_arithmetic.cc:122-149
    shp = (15, 8192)
    x = rnd(13, shp, dt, -4, 4)
    xb = acc.to_device(x)
    for name, ofn in (("silu", mo.silu), ("gelu", mo.gelu)):
        got = run2d(acc, f"{name}_{TN[dt]}", dt, shp, [L(shp), xb], *shp)
        ref = np.zeros_like(x)
        ofn(dt, mo.layout(shp), ref, mo.layout(shp), x)
        parity.exact(got, ref, name)
    a, m = rnd(14, (40, 1600), dt), rnd(15, (1600,), dt)
    got = run2d(acc, f"add_broadcast_{TN[dt]}", dt, a.shape,
                [L(a.shape), acc.to_device(a), L(m.shape), acc.to_device(m)], *a.shape)
    ref = np.zeros_like(a)
    mo.add_broadcast(dt, mo.layout(a.shape), ref, mo.layout(a.shape), a, mo.layout(m.shape), m)
    parity.exact(got, ref, "add_broadcast")


def test_command_ordering_add_chain(acc):
    # test/test_kernel_thread.cc:16-40: three chained adds on ones == 8, no explicit barriers
    x = np.ones((12, 15), np.float32)
    buf = acc.to_device(x)
    import metalchat_amd as mc

    k = acc.load("add", "float")
    grid, thread = mc.make_kernel_grid_2d(12, 15, k.max_threads_per_threadgroup())
    for _ in range(3):
        out = acc.alloc(x.nbytes)
        mc.KernelTask(k, grid, thread, [L(x.shape), out, L(x.shape), buf, L(x.shape), buf])()
        buf = out
    acc.wait()
    assert np.all(buf.download(np.float32, x.size) == 8.0)


def test_encoder_validation_and_errors(acc):
    # include/metalchat/kernel.h:119-141 ; src/accelerator.cc:125-129 ; test/test_accelerator.cc:15-21
    import metalchat_amd as mc

    with pytest.raises(mc.McError) as e:
        acc.load("no_such_kernel", "float")
    assert e.value.status == 1 and "not found in a shader library" in str(e.value)
    with pytest.raises(mc.McError) as e:
        mc.HardwareAccelerator(path="some/nonexisting/file")
    assert e.value.status == 2 and "library not found" in str(e.value)
    k = acc.load("add", "float")
    with pytest.raises(mc.McError) as e:
        mc.KernelTask(k, (4096, 1, 1), (2048, 1, 1), [])()
    assert e.value.status == 1 and "exceeds maximum number of threads" in str(e.value)
    with pytest.raises(mc.McError) as e:
        mc.KernelTask(k, (32, 1, 1), (64, 1, 1), [])()
    assert e.value.status == 1 and "less threads in grid" in str(e.value)
