"""Pins the CPU oracle against every known-answer / property check the reference's own kernel
tests hold for the decode hot path.  Each test cites the reference test it
restates; inputs are seeded instead of std::random_device.  CPU only."""
import numpy as np
import pytest

from oracle import mc_oracle as mo

BF16, F32 = mo.BF16, mo.F32
L = mo.layout


def rng(seed):
    return np.random.default_rng(seed)


def test_bf16_roundtrip_rne():
    # include/metalchat/dtype.h:32-58
    x = np.array([1.0, 1.00390625, 1.01171875, 3.0, -2.5, 0.0, 65280.0, 1e-3], dtype=np.float32)
    b = mo.to_bf16(x)
    for xi, bi in zip(x, b):
        assert mo.lib().mco_f32_to_bf16(float(xi)) == int(bi)
    # ties to even: 1 + 2^-8 is exactly between 1.0 and 1 + 2^-7 -> 1.0
    assert mo.from_bf16(mo.to_bf16(np.array([1.00390625], np.float32)))[0] == 1.0
    # 1 + 3*2^-8 ties up to 1 + 2^-6
    assert mo.from_bf16(mo.to_bf16(np.array([1.01171875], np.float32)))[0] == 1.015625


def test_softmax_predefined_array_bf16():
    # test/test_kernel_softmax.cc:19-39 -- the only hard numeric KAT in the reference
    x = mo.to_bf16(np.arange(5, dtype=np.float32)).reshape(1, 5)
    out = np.zeros_like(x)
    mo.softmax(BF16, L((1, 5)), out, L((1, 5)), x)
    expect = mo.from_bf16(mo.to_bf16(np.array(
        [0.0116577, 0.0317383, 0.0859375, 0.234375, 0.636719], np.float32)))
    np.testing.assert_allclose(mo.from_bf16(out)[0], expect, atol=1e-5, rtol=0)


def test_softmax_rows_sum_to_one():
    # test/test_kernel_softmax.cc:42-72
    x = rng(1).random((1 * 32 * 4, 4), dtype=np.float32)
    out = np.zeros_like(x)
    mo.softmax(F32, L(x.shape), out, L(x.shape), x)
    np.testing.assert_allclose(out.sum(axis=1), 1.0, atol=1e-5)
    x = rng(2).random((1, 30), dtype=np.float32)
    out = np.zeros_like(x)
    mo.softmax(F32, L(x.shape), out, L(x.shape), x)
    assert abs(out.sum() - 1.0) < 0.01


def test_rmsnorm_ones_bf16_exact():
    # test/test_kernel_rmsnorm.cc:18-37: ones[4,3,5,7] with w = 3 -> exactly 3.0
    x = mo.to_bf16(np.ones((4 * 3 * 5, 7), np.float32))
    w = mo.to_bf16(np.full((7,), 3.0, np.float32))
    out = np.zeros_like(x)
    mo.rmsnorm(BF16, L(x.shape), out, L(x.shape), x, L((7,)), w, 1e-5, 0.0)
    assert np.all(mo.from_bf16(out) == 3.0)


def test_rmsnorm_random_f32():
    # test/test_kernel_rmsnorm.cc:40-70
    r = rng(3)
    x = r.random((15, 2048), dtype=np.float32)
    w = r.random((2048,), dtype=np.float32)
    out = np.zeros_like(x)
    mo.rmsnorm(F32, L(x.shape), out, L(x.shape), x, L(w.shape), w, 1e-5, 0.0)
    inv = 1.0 / np.sqrt((x.astype(np.float64) ** 2).sum(1) / 2048 + 1e-5)
    np.testing.assert_allclose(out, w[None] * x * inv[:, None], atol=1e-5)


def test_hadamard_and_scalar_mul_f32():
    # test/test_kernel_mul.cc:16-38,68-93
    r = rng(4)
    a = r.random((15, 8192), dtype=np.float32)
    b = r.random((15, 8192), dtype=np.float32)
    out = np.zeros_like(a)
    mo.hadamard(F32, L(a.shape), out, L(a.shape), a, L(b.shape), b)
    np.testing.assert_allclose(out, a * b, atol=1e-5)
    x = r.random((32 * 4, 64), dtype=np.float32)
    out = np.zeros_like(x)
    mo.scalar_mul(F32, L(x.shape), out, L(x.shape), x, 8.0)
    np.testing.assert_allclose(out, x * 8.0, atol=1e-5)


def test_hadamard_broadcast_dequant():
    # test/test_kernel_mul.cc:41-65: int8 [512,64,32] in [1,10], f32 scales [512,64,1]
    r = rng(5)
    w = r.integers(1, 11, size=(512 * 64, 32), dtype=np.int8)
    s = r.random((512 * 64,), dtype=np.float32)
    out = np.zeros(w.shape, np.float32)
    mo.hadamard_broadcast(F32, F32, L(w.shape), out, L(w.shape), w, L(s.shape), s)
    np.testing.assert_allclose(out, w.astype(np.float32) * s[:, None], atol=1e-5)
    # bf16 output: product evaluated in bf16 (mul.metal:78-82)
    outb = np.zeros(w.shape, np.uint16)
    mo.hadamard_broadcast(BF16, F32, L(w.shape), outb, L(w.shape), w, L(s.shape), s)
    exp = mo.round_bf16(w.astype(np.float32) * mo.round_bf16(s)[:, None])
    assert np.array_equal(mo.from_bf16(outb), exp)


def test_bmm_against_triple_loop():
    # test/test_kernel_bmm.cc:33-61 (shape reduced: [1,5,256] x [512,256]^T), abs 1e-4
    r = rng(6)
    a = r.random((1, 5, 256), dtype=np.float32)
    w = r.random((512, 256), dtype=np.float32)
    out = np.zeros((1, 5, 512), np.float32)
    # weight.transpose({1,0}).expand_dims(0): sizes [1,K,N], strides [K*N?,1,K]
    bl = L((1, 256, 512), strides=(256 * 512, 1, 256))
    mo.bmm(F32, L(out.shape), out, L(a.shape), a, bl, w)
    np.testing.assert_allclose(out[0], a[0].astype(np.float64) @ w.T.astype(np.float64), atol=1e-4)


def test_rope_freqs_formula():
    # test/test_kernel_embedding.cc:72-137: dim 64, seq 1024, theta 5e5, start 100, abs 1e-4
    dim, seq, theta, start = 64, 1024, 500000.0, 100
    c = np.zeros((seq, dim // 2), np.float32)
    s = np.zeros((seq, dim // 2), np.float32)
    mo.rope_freqs(L(c.shape), c, L(s.shape), s, dim, start, theta)
    expo = (2.0 * np.arange(dim // 2) / dim).astype(np.float32)
    freqs = (np.float32(1.0) / np.power(np.float32(theta), expo)).astype(np.float32)
    ang = (np.arange(start, start + seq, dtype=np.float32)[:, None] * freqs[None]).astype(np.float32)
    # the reference checks abs 1e-4; a 1-ulp difference in `freq` between two float pow()
    # implementations moves cos/sin by |angle| * 2^-23, so the bound carries that term
    tol = 1e-4 + np.abs(ang) * 2.0 ** -23
    assert np.all(np.abs(c - np.cos(ang.astype(np.float64))) <= tol)
    assert np.all(np.abs(s - np.sin(ang.astype(np.float64))) <= tol)


def test_embedding_exact():
    # test/test_kernel_embedding.cc:19-53 (table reduced to 1000 x 256)
    r = rng(7)
    w = r.random((1000, 256), dtype=np.float32)
    ids = np.array([[0, 1, 2, 3], [2, 4, 1, 0], [4, 3, 3, 2]], np.int32)
    out = np.zeros((3, 4, 256), np.float32)
    mo.embedding(F32, L(out.shape), out, L(ids.shape), ids, L(w.shape), w)
    assert np.array_equal(out, w[ids])


def test_copy_exact_and_into_slice():
    # test/test_kernel_copy.cc:14-70
    r = rng(8)
    x = r.random((16, 4096), dtype=np.float32)
    out = np.zeros_like(x)
    mo.copy(F32, L(x.shape), out, L(x.shape), x)
    assert np.array_equal(out, x)
    # copy [1,6,8,1,64] into narrow(dim=3, offset=2, len=1) of zeros [1,6,8,4,64]
    src = r.random((48, 64), dtype=np.float32)
    dst = np.zeros((1, 6, 8, 4, 64), np.float32)
    # target.view({-1, 64}): rows of 64 with row stride 4*64, narrow offset 2*64 folded in offset[0]
    tl = L((48, 64), strides=(256, 1), offsets=(128, 0))
    mo.copy(F32, tl, dst, L(src.shape), src)
    assert np.array_equal(dst[0, :, :, 2, :].reshape(48, 64), src)
    assert np.all(dst[0, :, :, [0, 1, 3], :] == 0)


def test_roll_left_by_one():
    # test/test_kernel_roll.cc:16-71: out[b,s] == in[b,(s+1)%n]
    r = rng(9)
    for shape in [(2, 4, 5), (2, 128, 8, 64)]:
        x = r.random(shape, dtype=np.float32)
        out = np.zeros_like(x)
        n = x.size
        stride = int(np.prod(shape[2:]))
        mo.roll(F32, L((n,)), out, L((n,)), x, 1, shape[1], stride)
        assert np.array_equal(out, np.roll(x, -1, axis=1))


def test_add_chain_equals_eight():
    # test/test_kernel_thread.cc:16-40 and test/test_kernel_arithmetic.cc:18-42
    x = np.ones((12, 15), np.float32)
    for _ in range(3):
        out = np.zeros_like(x)
        mo.add(F32, L(x.shape), out, L(x.shape), x, L(x.shape), x)
        x = out
    assert np.all(x == 8.0)


def test_add_broadcast_mask():
    # test/test_kernel_arithmetic.cc:122-149 (shape reduced): [5*8*40, 40] + flattened [40*40]
    r = rng(10)
    a = r.random((5 * 8 * 40, 40), dtype=np.float32)
    m = r.random((40 * 40,), dtype=np.float32)
    out = np.zeros_like(a)
    # in2[j % n] with j the column index: the reference flattens scores to [.., 40*40]
    a2 = a.reshape(5 * 8, 1600)
    out2 = out.reshape(5 * 8, 1600)
    mo.add_broadcast(F32, L(a2.shape), out2, L(a2.shape), a2, L(m.shape), m)
    np.testing.assert_allclose(out2, a2 + m[None], atol=1e-5)


def test_silu_gelu():
    # test/test_kernel_activation.cc:19-83
    r = rng(11)
    x = r.random((15, 8192), dtype=np.float32)
    out = np.zeros_like(x)
    mo.silu(F32, L(x.shape), out, L(x.shape), x)
    np.testing.assert_allclose(out, x / (1 + np.exp(-x)), atol=1e-5)
    mo.gelu(F32, L(x.shape), out, L(x.shape), x)
    xd = x.astype(np.float64)
    ref = xd * 0.5 * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (xd + 0.044715 * xd ** 3)))
    np.testing.assert_allclose(out, ref, atol=1e-5)
    # GELU(12) in bf16 stays 12 (no NaN)
    xb = mo.to_bf16(np.full((1, 10), 12.0, np.float32))
    ob = np.zeros_like(xb)
    mo.gelu(BF16, L(xb.shape), ob, L(xb.shape), xb)
    np.testing.assert_allclose(mo.from_bf16(ob), 12.0, atol=1e-5)
