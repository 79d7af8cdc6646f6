"""numpy port of metalchat_amd/csrc/kernels/synth.h (the counter-based synthetic weights bench.py
runs on): regenerates whole matrices on the host so that the CPU oracle can be given EXACTLY the
model mc_decoder_init_synthetic put into HBM -- end-to-end parity at the benchmark's own widths.
Checked element for element against the C ABI's mc_synth_* in tests/test_full_size_gpu.py."""
import numpy as np

U64 = np.uint64


def _mix(z):
    z = (z ^ (z >> U64(30))) * U64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> U64(27))) * U64(0x94D049BB133111EB)
    return z ^ (z >> U64(31))


def _key(seed, matrix_id, a, b):
    with np.errstate(over="ignore"):
        base = _mix(U64(seed) + U64(0x9E3779B97F4A7C15) * U64(matrix_id + 1))
        return _mix(base + ((a.astype(U64) << U64(32)) | b.astype(U64)))


def weights(seed, matrix_id, rows, cols, bits):
    """int8 [rows, cols]"""
    r = np.arange(rows, dtype=U64)[:, None]
    c = np.arange(cols, dtype=U64)[None, :]
    out = np.empty((rows, cols), np.int8)
    step = max(1, (1 << 24) // cols)
    for r0 in range(0, rows, step):   # bounded temporaries
        h = _key(seed, matrix_id, np.broadcast_to(r[r0:r0 + step], (min(step, rows - r0), cols)),
                 np.broadcast_to(c, (min(step, rows - r0), cols)))
        if bits == 4:
            q = ((h >> U64(40)) & U64(15)).astype(np.int16) - 8
            q[q == -8] = 0
        else:
            q = ((h >> U64(40)) & U64(255)).astype(np.int16) - 128
            q[q == -128] = 0
        out[r0:r0 + step] = q.astype(np.int8)
    return out


def scales(seed, matrix_id, rows, groups, in_features, bits):
    """float32 [rows, groups]"""
    r = np.broadcast_to(np.arange(rows, dtype=U64)[:, None], (rows, groups))
    g = np.broadcast_to(np.arange(groups, dtype=U64)[None, :], (rows, groups))
    h = _key(seed ^ 0x5CA1E5, matrix_id, r, g)
    u = ((h >> U64(40)) & U64(0xFFFFFF)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    denom = np.float32(np.sqrt(np.float32(in_features))) * np.float32(8.0 if bits == 4 else 128.0)
    return ((np.float32(0.5) + u) / denom).astype(np.float32)


def values(seed, matrix_id, n, kind, count=1):
    """float32 [n] before rounding to T (kind 0: norm weights, 1: embedding, 2: plain T weights)"""
    i = np.arange(n, dtype=U64)
    h = _key(seed ^ 0xA11CE, matrix_id, i, np.full(n, kind, dtype=U64))
    if kind == 0:
        return (np.float32(0.5) + ((h >> U64(40)) & U64(0xFFFFFF)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)
    if kind == 1:
        s = ((h & U64(0xFFFF)).astype(np.float32) + ((h >> U64(16)) & U64(0xFFFF)).astype(np.float32)
             + ((h >> U64(32)) & U64(0xFFFF)).astype(np.float32) + ((h >> U64(48)) & U64(0xFFFF)).astype(np.float32))
        return ((s - np.float32(131070.0)) * np.float32(0.02 * 1.7320508 / 65536.0)).astype(np.float32)
    u = ((h >> U64(40)) & U64(0xFFFFFF)).astype(np.float32) * np.float32(1.0 / 8388608.0) - np.float32(1.0)
    return (u / np.float32(np.sqrt(np.float32(count)))).astype(np.float32)
