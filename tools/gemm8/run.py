"""Lab harness of the round-5 prompt GEMM (pf_gemm8.h): correctness against numpy on sampled rows, time per launch next to the round-4
kernel (mc_pf_gemm256_w_bfloat_d2_e0 of the product code object) on the shapes of a Llama-3-8B prompt.

  python tools/gemm8/run.py [--hsaco tools/gemm8/gemm8_lab.hsaco] [--reps 20]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import metalchat_amd as mc  # noqa: E402


def to_bf16(a):
    u = a.astype(np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def from_bf16(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def pack_i4(q):
    """[N][K] ints in [-8, 7] -> the decode layout: offset-binary nibbles, nibble p of a dword = weight {0,2,4,6,1,3,5,7}[p] of its 8-run"""
    n = (q + 8).astype(np.uint32).reshape(q.shape[0], -1, 8)
    order = [0, 2, 4, 6, 1, 3, 5, 7]
    d = np.zeros(n.shape[:2], np.uint32)
    for p_, w in enumerate(order):
        d |= n[:, :, w] << (4 * p_)
    return d.reshape(-1)


def quad_scales(sc):
    """[N][G] bf16 bits -> row quads [ceil(N / 4)][G][4]"""
    N, G = sc.shape
    pad = (-N) % 4
    if pad:
        sc = np.concatenate([sc, np.zeros((pad, G), sc.dtype)])
    return np.ascontiguousarray(sc.reshape(-1, 4, G).transpose(0, 2, 1)).reshape(-1)


def quant_case(lab, rng, fmt, M, N, K, group, reps, pre="gemm8", bm=256, splits=1):
    X = to_bf16(rng.uniform(-1, 1, (M, K)))
    qmax = 7 if fmt == "i4" else 127
    q = rng.integers(-qmax - 1, qmax + 1, (N, K))
    G = K // group
    sc = to_bf16(rng.uniform(0.5, 1.5, (N, G)) / qmax)
    wd = from_bf16(to_bf16(q.astype(np.float32) * np.repeat(from_bf16(sc), group, axis=1)))   # Wd = T(T(q) T(s))
    wb = lab.to_device(pack_i4(q) if fmt == "i4" else q.astype(np.int8).reshape(-1).view(np.uint8))
    sb = lab.to_device(quad_scales(sc))
    xb = lab.to_device(X.reshape(-1))
    yb = lab.to_device(np.zeros(M * N * (1 if splits == 1 else 2 * splits), np.uint16))
    kern = lab.load(f"mc_pf_{pre}_{fmt}_bfloat_e{0 if splits == 1 else 2}")
    nx, ny = (N + 255) // 256, (M + bm - 1) // bm
    task = mc.KernelTask(kern, (nx * 512, ny, splits), (512, 1, 1),
                         [wb, sb, xb, yb, None, np.uint32(M), np.uint32(N), np.uint32(K), np.uint32(group), None, None, np.uint32(0), np.float32(0)])
    task()
    lab.wait()
    if splits == 1:
        Y = yb.download(np.uint16, M * N).reshape(M, N)
    else:
        Y = to_bf16(yb.download(np.float32, splits * M * N).reshape(splits, M, N).sum(0))
    rows = sorted(set([0, 1, 63, 64, 127, 128, 255, 256 % M, M - 1, M // 2] + list(rng.integers(0, M, 6))))
    ref = from_bf16(X[rows]).astype(np.float64) @ wd.astype(np.float64).T
    got = from_bf16(Y[rows]).astype(np.float64)
    err = np.abs(got - ref) / (np.abs(ref) + np.sqrt(K) * 0.02)
    bad = int((err > 2.0 ** -7).sum())
    for _ in range(10):
        task()
    lab.wait()
    t0 = time.perf_counter()
    for _ in range(reps):
        task()
    lab.wait()
    us = (time.perf_counter() - t0) / reps * 1e6
    print(f"{M:5d} x {N:6d} x {K:6d}  {pre} {fmt} g{group:<4d} x{splits}  {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s   max rel err {err.max():.2e}  bad {bad}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hsaco", default=os.path.join(ROOT, "tools", "gemm8", "gemm8_lab.hsaco"))
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--kernel", default="mc_pf_gemm8_w_bfloat_e0")
    ap.add_argument("--shapes", default="512x28672x4096,2048x28672x4096,512x6144x4096,512x4096x4096,512x4096x14336,300x768x256,1024x28672x4096")
    ap.add_argument("--no-old", action="store_true")
    ap.add_argument("--no-splitk", action="store_true")
    ap.add_argument("--bm128", action="store_true", help="the 128-row tile variants too (gemm8h)")
    ap.add_argument("--quant", default="i4,i8", help="quantised-W variants to run per shape")
    ap.add_argument("--extra", default="ns8,nostage,nomfma", help="lab variants of --kernel: suffixes")
    args = ap.parse_args()
    lab = mc.HardwareAccelerator(path=args.hsaco, ordinal=0)
    prod = mc.HardwareAccelerator(ordinal=0)
    k_new = lab.load(args.kernel)
    k_part = lab.load("mc_pf_gemm8_w_bfloat_e2")
    k_old = prod.load("mc_pf_gemm256_w_bfloat_d2_e0")
    rng = np.random.default_rng(0)
    # (wake the device up: the first timed loop of a process otherwise measures the clocks coming up)
    wx = lab.to_device(np.zeros(1 << 24, np.uint16))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        mc.KernelTask(k_new, (64 * 512, 2, 1), (512, 1, 1), [wx, None, wx, wx, None, np.uint32(512), np.uint32(2048), np.uint32(4096), np.uint32(0), None, None, np.uint32(0), np.float32(0)])()
        lab.wait()
    for shp in args.shapes.split(","):
        M, N, K = (int(x) for x in shp.split("x"))
        if args.quant:
            for fmt in args.quant.split(","):
                quant_case(lab, rng, fmt, M, N, K, 128 if K % 128 == 0 else 32, args.reps)
                if args.bm128:
                    for sp in (1, 2, 4):
                        if (K // 64) % sp == 0 and (sp == 1 or fmt == "i4"):
                            quant_case(lab, rng, fmt, M, N, K, 128 if K % 128 == 0 else 32, args.reps, "gemm8h", 128, sp)
                quant_case(lab, rng, fmt, M, N, K, 128 if K % 128 == 0 else 32, args.reps, "gemm8x", 256, 1)   # 32 x 32 x 16 MFMAs
        X = to_bf16(rng.uniform(-1, 1, (M, K)))
        W = to_bf16(rng.uniform(-1, 1, (N, K)))
        variants = [(lab, k_new, "gemm8", 256, True)]
        for extra in args.extra.split(","):
            if extra:
                variants.append((lab, lab.load(args.kernel + "_" + extra), "gemm8_" + extra, 256, extra.startswith("ns")))
        if not args.no_old:
            variants.append((prod, k_old, "round-4 gemm256", 128, True))
        for acc, kern, name, tile_n, check in variants:
            xb, wb = acc.to_device(X.reshape(-1)), acc.to_device(W.reshape(-1))
            yb = acc.to_device(np.zeros(M * N, np.uint16))
            nx, ny = (N + tile_n - 1) // tile_n, (M + 255) // 256
            task = mc.KernelTask(kern, (nx * 512, ny, 1), (512, 1, 1),
                                 [wb, None, xb, yb, None, np.uint32(M), np.uint32(N), np.uint32(K), np.uint32(0), None, None, np.uint32(0), np.float32(0)])
            task()
            acc.wait()
            Y = yb.download(np.uint16, M * N).reshape(M, N)
            rows = sorted(set([0, 1, 127, 128, 255, 256, M - 1, M // 2] + list(rng.integers(0, M, 6))))
            rows = [r for r in rows if r < M]
            ref = from_bf16(X[rows]).astype(np.float64) @ from_bf16(W).astype(np.float64).T
            got = from_bf16(Y[rows]).astype(np.float64)
            err = np.abs(got - ref) / (np.abs(ref) + np.sqrt(K) * 0.02)
            bad = int((err > 2.0 ** -7).sum()) if check else -1
            for _ in range(10):
                task()
            acc.wait()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                task()
            acc.wait()
            us = (time.perf_counter() - t0) / args.reps * 1e6
            print(f"{M:5d} x {N:6d} x {K:6d}  {name:16s} {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s   max rel err {err.max():.2e}  bad {bad}", flush=True)
        if not args.no_splitk:
            # split-K form: fp32 partials of 2 and 4 K ranges summed on the host
            for splits in (2, 4):
                if (K // 64) % splits:
                    continue
                xb, wb = lab.to_device(X.reshape(-1)), lab.to_device(W.reshape(-1))
                pb = lab.to_device(np.zeros(splits * M * N, np.float32))
                nx, ny = (N + 255) // 256, (M + 255) // 256
                task = mc.KernelTask(k_part, (nx * 512, ny, splits), (512, 1, 1),
                                     [wb, None, xb, pb, None, np.uint32(M), np.uint32(N), np.uint32(K), np.uint32(0), None, None, np.uint32(0), np.float32(0)])
                task()
                lab.wait()
                P = pb.download(np.float32, splits * M * N).reshape(splits, M, N).sum(0)
                rows = [0, M - 1, M // 3]
                ref = from_bf16(X[rows]).astype(np.float64) @ from_bf16(W).astype(np.float64).T
                err = np.abs(P[rows] - ref) / (np.abs(ref) + np.sqrt(K) * 0.02)
                for _ in range(3):
                    task()
                lab.wait()
                t0 = time.perf_counter()
                for _ in range(args.reps):
                    task()
                lab.wait()
                us = (time.perf_counter() - t0) / args.reps * 1e6
                print(f"{M:5d} x {N:6d} x {K:6d}  gemm8 split-K {splits:2d} {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s   max rel err {err.max():.2e}", flush=True)


if __name__ == "__main__":
    main()
