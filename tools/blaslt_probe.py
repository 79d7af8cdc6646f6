"""Tuning aid: what the ROCm library GEMM (torch.mm -> hipBLASLt / rocBLAS) reaches on the prompt pass's shapes with plain
bfloat16 operands -- the ceiling a dequantised copy of the weights could buy.  usage: python tools/blaslt_probe.py [M ...]"""
import sys
import torch

Ms = [int(a) for a in sys.argv[1:]] or [512, 2048]
shapes = {"wq|wk|wv": (6144, 4096), "wo": (4096, 4096), "w1|w3": (28672, 4096), "w2": (4096, 14336)}
for M in Ms:
    tot = 0.0
    for name, (N, K) in shapes.items():
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
        for _ in range(5):
            y = torch.nn.functional.linear(x, w)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        reps = 20
        for _ in range(reps):
            y = torch.nn.functional.linear(x, w)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / reps
        tot += us
        print(f"M {M:5d} {name:9s} N {N:6d} K {K:6d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
    print(f"M {M:5d} the four GEMMs of a layer: {tot:8.1f} us", flush=True)
