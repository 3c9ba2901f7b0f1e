#!/usr/bin/env python3
"""Vendor-library reference point for the step's row GEMMs: torch.mm (hipBLASLt / rocBLAS fp32) vs rs_mlp_gemm_rows
(identity prologue, plain store) at the same shapes, timed back to back in a loop (launch gaps hidden by queue depth).
Run on the GPU box: python tools/blas_compare.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repsurf_amd import mlp_hip as H

dev = torch.device("cuda")
SHAPES = [(66754, 64, 64), (66754, 64, 128), (48202, 138, 128), (48202, 128, 128), (48202, 128, 256),
          (4096, 266, 256), (4096, 256, 512), (4096, 512, 1024), (4096, 1024, 512), (48202, 256, 128), (262144, 128, 128)]


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for rows, k, n in SHAPES:
    kp = (k + 3) // 4 * 4
    x = torch.randn(rows, kp, device=dev)
    w = torch.randn(n, kp, device=dev) / k ** 0.5
    wt = w.t().contiguous()
    out = torch.empty(rows, n, device=dev)
    t_mm = timeit(lambda: torch.mm(x, wt, out=out))
    t_nt = timeit(lambda: torch.mm(x, w.t(), out=out))
    wk = H.w_fwd(w)
    op = H.operand(H.OP_ID, x, kp)
    epi = H.Epilogue(bias=None, out=H._ptr(out), ldo=n, mode=H.EPI_STORE)
    t_rs = timeit(lambda: H.gemm_rows(rows, kp, n, op, wk, epi, None))
    fl = 2.0 * rows * kp * n
    print(f"rows={rows:>7} K={kp:>4} N={n:>4} | torch.mm NN {t_mm:7.1f}us {fl/t_mm/1e6:5.1f}TF | NT {t_nt:7.1f}us {fl/t_nt/1e6:5.1f}TF"
          f" | rs_mlp_gemm_rows {t_rs:7.1f}us {fl/t_rs/1e6:5.1f}TF", flush=True)
