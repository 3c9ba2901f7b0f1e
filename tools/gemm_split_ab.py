#!/usr/bin/env python3
"""fp32 product as six bf16 MFMAs over three-part operands (unit 4 of csrc/mlp.hip, RS_GEMM_SPLIT3=1) against the fp32 MFMA
instances: error against an fp64 product and time per launch at the step's shapes.  The switch is read once per process:
    RS_GEMM_SPLIT3=0 python tools/gemm_split_ab.py      # fp32 MFMA
    python tools/gemm_split_ab.py                       # the default: six bf16 MFMAs over three-part operands
(GPU box; round 4 ran both and the step A/B: profiles/r04/gemm_split3_ab.txt)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repsurf_amd import mlp_hip as H
from tools.gemm_bench import timeit

dev = torch.device("cuda")
tag = "split3 (6 x bf16 MFMA)" if H.gemm_split3() else "fp32 MFMA"


def accuracy(rows, k, n):
    g = torch.Generator(device="cpu").manual_seed(rows + k + n)
    x = torch.randn(rows, k, generator=g).to(dev)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
    s = (torch.rand(k, generator=g) + 0.5).to(dev)
    t = (torch.randn(k, generator=g) * 0.1).to(dev)
    out = torch.empty(rows, n, device=dev)
    H.gemm_rows(rows, k, n, H.operand(H.OP_RELU1, x, k, s1=s, t1=t), H.w_fwd(w), H.Epilogue(bias=None, out=H._ptr(out), ldo=n, mode=H.EPI_STORE))
    e = torch.relu(x * s + t)                        # the prologue in fp32 (a rounding apart from the kernel's: below the errors measured here)
    ref = e.double() @ w.double().t()
    err = (out.double() - ref).abs()
    p = torch.randn(rows, n, generator=g).to(dev)
    dw = H.wgrad(rows, n, k, H.operand(H.OP_ID, p, n), H.operand(H.OP_ID, x, k), dev)
    refw = p.double().t() @ x.double()
    errw = (dw.double() - refw).abs()
    print(f"[{tag}] rows={rows:>7} K={k:>4} N={n:>4}: forward max |err| {err.max().item():.2e} rms {err.pow(2).mean().sqrt().item():.2e} "
          f"(output rms {ref.pow(2).mean().sqrt().item():.2f}) | weight gradient max |err| {errw.max().item():.2e} rms "
          f"{errw.pow(2).mean().sqrt().item():.2e} (rms {refw.pow(2).mean().sqrt().item():.1f})", flush=True)


def time_shapes():
    for rows, k, n in [(4096, 512, 1024), (4096, 1024, 512), (4096, 256, 512), (49152, 128, 128), (49152, 128, 256), (66560, 64, 128),
                       (66560, 64, 64), (524288, 32, 32), (524288, 32, 64), (131072, 64, 64), (131072, 64, 128)]:
        x = torch.randn(rows, k, device=dev)
        w = torch.randn(n, k, device=dev) / k ** 0.5
        wk = H.w_fwd(w)
        out = torch.empty(rows, n, device=dev)
        s = torch.rand(k, device=dev) + 0.5
        t = torch.randn(k, device=dev) * 0.1
        part = torch.empty((H.PARTIAL_BLOCKS, 3, n), dtype=torch.float64, device=dev)
        op = H.operand(H.OP_RELU1, x, k, s1=s, t1=t)
        ep = H.Epilogue(bias=None, out=H._ptr(out), ldo=n, mode=H.EPI_STATS, partial=part.data_ptr(), partial_blocks=H.PARTIAL_BLOCKS)
        us = timeit(lambda: H.gemm_rows(rows, k, n, op, wk, ep))
        p = torch.randn(rows, n, device=dev)
        usw = timeit(lambda: H.wgrad(rows, n, k, H.operand(H.OP_ID, p, n), H.operand(H.OP_ID, x, k), dev))
        fl = 2.0 * rows * k * n
        print(f"[{tag}] rows={rows:>7} K={k:>4} N={n:>4}: RELU1+STATS {us:7.1f} us {fl / us / 1e6:6.1f} TF | weight gradient (+ reduction) {usw:7.1f} us "
              f"{fl / usw / 1e6:6.1f} TF", flush=True)


if __name__ == "__main__":
    for shape in [(4096, 512, 1024), (66560, 64, 128), (4099, 272, 256)]:
        accuracy(*shape)
    time_shapes()
