#!/usr/bin/env python3
"""SA stack: HIP bf16 mode vs the restated executor (tests/torch_executor.py, torch_bf16), per output and gradient (GPU box)."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests import torch_executor
from tests.test_mlp_gpu import make_cd, run_cd, CASES
from repsurf_amd import mlp


def cos(a, b):
    return torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()


for groups, ns, pos, feat, widths in CASES[:4]:
    mod = make_cd(pos, feat, widths, 1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(groups * ns, pos + feat, generator=g).cuda()
    w = torch.randn(groups, widths[-1], generator=g).cuda()
    out_f, g_f = run_cd(copy.deepcopy(mod), x, ns, pos, "hip", w)
    mlp.set_precision("bf16")
    try:
        out_b, g_b = run_cd(copy.deepcopy(mod), x, ns, pos, "hip", w)
    finally:
        mlp.set_precision("fp32")
    out_r, g_r = run_cd(copy.deepcopy(mod), x, ns, pos, "torch_bf16", w)
    torch_executor.set_backend("hip")
    sc = out_f.abs().max().item()
    print(f"case {groups}x{ns} pos {pos} feat {feat} {widths}: out vs fp32 {(out_b - out_f).abs().max().item() / sc:.2e}; vs restated {(out_b - out_r).abs().max().item() / sc:.2e}; "
          f"elements differing {int((out_b != out_r).sum())} of {out_b.numel()}")
    for name in g_f:
        if g_f[name].abs().max() == 0:
            continue
        print(f"      {name:18s} cos vs fp32 {cos(g_b[name], g_f[name]):.5f}   vs restated {cos(g_b[name], g_r[name]):.6f}  rel-L2 {float((g_b[name] - g_r[name]).norm() / g_r[name].norm()):.2e}")
