#!/usr/bin/env python3
"""Constructor MLP: the matrix-pipe passes (csrc/umbrella_mfma.hip) against the register-resident VALU passes they replace
(csrc/umbrella_mlp.hip), forward and backward of mlp.umbrella_mlp / umbrella_mlp2 in one hipGraph each (what the step replays),
HIP events around 50 replays.  python tools/umb_bench.py [blocks ...]"""
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from repsurf_amd import mlp, mlp_hip as H  # noqa: E402


def graph_time(fn, reps=50):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def case(layers, points, group, label):
    torch.manual_seed(0)
    if layers == 3:
        m = nn.Sequential(nn.Conv2d(10, 10, 1, bias=False), nn.BatchNorm2d(10), nn.ReLU(True), nn.Conv2d(10, 10, 1),
                          nn.BatchNorm2d(10), nn.ReLU(True), nn.Conv2d(10, 10, 1)).cuda().train()
    else:
        m = nn.Sequential(nn.Conv1d(10, 10, 1), nn.BatchNorm1d(10), nn.ReLU(True), nn.Conv1d(10, 10, 1)).cuda().train()
    x = torch.randn(points * group, 10).cuda()
    w = torch.randn(points, 10).cuda()
    mom = mlp.umbrella_moments(x)

    def fwd_bwd(moments):
        for p in m.parameters():
            p.grad = None
        out = mlp.umbrella_mlp(x, m, group, "sum", moments=moments) if layers == 3 else mlp.umbrella_mlp2(x, m, group, moments=moments)
        out.backward(w)

    def fwd_only(moments):
        with torch.no_grad():
            (mlp.umbrella_mlp(x, m, group, "sum", moments=moments) if layers == 3 else mlp.umbrella_mlp2(x, m, group, moments=moments))

    H.UMB_MFMA = False
    t_old, f_old = graph_time(lambda: fwd_bwd(None)), graph_time(lambda: fwd_only(None))
    H.UMB_MFMA = H.UMB_MFMA_FWD3 = True
    print(f"{label}: VALU passes fwd+bwd {t_old:7.1f} us (fwd {f_old:6.1f})")
    for blocks in BLOCKS:
        H.UMB_MFMA_BLOCKS = blocks
        t_new, f_new = graph_time(lambda: fwd_bwd(mom)), graph_time(lambda: fwd_only(mom))
        t_mom = graph_time(lambda: mlp.umbrella_moments(x))
        print(f"{label}: MFMA passes, {blocks:4d} workgroups: fwd+bwd {t_new:7.1f} us (fwd {f_new:6.1f}); moments (geometry stage) {t_mom:5.1f} us")
    H.UMB_MFMA_BLOCKS = 256


BLOCKS = [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024]
case(3, 32 * 1024, 8, "cls  B=32x1024, fan 8 (262 144 rows)")
case(2, 16 * 4096, 9, "seg  16x4096, fan 9 (589 824 rows)")
