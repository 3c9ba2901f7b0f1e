#!/usr/bin/env python3
"""Do two independent branches of a captured hipGraph run concurrently?  FPS (32 workgroups, 250 us latency chain)
on a side stream against the umbrella kernel (512 workgroups) on the capture stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repsurf_amd import ops

dev = torch.device("cuda")
xyz = (torch.rand(32, 1024, 3, device=dev) * 2 - 1).contiguous()
start = torch.zeros(32, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()


def t_us(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def work(order, fork=True, extra_main=0):
    main = torch.cuda.current_stream()
    if fork:
        side.wait_stream(main)
    def a():
        ctx = torch.cuda.stream(side) if fork else torch.cuda.stream(main)
        with ctx:
            return ops.furthestsampling(xyz, 512, start)
    def b():
        out = ops.umbrella_features(xyz, 9, None)
        for _ in range(extra_main):
            out = out + 1.0
        return out
    r = [a(), b()] if order == "fps_first" else [b(), a()][::-1]
    if fork:
        main.wait_stream(side)
    return r


for order in ("fps_first", "umb_first"):
    for fork in (False, True):
        for extra in (0, 8):
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                work(order, fork, extra)
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=s):
                    keep = work(order, fork, extra)
            eager = t_us(lambda: work(order, fork, extra))
            rep = t_us(g.replay)
            print(f"order={order:9s} fork={fork!s:5s} extra_main_kernels={extra}: eager {eager:7.1f} us, graph replay {rep:7.1f} us", flush=True)
