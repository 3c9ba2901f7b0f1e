#!/usr/bin/env python3
"""Cost of one tiny dependent kernel inside a replayed hipGraph (no profiler attached)."""
import os, sys, time
import torch
dev = torch.device("cuda")
x = torch.zeros(1024, device=dev)


def t_us(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for count in (1, 50, 200, 400):
    def work():
        y = x
        for _ in range(count):
            y = y + 1.0
        return y
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        work(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            keep = work()
    rep = t_us(g.replay)
    print(f"{count:4d} dependent tiny kernels: graph replay {rep:8.1f} us  -> {rep / count:6.2f} us per kernel", flush=True)
    def fills():
        for _ in range(count):
            x.zero_()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fills(); torch.cuda.synchronize()
        with torch.cuda.graph(g2, stream=s):
            fills()
    rep = t_us(g2.replay)
    print(f"{count:4d} zero_() fills:           graph replay {rep:8.1f} us  -> {rep / count:6.2f} us per kernel", flush=True)
