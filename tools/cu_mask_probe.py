"""Is the CU mask of streams.pair (REPSURF_STREAM_KIND=masked) honoured -- by eager launches, and by graph replays launched on the stream?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["REPSURF_STREAM_KIND"] = "masked"
import torch
from repsurf_amd import streams
dev = torch.device("cuda")
a = torch.randn(8192, 8192, device=dev); b = torch.randn(8192, 8192, device=dev); c = torch.empty_like(a)
def t(stream, fn, n=5):
    with torch.cuda.stream(stream):
        fn(); fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3)
for cus in (32, 128):
    streams.SIDE_CUS = cus; streams._pairs.clear()
    main, side = streams.pair(dev)
    plain = torch.cuda.Stream()
    mm = lambda: torch.matmul(a, b, out=c)
    print("side share", cus, "eager matmul ms: plain", t(plain, mm), "main", t(main, mm), "side", t(side, mm))
    for name, s in (("plain", plain), ("main", main), ("side", side)):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            torch.matmul(a, b, out=c)
        print("   graph captured on", name, "replayed on: plain", t(plain, g.replay), "main", t(main, g.replay), "side", t(side, g.replay))
