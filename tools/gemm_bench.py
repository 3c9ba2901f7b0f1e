#!/usr/bin/env python3
"""Micro-benchmark of rs_mlp_gemm_rows / rs_mlp_wgrad at the step's shapes (GPU box).
Separates operand prologue, MFMA main loop and epilogue costs by timing the same shape in different modes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repsurf_amd import mlp_hip as H

dev = torch.device("cuda")
PEAK = 157.3


def timeit(fn, iters=20):
    """GPU time per launch: `iters` launches recorded into a hipGraph and replayed (a ctypes call costs ~10 us of host time,
    more than the kernels under test; the training step replays a graph too).  GEMM_BENCH_EAGER=1: plain back-to-back calls."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if os.environ.get("GEMM_BENCH_EAGER", "0") == "1":
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3   # us
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e3   # us


def bench(rows, k, n, rows_dev_frac=None):
    x = torch.randn(rows, k, device=dev)
    y_prev = torch.randn(rows, n, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    wk = H.w_fwd(w)
    out = torch.empty(rows, n, device=dev)
    s = torch.rand(k, device=dev) + 0.5
    t = torch.randn(k, device=dev) * 0.1
    sn = torch.rand(n, device=dev) + 0.5
    tn = torch.randn(n, device=dev) * 0.1
    part = torch.empty((H.PARTIAL_BLOCKS, 3, n), dtype=torch.float64, device=dev)
    rows_dev = None
    rd = None
    if rows_dev_frac:
        rd = torch.tensor([int(rows * rows_dev_frac)], dtype=torch.int32, device=dev)
        rows_dev = rd.data_ptr()
    eff_rows = int(rows * rows_dev_frac) if rows_dev_frac else rows
    res = {}
    ops = {"ID": H.operand(H.OP_ID, x, k), "RELU1": H.operand(H.OP_RELU1, x, k, s1=s, t1=t),
           "AFF2": H.operand(H.OP_AFF2, x, k, x, k, s1=s, t1=t, s2=s)}
    epis = {"STORE": H.Epilogue(bias=None, out=H._ptr(out), ldo=n, mode=H.EPI_STORE),
            "STATS": H.Epilogue(bias=None, out=H._ptr(out), ldo=n, mode=H.EPI_STATS, partial=part.data_ptr(), partial_blocks=H.PARTIAL_BLOCKS),
            "MASK": H.Epilogue(bias=None, out=H._ptr(out), ldo=n, mode=H.EPI_MASK, my1=H._ptr(y_prev), ldm1=n, ms1=H._ptr(sn), mt1=H._ptr(tn),
                               mean1=H._ptr(tn), invstd1=H._ptr(sn), partial=part.data_ptr(), partial_blocks=H.PARTIAL_BLOCKS)}
    for on, en in (("ID", "STORE"), ("RELU1", "STORE"), ("RELU1", "STATS"), ("AFF2", "MASK")):
        us = timeit(lambda: H.gemm_rows(rows, k, n, ops[on], wk, epis[en], rows_dev))
        res[f"{on}+{en}"] = us
    fl = 2.0 * eff_rows * k * n
    line = f"gemm rows={eff_rows:>7} (cap {rows:>7}) K={k:>4} N={n:>4} | " + " | ".join(
        f"{m}: {us:7.1f}us {fl / us / 1e6:5.1f}TF" for m, us in res.items())
    print(line, flush=True)


def bench_wgrad(rows, n, k, rows_dev_frac=None):
    p = torch.randn(rows, n, device=dev)
    q = torch.randn(rows, k, device=dev)
    rows_dev = None
    if rows_dev_frac:
        rd = torch.tensor([int(rows * rows_dev_frac)], dtype=torch.int32, device=dev)
        rows_dev = rd.data_ptr()
    eff_rows = int(rows * rows_dev_frac) if rows_dev_frac else rows
    us = timeit(lambda: H.wgrad(rows, n, k, H.operand(H.OP_ID, p, n), H.operand(H.OP_ID, q, k), dev, rows_dev))
    fl = 2.0 * eff_rows * k * n
    print(f"wgrad rows={eff_rows:>7} (cap {rows:>7}) N={n:>4} K={k:>4} | {us:7.1f}us {fl / us / 1e6:5.1f}TF", flush=True)


def bench_cold(rows, k, n, frac=None, mode=("RELU1", "STATS")):
    """Same launch over a ring of buffers larger than the 256 MB Infinity Cache: every launch reads HBM-cold data
    (what happens inside a training step, where each layer's input was written ~1 GB of traffic earlier)."""
    eff_rows = int(rows * frac) if frac else rows
    per = eff_rows * (k + n) * 4
    nbuf = max(2, int(1.5e9 // max(per, 1)))
    nbuf = min(nbuf, 64)
    xs = [torch.randn(eff_rows if frac is None else rows, k, device=dev) for _ in range(nbuf)]
    outs = [torch.empty(rows, n, device=dev) for _ in range(nbuf)]
    w = torch.randn(n, k, device=dev) / k ** 0.5
    wk = H.w_fwd(w)
    s = torch.rand(k, device=dev) + 0.5
    t = torch.randn(k, device=dev) * 0.1
    part = torch.empty((H.PARTIAL_BLOCKS, 3, n), dtype=torch.float64, device=dev)
    rd = torch.tensor([eff_rows], dtype=torch.int32, device=dev)
    rows_dev = rd.data_ptr() if frac else None
    state = {"i": 0}

    def fn():
        i = state["i"] = (state["i"] + 1) % nbuf
        op = H.operand(H.OP_RELU1, xs[i], k, s1=s, t1=t)
        ep = H.Epilogue(bias=None, out=H._ptr(outs[i]), ldo=n, mode=H.EPI_STATS, partial=part.data_ptr(), partial_blocks=H.PARTIAL_BLOCKS)
        H.gemm_rows(rows, k, n, op, wk, ep, rows_dev)
    us = timeit(fn, iters=2 * nbuf)
    fl = 2.0 * eff_rows * k * n
    print(f"COLD gemm rows={eff_rows:>7} K={k:>4} N={n:>4} ring={nbuf:>2} x {per / 1e6:6.1f}MB | RELU1+STATS {us:7.1f}us {fl / us / 1e6:5.1f}TF "
          f"{per / us / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "wtiming":    # wgrad phases; needs REPSURF_HIP_LIB=build_exp/librepsurf_TIMING.so
        rows, n, k = (int(v) for v in sys.argv[2:5])
        frac = float(sys.argv[5]) if len(sys.argv) > 5 else None
        p_ = torch.randn(rows, n, device=dev); q_ = torch.randn(rows, k, device=dev)
        dbg = torch.zeros((8192, 10), dtype=torch.int64, device=dev)
        rd = torch.tensor([int(rows * frac)], dtype=torch.int32, device=dev) if frac else None
        pop = H.operand(H.OP_ID, p_, n)
        pop.t2 = dbg.data_ptr()
        for _ in range(3):
            H.wgrad(rows, n, k, pop, H.operand(H.OP_ID, q_, k), dev, rd.data_ptr() if frac else None)
        torch.cuda.synchronize()
        d = dbg.cpu().numpy()
        act = d[d[:, 7] > 0]
        names = ["first prefetch issue", "commit (wait loads + transform + LDS writes)", "barrier", "next prefetch issue",
                 "fragment reads + MFMA issue", "-", "epilogue (partial tile stores)", "TOTAL"]
        print(f"wgrad rows={rows} n={n} k={k} frac={frac}: {len(act)} workgroups, stages/WG={act[:, 8].mean():.1f}")
        for i, nm in enumerate(names):
            print(f"  {nm:48s} {act[:, i].mean():10.0f} cycles  ({100 * act[:, i].mean() / act[:, 7].mean():5.1f}%)")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "timing":     # needs REPSURF_HIP_LIB=build_exp/librepsurf_TIMING.so
        rows, k, n = (int(v) for v in sys.argv[2:5])
        frac = float(sys.argv[5]) if len(sys.argv) > 5 else None
        x = torch.randn(rows, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5
        out = torch.empty(rows, n, device=dev); s_ = torch.rand(k, device=dev) + 0.5; t_ = torch.randn(k, device=dev) * 0.1
        part = torch.empty((H.PARTIAL_BLOCKS, 3, n), dtype=torch.float64, device=dev)
        dbg = torch.zeros((8192, 10), dtype=torch.int64, device=dev)
        rd = torch.tensor([int(rows * frac)], dtype=torch.int32, device=dev) if frac else None
        op = H.operand(H.OP_RELU1, x, k, s1=s_, t1=t_)
        ep = H.Epilogue(bias=None, out=H._ptr(out), ldo=n, mode=H.EPI_STATS, partial=part.data_ptr(), partial_blocks=H.PARTIAL_BLOCKS)
        ep.pool_amax = dbg.data_ptr()
        for _ in range(3):
            H.gemm_rows(rows, k, n, op, H.w_fwd(w), ep, rd.data_ptr() if frac else None)
        torch.cuda.synchronize()
        d = dbg.cpu().numpy()
        names = ["first prefetch issue", "commit (wait loads + transform + LDS writes)", "barrier", "next prefetch issue",
                 "fragment reads + MFMA issue", "barrier before epilogue", "epilogue (C tile, stats, stores)", "TOTAL"]
        for tag, blk in (("wave 0 (MFMA wave of a specialised workgroup)", d[:4096]), ("first LOADER wave (wave-specialised instances only)", d[4096:])):
            act = blk[blk[:, 7] > 0]
            if len(act) == 0:
                continue
            print(f"{tag}: rows={rows} k={k} n={n} frac={frac}: {len(act)} workgroups reported, tiles={act[0, 9]}")
            for i, nm in enumerate(names):
                print(f"  {nm:48s} {act[:, i].mean():10.0f} cycles  ({100 * act[:, i].mean() / act[:, 7].mean():5.1f}%)")
            print(f"  {'(wait for the chunk global loads, own stamp)':48s} {act[:, 8].mean():10.0f} cycles  ({100 * act[:, 8].mean() / act[:, 7].mean():5.1f}%)")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "one":        # one shape, for PMC runs: one <rows> <k> <n> [frac]
        rows, k, n = (int(v) for v in sys.argv[2:5])
        frac = float(sys.argv[5]) if len(sys.argv) > 5 else None
        bench(rows, k, n, frac)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mid":         # 64 ... 128 channels on both sides (gemm_mid_kernel; RS_GEMM_MID=0: the tiled kernel)
        for rows, k, n, frac in [(524288, 64, 64, 0.127), (524288, 64, 128, 0.127), (524288, 128, 64, 0.127), (262144, 128, 64, 0.184),
                                 (262144, 64, 128, 0.184), (65536, 64, 64, None)]:
            bench(rows, k, n, frac)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "wnarrow":     # narrow weight gradients (kcols <= 16): dense segmentation shapes and compacted classification ones
        for rows, n, k, frac in [(524288, 32, 16, None), (524288, 32, 6, None), (524288, 64, 10, None), (524288, 64, 10, 0.127), (262144, 128, 6, 0.184)]:
            bench_wgrad(rows, n, k, frac)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "small":       # narrow first-layer GEMMs (kdim <= 16)
        for rows, k, n, frac in [(524288, 6, 64, 0.127), (524288, 10, 64, 0.127), (262144, 6, 128, 0.184), (4096, 6, 256, None),
                                 (524288, 3, 32, None), (524288, 16, 32, None)]:
            x = torch.randn(rows, 16, device=dev)
            w = torch.randn(n, k, device=dev)
            wk = H.w_fwd(w)
            out = torch.empty(rows, n, device=dev)
            part = torch.empty((H.PARTIAL_BLOCKS, 3, n), dtype=torch.float64, device=dev)
            rd = torch.tensor([int(rows * frac)], dtype=torch.int32, device=dev) if frac else None
            eff = int(rows * frac) if frac else rows
            op = H.operand(H.OP_ID, x, 16)
            ep = H.Epilogue(bias=None, out=H._ptr(out), ldo=n, mode=H.EPI_STATS, partial=part.data_ptr(), partial_blocks=H.PARTIAL_BLOCKS)
            us = timeit(lambda: H.gemm_rows(rows, k, n, op, wk, ep, rd.data_ptr() if frac else None))
            byt = eff * (16 + n) * 4
            print(f"small gemm rows={eff:>7} K={k:>3} N={n:>4} ID+STATS: {us:7.1f} us  {byt / us / 1e3:7.1f} GB/s", flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cold":
        for rows, k, n, frac in [(262144, 128, 128, None), (262144, 128, 128, 0.184), (262144, 256, 128, 0.184), (262144, 128, 256, 0.184),
                                 (524288, 64, 64, 0.127), (524288, 64, 128, 0.127), (4096, 1024, 512, None), (4096, 512, 1024, None),
                                 (4096, 256, 512, None), (4096, 272, 256, None)]:
            bench_cold(rows, k, n, frac)
        sys.exit(0)
    for rows, k, n, frac in [(262144, 128, 128, None), (262144, 128, 256, None), (262144, 256, 128, None),
                             (262144, 128, 128, 0.184), (262144, 128, 256, 0.184), (262144, 256, 128, 0.184),
                             (524288, 64, 64, 0.127), (524288, 64, 128, 0.127), (524288, 128, 64, 0.127),
                             (65536, 128, 128, None), (32768, 128, 128, None), (16384, 128, 128, None),
                             (4096, 1024, 512, None), (4096, 512, 1024, None), (4096, 256, 512, None), (4096, 512, 256, None),
                             (4096, 272, 256, None)]:
        bench(rows, k, n, frac)
    for rows, n, k, frac in [(524288, 32, 16, None), (524288, 32, 3, None), (131072, 64, 74, None), (524288, 64, 10, 0.127),
                             (262144, 256, 128, None), (262144, 256, 128, 0.184), (262144, 128, 128, 0.184),
                             (524288, 128, 64, 0.127), (524288, 64, 64, 0.127), (4096, 1024, 512, None),
                             (4096, 512, 256, None), (4096, 256, 272, None)]:
        bench_wgrad(rows, n, k, frac)
