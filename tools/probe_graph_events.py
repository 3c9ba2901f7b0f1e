"""Can HIP events time kernels INSIDE a replayed hipGraph on this runtime?  hipEventRecordWithFlags(hipEventRecordExternal)
under stream capture makes event-record nodes; after a replay hipEventElapsedTime between two of them is the in-graph
duration.  Prints the per-kernel times of a captured chain next to the eager back-to-back times of the same kernels."""
import ctypes
import torch

hip = ctypes.CDLL("libamdhip64.so")
P = ctypes.c_void_p
hip.hipEventCreate.argtypes = [ctypes.POINTER(P)]
hip.hipEventRecordWithFlags.argtypes = [P, P, ctypes.c_uint]
hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), P, P]
hip.hipEventSynchronize.argtypes = [P]


def ev():
    e = P()
    assert hip.hipEventCreate(ctypes.byref(e)) == 0
    return e


def main():
    x = torch.randn(48000, 128, device="cuda")
    w = torch.randn(128, 128, device="cuda")
    s = torch.cuda.Stream()
    n = 6
    events = [ev() for _ in range(n + 1)]
    with torch.cuda.stream(s):
        for _ in range(3):
            y = x @ w
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            st = torch.cuda.current_stream().cuda_stream
            rc = hip.hipEventRecordWithFlags(events[0], P(st), 1)
            print("record under capture rc", rc)
            for i in range(n):
                y = x @ w
                hip.hipEventRecordWithFlags(events[i + 1], P(st), 1)
        for rep in range(3):
            g.replay()
            torch.cuda.synchronize()
            out = []
            for i in range(n):
                t = ctypes.c_float()
                rc = hip.hipEventElapsedTime(ctypes.byref(t), events[i], events[i + 1])
                out.append((rc, round(t.value * 1e3, 1)))
            print("replay", rep, "per-kernel us (rc, us):", out)
        # eager back to back
        te = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        te[0].record()
        for i in range(n):
            y = x @ w
            te[i + 1].record()
        torch.cuda.synchronize()
        print("eager per-kernel us:", [round(te[i].elapsed_time(te[i + 1]) * 1e3, 1) for i in range(n)])


main()
