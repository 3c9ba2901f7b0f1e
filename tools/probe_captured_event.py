"""The HIP behaviour behind the round-3/4 aborts of the sharded steps, shown in isolation (no process group):
an event recorded EAGERLY on stream S, queried from another thread while S is being captured, fails with hipErrorCapturedEvent
and invalidates the capture; the same event recorded on a stream that is not captured can be queried freely.
    python tools/probe_captured_event.py"""
import threading

import torch


def trial(record_on_capture_stream):
    s, other = torch.cuda.Stream(), torch.cuda.Stream()
    e = torch.cuda.Event()
    with torch.cuda.stream(s if record_on_capture_stream else other):
        torch.zeros(8, device="cuda").add_(1)
        e.record()
    torch.cuda.synchronize()
    seen = {}

    def poll():
        try:
            seen["query"] = e.query()
        except Exception as ex:  # noqa: BLE001
            seen["query"] = "raised: " + str(ex).splitlines()[0]

    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            y = torch.zeros(8, device="cuda") + 1
            t = threading.Thread(target=poll)
            t.start()
            t.join()
            y = y * 2
        seen["capture"] = "ok"
    except Exception as ex:  # noqa: BLE001
        seen["capture"] = "raised: " + str(ex).splitlines()[0]
    torch.cuda.synchronize()
    return seen


if __name__ == "__main__":
    print("event last recorded (eagerly, long complete) on the stream being captured :", trial(True))
    print("event last recorded on another stream                                     :", trial(False))
