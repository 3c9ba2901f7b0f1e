"""The two HIP streams of the pipelined steps (graph.PipelinedStep: geometry graph beside the network graph).

`pair(device)` -> (main, side): two streams of torch's pool (non-blocking), what every measured number of this repository ran on.

REPSURF_STREAM_KIND=masked|blocking|nonblocking makes the pair outside torch's pool instead -- DIAGNOSIS of the two-stream hazard
(profiles/r06/eager_beside_graph.txt): hipExtStreamCreateWithCUMask with disjoint compute-unit sets / plain
hipStreamCreateWithFlags.  What the round-6 runs of tools/pipelined_flake.py and tools/cu_mask_probe.py showed about them:
  * the CU mask is NOT applied on the GPU boxes of this pool (an 8192^3 matmul takes 7.15 ms on a "32-CU" stream, on a "224-CU" stream and
    on a pool stream; the call itself reports success), so the pair cannot keep the two graphs on different compute units here;
  * streams made by hipExtStreamCreateWithCUMask are BLOCKING streams (the call takes no flags): an event recorded on the legacy default
    stream between the two launches orders the geometry after the network, which is why a first run of this kind showed no deviating
    step -- it had no concurrency left; with the default stream kept out of the way it deviates like the others (worse);
  * none of the kinds removes the hazard (its cause was found in the victim kernel's code instead: the Makefile's vectorizer flags).
    They stay selectable for whoever wants to repeat the diagnosis on a box where the mask works."""
import ctypes
import os

import torch

KIND = os.environ.get("REPSURF_STREAM_KIND", "")
SIDE_CUS = int(os.environ.get("REPSURF_SIDE_CUS", "32"))
_hip = None
_pairs = {}         # device index -> (main, side) of the diagnosis kinds: made once (each owns a hardware queue), kept for the process


def _runtime():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        _hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
        _hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32]
        _hip.hipStreamCreateWithFlags.restype = ctypes.c_int
    return _hip


def _masked(device, cus, total):
    """A stream asked to run its kernels on the compute units listed in `cus` (indices < total)."""
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for c in cus:
        mask[c // 32] |= 1 << (c % 32)
    raw = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = _runtime().hipExtStreamCreateWithCUMask(ctypes.byref(raw), words, mask)
    if rc != 0 or not raw.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed (hip error {rc})")
    return torch.cuda.ExternalStream(raw.value, device=device)


def _flagged(device, flags):
    raw = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = _runtime().hipStreamCreateWithFlags(ctypes.byref(raw), flags)
    if rc != 0 or not raw.value:
        raise RuntimeError(f"hipStreamCreateWithFlags failed (hip error {rc})")
    return torch.cuda.ExternalStream(raw.value, device=device)


def pair(device=None):
    """(main, side) for a pipelined step on `device`."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if KIND not in ("masked", "blocking", "nonblocking"):
        return torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
    if device.index not in _pairs:
        if KIND == "masked":
            total = torch.cuda.get_device_properties(device).multi_processor_count
            n = min(max(1, SIDE_CUS), total - 1)
            side = sorted({int(i * total / n) for i in range(n)})          # every (total / n)-th unit
            main = [c for c in range(total) if c not in set(side)]
            _pairs[device.index] = (_masked(device, main, total), _masked(device, side, total))
        else:
            flags = 0 if KIND == "blocking" else 1                          # hipStreamDefault / hipStreamNonBlocking
            _pairs[device.index] = (_flagged(device, flags), _flagged(device, flags))
    return _pairs[device.index]


def after(stream, caller):
    """Order `stream`'s next work after what `caller` holds now.  A BLOCKING stream (kinds masked / blocking) is already ordered after
    the legacy default stream; an event recorded there would also wait for the OTHER stream of the pair."""
    if KIND in ("masked", "blocking") and caller.cuda_stream == 0:
        return
    stream.wait_stream(caller)
