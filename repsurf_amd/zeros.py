"""Exact-zero fp32 slices for a backward pass out of ONE fill kernel.

Several backward functions hand out gradients that are zero by construction (conv / Linear biases in front of a
BatchNorm).  Each used to fill its own buffer: one small launch per stack and pass.  `take(n, device)` carves them out of a
pool that is allocated and filled at the first request of a backward pass and dropped when the autograd engine finishes
the pass (queue_callback), so every pass -- every hipGraph replay included: the fill is captured with the pass -- gets
fresh zeros, and every slice is distinct memory (an in-place op on one gradient cannot alias another)."""
import torch

CAPACITY = 8192            # floats per pool chunk (the classifier needs ~4.3 k per pass)
_pool = {}                 # (device, stream) -> [buffer, used]: the fill that makes a pool is ordered on ONE stream
_armed = False


def _end_of_pass():
    global _armed
    _pool.clear()
    _armed = False


def take(n, device):
    global _armed
    if not _armed:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_end_of_pass)
            _armed = True
        except RuntimeError:                 # not inside a backward pass: nothing to scope a pool to
            return torch.zeros((n,), dtype=torch.float32, device=device)
    key = (str(device), torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    ent = _pool.get(key)
    if ent is None or ent[1] + n > ent[0].numel():
        ent = [torch.zeros((max(CAPACITY, n),), dtype=torch.float32, device=device), 0]
        _pool[key] = ent
    out = ent[0][ent[1]:ent[1] + n]
    ent[1] += n
    return out
