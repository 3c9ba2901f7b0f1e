"""Adam on the HIP kernel of csrc/adam.hip: the optimizer of the reference's training tools.

`Adam(params, lr, betas, eps, weight_decay)` has torch.optim.Adam's update rule, defaults, `param_groups` (LR schedulers
work on it) and state-dict layout (`step`, `exp_avg`, `exp_avg_sq` per parameter; classification/tool/
train_cls_scanobjectnn.py:179-185 builds it, :166-170 / :261-271 resume from / save its state dict).  The ~70 parameter
tensors of a RepSurf-U classifier are updated by one launch (rs_adam_step takes 80 tensors per launch) instead of the
framework's chunked multi-tensor kernels; learning rate and step count are read from device memory, so a captured step
(graph.GraphedStep) follows a scheduler -- call `sync_hyper()` before each replay, GraphedStep does.
"""
import ctypes

import torch

from . import _lib, mlp_hip

P, c_int = ctypes.c_void_p, ctypes.c_int
MAX_TENSORS = 80          # RS_ADAM_MAX


class AdamTable(ctypes.Structure):           # rs_adam_table
    _fields_ = [("p", P * MAX_TENSORS), ("g", P * MAX_TENSORS), ("m", P * MAX_TENSORS), ("v", P * MAX_TENSORS),
                ("n", c_int * MAX_TENSORS), ("count", c_int)]


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("Adam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._dev = {}        # group index -> dict(hyper=device floats, host=last values, step, done)

    # ---- device-side hyper-parameters / counters ---------------------------------------------------------------
    def _group_state(self, gi, group, device):
        st = self._dev.get(gi)
        if st is None:
            st = dict(hyper=torch.zeros(10, dtype=torch.float64, device=device), host=None,      # [5..9]: the kernel's (include/repsurf_hip.h)
                      step=torch.zeros(1, dtype=torch.int32, device=device),
                      done=torch.zeros(1, dtype=torch.int32, device=device))
            self._dev[gi] = st
        return st

    @staticmethod
    def _hyper_values(group):
        return (float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                float(group["weight_decay"]))

    def sync_hyper(self):
        """push changed learning rates etc. to the device (host-side; not capturable -- call it before a replay)"""
        for gi, group in enumerate(self.param_groups):
            st = self._dev.get(gi)
            if st is None:
                continue
            vals = self._hyper_values(group)
            if st["host"] != vals:
                st["hyper"][:5].copy_(torch.tensor(vals, dtype=torch.float64))
                st["host"] = vals

    # ---- state dict in torch.optim.Adam's layout ------------------------------------------------------------------
    def state_dict(self):
        for gi, group in enumerate(self.param_groups):
            st = self._dev.get(gi)
            if st is None:
                continue
            step = float(st["step"].item())
            for p in group["params"]:
                if p in self.state:
                    self.state[p]["step"] = torch.tensor(step, dtype=torch.float32)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for gi, group in enumerate(self.param_groups):
            steps = [float(self.state[p]["step"]) for p in group["params"] if p in self.state and "step" in self.state[p]]
            if steps:
                dev = next(p.device for p in group["params"])
                self._group_state(gi, group, dev)["step"].fill_(int(steps[0]))
        self._dev_reset_host()

    def _dev_reset_host(self):
        for st in self._dev.values():
            st["host"] = None

    # ---- the update -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            if dev.type != "cuda":
                raise _lib.RepSurfHipError("repsurf_amd.optim.Adam runs on the HIP device only")
            st = self._group_state(gi, group, dev)
            if st["host"] is None:
                if torch.cuda.is_current_stream_capturing():
                    raise _lib.RepSurfHipError("repsurf_amd.optim.Adam: run one eager step (or sync_hyper()) before capture")
                self.sync_hyper()
            stream = _lib.current_stream()
            keep = []
            for c0 in range(0, len(ps), MAX_TENSORS):
                chunk = ps[c0:c0 + MAX_TENSORS]
                tab = AdamTable()
                for i, p in enumerate(chunk):
                    if p.dtype != torch.float32 or not p.is_contiguous():
                        raise _lib.RepSurfHipError("repsurf_amd.optim.Adam: parameters must be contiguous fp32")
                    s = self.state[p]
                    if "exp_avg" not in s:
                        s["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        s["step"] = torch.tensor(0.0)
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    keep.append(g)
                    tab.p[i], tab.g[i], tab.m[i], tab.v[i] = p.data_ptr(), g.data_ptr(), s["exp_avg"].data_ptr(), \
                        s["exp_avg_sq"].data_ptr()
                    tab.n[i] = p.numel()
                tab.count = len(chunk)
                last = c0 + MAX_TENSORS >= len(ps)
                _lib.call("rs_adam_step", ctypes.byref(tab), st["hyper"].data_ptr(), st["step"].data_ptr(),
                          st["done"].data_ptr(), 1 if last else 0, stream)
        mlp_hip.weights_changed()       # raw-pointer update: tensor._version does not move (mlp_hip._prepacked)
        return loss
