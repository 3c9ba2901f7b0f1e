"""Classifier head and label-smoothing loss on the fused HIP kernels of csrc/head.hip.

`classifier_logprobs(seq, x)` runs the reference's `classfier` Sequential
(Linear-BN1d-ReLU-Dropout-Linear-BN1d-ReLU-Dropout-Linear, classification/models/repsurf/repsurf_ssg_umb.py:32-41)
followed by log_softmax (:56-57) in 3 launches forward and 4 backward, on the parameters of the nn modules themselves
(state-dict compatible).  It applies when the modules are in training mode and the batch has at most 64 rows; otherwise
`usable()` is False and the caller runs the nn.Sequential (plain library GEMMs).

Dropout: the mask is a counter-based hash, not the framework generator's stream (no implementation can reproduce that
bit for bit; parity tests disable dropout).  Seed: `torch.initial_seed()` at first use; a device-side step counter makes
every forward -- every replay of a captured graph -- draw a fresh mask.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import _lib
from . import zeros as _zeros

P, c_int, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
MAX_ROWS = 64
ENABLED = os.environ.get("REPSURF_HEAD", "hip") == "hip"
DEBUG = None          # set to a dict to receive the intermediate activations of the next forward


class HeadLayer(ctypes.Structure):           # rs_head_layer
    _fields_ = [("x", P), ("ldx", c_int), ("w", P), ("b", P), ("gamma", P), ("beta", P), ("running_mean", P),
                ("running_var", P), ("momentum", c_float), ("eps", c_float), ("drop_p", c_float), ("y", P), ("h", P),
                ("mean", P), ("invstd", P), ("seed", ctypes.c_uint), ("step", P), ("layer", c_int), ("R", c_int),
                ("K", c_int), ("N", c_int)]


class HeadLayerBwd(ctypes.Structure):        # rs_head_layer_bwd
    _fields_ = [("dz_next", P), ("n2", c_int), ("w_next", P), ("y", P), ("mean", P), ("invstd", P), ("gamma", P),
                ("beta", P), ("x", P), ("ldx", c_int), ("dz", P), ("dw", P), ("dgamma", P), ("dbeta", P),
                ("drop_p", c_float), ("seed", ctypes.c_uint), ("step", P), ("layer", c_int), ("step_back", c_int),
                ("R", c_int), ("K", c_int), ("N", c_int)]


def _stream():
    return _lib.current_stream()


_state = {}


def _step_counter(device):
    """(seed, device int32 counter) shared by every head on this device"""
    key = str(device)
    if key not in _state:
        _state[key] = (int(torch.initial_seed()) & 0xFFFFFFFF, torch.zeros(1, dtype=torch.int32, device=device))
    return _state[key]


def _parse(seq):
    """nn.Sequential(Linear, BN1d, ReLU, Dropout, Linear, BN1d, ReLU, Dropout, Linear) -> pieces, or None"""
    import torch.nn as nn
    mods = list(seq)
    bn = (nn.BatchNorm1d, nn.SyncBatchNorm)       # (convert_sync_batchnorm replaces the BatchNorm1d modules)
    kinds = [nn.Linear, bn, nn.ReLU, nn.Dropout, nn.Linear, bn, nn.ReLU, nn.Dropout, nn.Linear]
    if len(mods) != len(kinds) or not all(isinstance(m, k) for m, k in zip(mods, kinds)):
        return None
    return mods[0], mods[1], mods[3], mods[4], mods[5], mods[7], mods[8]


def usable(seq, x):
    if not (ENABLED and x.is_cuda and x.dim() == 2 and x.shape[0] <= MAX_ROWS and x.dtype == torch.float32):
        return False
    parts = _parse(seq)
    if parts is None:
        return False
    l1, bn1, d1, l2, bn2, d2, l3 = parts
    from . import mlp_hip
    if mlp_hip.sync_of(bn1) is not None or mlp_hip.sync_of(bn2) is not None:
        return False          # SyncBatchNorm: statistics over the ranks -- the row-stack route below (one all-reduce per BatchNorm)
    return (bn1.training and bn2.training and bn1.momentum is not None and bn2.momentum is not None
            and bn1.affine and bn2.affine and l1.bias is not None and l2.bias is not None and l3.bias is not None
            and l3.out_features <= 256)


def rows_usable(seq, x):
    """The head outside its fused case (eval mode, more than MAX_ROWS clouds per GPU, SyncBatchNorm): the same Sequential on the
    shared-MLP kernels -- no library GEMM on the path."""
    return bool(x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and _parse(seq) is not None)


def classifier_logprobs_rows(seq, x):
    """Linear-BN1d-ReLU-Dropout x 2, Linear, log-softmax (classification/models/repsurf/repsurf_ssg_umb.py:32-41,56) as row
    stacks of the shared-MLP kernels: each Linear + BatchNorm1d + ReLU block is a stack of groups of ONE row (row GEMM with the
    statistics in its epilogue, finalize, BN + ReLU pass; training and eval mode; SyncBatchNorm-aware), the output Linear is
    mlp.row_linear; Dropout and the log-softmax over <= 40 classes stay elementwise framework kernels."""
    import torch.nn.functional as F
    from . import mlp_hip
    l1, bn1, d1, l2, bn2, d2, l3 = _parse(seq)
    h = d1(mlp_hip.sa_mlp_plain(x, [l1], [bn1], 1))
    h = d2(mlp_hip.sa_mlp_plain(h, [l2], [bn2], 1))
    return F.log_softmax(mlp_hip.row_linear(h, l3), -1)


class _Head(Function):
    @staticmethod
    def forward(ctx, x, meta, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3):
        dev = x.device
        x = x.contiguous()
        r, k0 = x.shape
        n1, n2, nc = w1.shape[0], w2.shape[0], w3.shape[0]
        seed, step = _step_counter(dev)
        bn1, bn2 = meta["bns"]
        p1, p2 = meta["p"]
        buf1 = torch.empty((2, r, n1), dtype=torch.float32, device=dev)     # y1, h1
        buf2 = torch.empty((2, r, n2), dtype=torch.float32, device=dev)     # y2, h2
        st1 = torch.empty((2, n1), dtype=torch.float32, device=dev)         # mean, invstd
        st2 = torch.empty((2, n2), dtype=torch.float32, device=dev)
        logp = torch.empty((r, nc), dtype=torch.float32, device=dev)

        def layer(xin, w, b, g, be, bn, p, buf, st, idx):
            track = bn.track_running_stats and bn.running_mean is not None
            d = HeadLayer(x=xin.data_ptr(), ldx=xin.shape[1], w=w.data_ptr(), b=b.data_ptr(), gamma=g.data_ptr(),
                          beta=be.data_ptr(), running_mean=bn.running_mean.data_ptr() if track else None,
                          running_var=bn.running_var.data_ptr() if track else None, momentum=float(bn.momentum),
                          eps=float(bn.eps), drop_p=float(p), y=buf[0].data_ptr(), h=buf[1].data_ptr(),
                          mean=st[0].data_ptr(), invstd=st[1].data_ptr(), seed=seed, step=step.data_ptr(), layer=idx,
                          R=r, K=xin.shape[1], N=w.shape[0])
            _lib.call("rs_head_layer_forward", ctypes.byref(d), _stream())
            if track:
                meta["counters"].append(bn.num_batches_tracked)

        layer(x, w1, b1, g1, be1, bn1, p1, buf1, st1, 1)
        layer(buf1[1], w2, b2, g2, be2, bn2, p2, buf2, st2, 2)
        _lib.call("rs_head_output_forward", r, n2, nc, buf2[1].data_ptr(), w3.data_ptr(), b3.data_ptr(), logp.data_ptr(),
                  step.data_ptr(), _stream())
        if meta["counters"]:                   # joins the model's one deferred counter update when inside deferred_counters()
            from . import mlp_hip
            mlp_hip._pending_counters.extend(meta["counters"])
            meta["counters"].clear()
            mlp_hip._flush_counters()
        if DEBUG is not None:
            DEBUG.update(y1=buf1[0], h1=buf1[1], y2=buf2[0], h2=buf2[1], st1=st1, st2=st2)
        ctx.save_for_backward(x, w1, g1, be1, w2, g2, be2, w3, logp)
        ctx.bufs = (buf1, buf2, st1, st2)
        ctx.cfg = (seed, step, p1, p2)
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        x, w1, g1, be1, w2, g2, be2, w3, logp = ctx.saved_tensors
        buf1, buf2, st1, st2 = ctx.bufs
        seed, step, p1, p2 = ctx.cfg
        dev = x.device
        r, k0 = x.shape
        n1, n2, nc = w1.shape[0], w2.shape[0], w3.shape[0]
        dlogp = dlogp.contiguous()
        dlogits = torch.empty((r, nc), dtype=torch.float32, device=dev)
        dw3 = torch.empty_like(w3)
        db3 = torch.empty((nc,), dtype=torch.float32, device=dev)
        _lib.call("rs_head_output_backward", r, n2, nc, dlogp.data_ptr(), logp.data_ptr(), buf2[1].data_ptr(),
                  dlogits.data_ptr(), dw3.data_ptr(), db3.data_ptr(), _stream())

        def layer(dz_next, w_next, buf, st, g, be, xin, p, idx, w):
            n = w.shape[0]
            dz = torch.empty((r, n), dtype=torch.float32, device=dev)
            dw = torch.empty_like(w)
            dgb = torch.empty((2, n), dtype=torch.float32, device=dev)
            d = HeadLayerBwd(dz_next=dz_next.data_ptr(), n2=dz_next.shape[1], w_next=w_next.data_ptr(), y=buf[0].data_ptr(),
                             mean=st[0].data_ptr(), invstd=st[1].data_ptr(), gamma=g.data_ptr(), beta=be.data_ptr(),
                             x=xin.data_ptr(), ldx=xin.shape[1], dz=dz.data_ptr(), dw=dw.data_ptr(), dgamma=dgb[0].data_ptr(),
                             dbeta=dgb[1].data_ptr(), drop_p=float(p), seed=seed, step=step.data_ptr(), layer=idx, step_back=1,
                             R=r, K=xin.shape[1], N=n)
            _lib.call("rs_head_layer_backward", ctypes.byref(d), _stream())
            return dz, dw, dgb[0], dgb[1]

        dz2, dw2, dg2, dbe2 = layer(dlogits, w3, buf2, st2, g2, be2, buf1[1], p2, 2, w2)
        dz1, dw1, dg1, dbe1 = layer(dz2, w2, buf1, st1, g1, be1, x, p1, 1, w1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.call("rs_head_input_backward", r, n1, k0, dz1.data_ptr(), w1.data_ptr(), dx.data_ptr(), _stream())
        zeros = _zeros.take(n1 + n2, dev)     # Linear biases in front of a BatchNorm: exactly 0
        return dx, None, dw1, zeros[:n1], dg1, dbe1, dw2, zeros[n1:], dg2, dbe2, dw3, db3


def classifier_logprobs(seq, x):
    """log_softmax(seq(x)) for the reference's classfier Sequential; see `usable`."""
    l1, bn1, d1, l2, bn2, d2, l3 = _parse(seq)
    meta = {"bns": (bn1, bn2), "p": (d1.p if d1.training else 0.0, d2.p if d2.training else 0.0), "counters": []}
    return _Head.apply(x, meta, l1.weight, l1.bias, bn1.weight, bn1.bias, l2.weight, l2.bias, bn2.weight, bn2.bias,
                       l3.weight, l3.bias)


class _SmoothLoss(Function):
    @staticmethod
    def forward(ctx, logp, target, eps):
        logp = logp.contiguous()
        r, c = logp.shape
        out = torch.empty((1,), dtype=torch.float32, device=logp.device)
        dlogp = torch.empty_like(logp)
        _lib.call("rs_smooth_cls_loss", r, c, float(eps), logp.data_ptr(), target.contiguous().data_ptr(), out.data_ptr(),
                  dlogp.data_ptr(), _stream())
        ctx.save_for_backward(dlogp)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        (dlogp,) = ctx.saved_tensors
        one = _unit.get(str(dlogp.device))
        if one is not None and gout.data_ptr() == one.data_ptr():     # loss.backward(unit_gradient(device)): d loss = 1
            return dlogp, None, None
        return dlogp * gout, None, None


_unit = {}


def unit_gradient(device):
    """A persistent scalar 1.0 to pass as `loss.backward(unit_gradient(device))`: autograd then neither fills a fresh
    ones tensor nor multiplies the loss gradient by it (two launches per step)."""
    key = _lib.device_key(device)
    if key not in _unit:
        _unit[key] = torch.ones((), dtype=torch.float32, device=device)
    return _unit[key]


def smooth_cls_loss(logp, target, eps):
    """SmoothClsLoss (classification/util/utils.py:55-69) in one launch (+ one for the backward scale)."""
    return _SmoothLoss.apply(logp, target.to(torch.int64), eps)


_bad = {}


def _bad_counter(device):
    key = _lib.device_key(device)
    if key not in _bad:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            # created inside a capture, the counter would live in the graph's pool and its zero fill would be replayed: every
            # replay would reset it.  The graphed steps of this package warm up eagerly first.
            raise RuntimeError("repsurf_amd.head: the first cross-entropy call on a device must run eagerly (one warm-up pass) "
                               "before the step is captured into a hipGraph")
        _bad[key] = torch.zeros((1,), dtype=torch.int32, device=device)
    return _bad[key]


def bad_label_count(device=None):
    """Labels outside [0, classes) (and not the ignore label) the cross-entropy kernel has met on this device since the process
    started: each made its row's loss and gradient NaN.  Inside a replayed hipGraph nothing else tells the host (the NaN reaches the
    optimizer state silently), so a training loop calls this -- one 4-byte read-back -- whenever it reads its loss."""
    if device is None:
        return sum(int(t.item()) for t in _bad.values())
    return int(_bad_counter(device).item())


class _CrossEntropy(Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        logits = logits.contiguous()
        rows, classes = logits.shape
        dev = logits.device
        out = torch.empty((2,), dtype=torch.float32, device=dev)             # loss, 1 / (rows that count)
        d = torch.empty_like(logits)
        part = torch.empty((2 * ((rows + 255) // 256),), dtype=torch.float64, device=dev)
        _lib.call("rs_cross_entropy_forward", rows, classes, int(ignore_index), logits.data_ptr(), target.contiguous().data_ptr(),
                  out.data_ptr(), out.data_ptr() + 4, d.data_ptr(), part.data_ptr(), _bad_counter(dev).data_ptr(), _stream())
        ctx.save_for_backward(d, out)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        d, out = ctx.saved_tensors
        one = _unit.get(str(d.device))
        unit = one is not None and gout.data_ptr() == one.data_ptr()          # loss.backward(unit_gradient(device)): d loss = 1
        g = torch.empty_like(d)
        _lib.call("rs_scale_by_scalars", d.numel(), d.data_ptr(), out.data_ptr() + 4, None if unit else gout.contiguous().data_ptr(),
                  g.data_ptr(), _stream())
        return g, None, None


def cross_entropy(logits, target, ignore_index=-100):
    """F.cross_entropy(logits (rows, classes), target (rows,), ignore_index=...) with mean reduction -- the criterion of the
    segmentation train loop (nn.CrossEntropyLoss(ignore_index=args.ignore_label), segmentation/tool/train.py:110,296) -- in two
    launches forward and one backward (torch: log_softmax, a single-workgroup nll reduction, two backward kernels)."""
    if not (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2) or logits.shape[0] == 0:
        # not this kernel's case (other dtype / layout, or an EMPTY batch, where torch defines the result as NaN)
        return torch.nn.functional.cross_entropy(logits, target, ignore_index=ignore_index)
    return _CrossEntropy.apply(logits, target.to(torch.int64), ignore_index)


class CrossEntropyLoss(torch.nn.Module):
    """Drop-in for the train loop's `nn.CrossEntropyLoss(ignore_index=...)` (mean reduction, no class weights)."""

    def __init__(self, ignore_index=-100):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, logits, target):
        return cross_entropy(logits, target, self.ignore_index)

    @staticmethod
    def bad_labels(device=None):
        """see bad_label_count"""
        return bad_label_count(device)


def col_sum(x, scale=1.0):
    """x (rows, n) -> scale * x.sum(0) in two launches with a fixed summation order (bias gradient of a row Linear)."""
    rows, n = x.shape
    if x.stride(1) != 1:
        x = x.contiguous()
    nblk = max(1, min(256, rows // 256))
    part = torch.empty((nblk, n), dtype=torch.float32, device=x.device)
    out = torch.empty((n,), dtype=torch.float32, device=x.device)
    from . import ragged as _ragged
    rows_dev = _ragged.dev(rows)      # (a packed batch under a captured capacity: rows beyond the count are not summed)
    if rows_dev is not None:
        _lib.call("rs_col_sum_partials_dev", rows, n, x.data_ptr(), x.stride(0), float(scale), part.data_ptr(), nblk, rows_dev, _stream())
    else:
        _lib.call("rs_col_sum_partials", rows, n, x.data_ptr(), x.stride(0), float(scale), part.data_ptr(), nblk, _stream())
    _lib.call("rs_reduce_partials", nblk, n, part.data_ptr(), out.data_ptr(), _stream())
    return out

