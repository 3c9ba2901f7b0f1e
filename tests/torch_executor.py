"""Plain PyTorch fp32 executor of the grouped shared-MLP stacks (F.linear / F.batch_norm / relu / max): the
floating-point REFERENCE the MFMA kernels are tested against (tests/test_mlp_gpu.py, tests/test_model_gpu.py).
TEST INFRASTRUCTURE ONLY: it lives here, not in repsurf_amd/, so that the product package has ONE executor.

`set_backend("torch")` swaps repsurf_amd.mlp's dispatch functions for the ones below (and switches the compacted
groups off: the reference executor works on dense groups); `set_backend("hip")` restores the product functions."""
import contextlib

import torch
import torch.nn.functional as F

from repsurf_amd import mlp as _mlp

_PRODUCT = {n: getattr(_mlp, n) for n in ("sa_mlp_cd", "sa_mlp_plain", "umbrella_mlp", "umbrella_mlp2", "fp_front", "fp_front_usable", "lazy_rows_usable", "prepack",
                                          "deferred_counters")}
_PRODUCT_COMPACT = _mlp.COMPACT_GROUPS
BACKEND = "hip"


def _w2d(conv):
    w = conv.weight
    return w.view(w.shape[0], w.shape[1])


def _bn(y, bn):
    """BatchNorm over rows (training: batch statistics + running-stat update, like nn.BatchNorm2d
    over (B,C,nsample,npoint))."""
    if bn.training and bn.track_running_stats:
        bn.num_batches_tracked.add_(1)
    return F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                        bn.training or not bn.track_running_stats, bn.momentum, bn.eps)


def sa_mlp_cd(x, pos_channel, mlp_l0, bn_l0, mlp_f0, bn_f0, convs, bns, nsample, compact=None, feat_off=None, feat_k=None):
    assert compact is None, "the torch reference executor works on dense groups"
    if feat_off is not None:       # aligned (padded) rows: back to the tight layout
        x = torch.cat([x[:, :pos_channel], x[:, feat_off:feat_off + feat_k]], dim=1)
    loc = _bn(F.linear(x[:, :pos_channel], _w2d(mlp_l0), mlp_l0.bias), bn_l0)
    feat = _bn(F.linear(x[:, pos_channel:], _w2d(mlp_f0), mlp_f0.bias), bn_f0)
    h = F.relu(loc + feat)
    for conv, bn in zip(convs, bns):
        h = F.relu(_bn(F.linear(h, _w2d(conv), conv.bias), bn))
    return h.view(-1, nsample, h.shape[1]).max(dim=1)[0]


def sa_mlp_plain(x, convs, bns, nsample, relu_last=True, lazy_out=False):
    h = x
    for i, (conv, bn) in enumerate(zip(convs, bns)):
        h = _bn(F.linear(h, _w2d(conv), conv.bias), bn)
        if relu_last or i + 1 < len(convs):
            h = F.relu(h)
    return h.view(-1, nsample, h.shape[1]).max(dim=1)[0]


def umbrella_mlp(x, mlps, group, aggr, moments=None):
    conv0, bn0, _, conv1, bn1, _, conv2 = mlps
    h = F.relu(_bn(F.linear(x, _w2d(conv0), conv0.bias), bn0))
    h = F.relu(_bn(F.linear(h, _w2d(conv1), conv1.bias), bn1))
    h = F.linear(h, _w2d(conv2), conv2.bias).view(-1, group, conv2.weight.shape[0])
    if aggr == "max":
        return h.max(dim=1)[0]
    if aggr == "avg":
        return h.mean(dim=1)
    return h.sum(dim=1)


def umbrella_mlp2(x, mlps, group, moments=None):
    conv0, bn0, _, conv1 = mlps
    h = F.relu(_bn(F.linear(x, _w2d(conv0), conv0.bias), bn0))
    return F.linear(h, _w2d(conv1), conv1.bias).view(-1, group, conv1.weight.shape[0]).sum(dim=1)


def fp_front_usable(lin_f, bn_f, lin_s, bn_s):
    return True


def fp_front(points2, points1, idx, weight, lin_f, bn_f, lin_s, bn_s, csr=None):
    z2 = _bn(F.linear(points2, lin_f.weight, lin_f.bias), bn_f)
    z1 = _bn(F.linear(points1, lin_s.weight, lin_s.bias), bn_s)
    return F.relu((z2[idx.long()] * weight.unsqueeze(-1)).sum(1) + z1)


# ---- bf16 mode restated (BASELINE configs[4]): the SA stacks with every rounding the HIP kernels of mlp.set_precision("bf16") apply, at the
# same points -- forward: both GEMM operands rounded to bf16 AFTER their fp32 prologue (BatchNorm + ReLU of the previous layer), fp32
# accumulation, the conv output rounded to bf16 for storage when every width of the stack is a multiple of 4 (the BatchNorm statistics
# and the max-pool see the rounded values); backward: the incoming gradient (= the BatchNorm-backward affine, fp32) rounded to bf16 as
# the P operand of BOTH the data gradient (P . round(W)) and the weight gradient (round(P)^T . round(Q)), everything else fp32
# (repsurf_amd/mlp_hip.py, "bf16 mode"; csrc/mlp.hip "storage roles").  bf16 x bf16 products are exact in fp32, so this executor and
# the MFMA kernels differ by summation order only: the model-level bf16 claim becomes "equals the rounded-operand network".
def _r(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _LinearBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, store):
        xr, wr = _r(x), _r(w)
        y = xr @ wr.t()
        if b is not None:
            y = y + b
        ctx.save_for_backward(xr, wr)
        ctx.has_b = b is not None
        return _r(y) if store else y

    @staticmethod
    def backward(ctx, g):
        xr, wr = ctx.saved_tensors
        gr = _r(g)
        return gr @ wr, gr.t() @ xr, (g.sum(0) if ctx.has_b else None), None


def _stores_bf16(widths):
    return all(int(c) % 4 == 0 for c in widths)


def sa_mlp_cd_bf16(x, pos_channel, mlp_l0, bn_l0, mlp_f0, bn_f0, convs, bns, nsample, compact=None, feat_off=None, feat_k=None):
    assert compact is None, "the restated executor works on dense groups"
    if feat_off is not None:
        x = torch.cat([x[:, :pos_channel], x[:, feat_off:feat_off + feat_k]], dim=1)
    store = _stores_bf16([mlp_l0.weight.shape[0], mlp_f0.weight.shape[0]] + [c.weight.shape[0] for c in convs])
    loc = _bn(_LinearBF16.apply(x[:, :pos_channel], _w2d(mlp_l0), mlp_l0.bias, store), bn_l0)
    feat = _bn(_LinearBF16.apply(x[:, pos_channel:], _w2d(mlp_f0), mlp_f0.bias, store), bn_f0)
    h = F.relu(loc + feat)
    for conv, bn in zip(convs, bns):
        h = F.relu(_bn(_LinearBF16.apply(h, _w2d(conv), conv.bias, store), bn))
    return h.view(-1, nsample, h.shape[1]).max(dim=1)[0]


def sa_mlp_plain_bf16(x, convs, bns, nsample, relu_last=True, lazy_out=False):
    store = _stores_bf16([c.weight.shape[0] for c in convs])
    h = x
    for i, (conv, bn) in enumerate(zip(convs, bns)):
        h = _bn(_LinearBF16.apply(h, _w2d(conv), conv.bias, store), bn)
        if relu_last or i + 1 < len(convs):
            h = F.relu(h)
    return h.view(-1, nsample, h.shape[1]).max(dim=1)[0]


def set_backend(name):
    global BACKEND
    if name not in ("hip", "torch", "torch_bf16"):
        raise ValueError(name)
    BACKEND = name
    if name == "hip":
        for n, f in _PRODUCT.items():
            setattr(_mlp, n, f)
        _mlp.COMPACT_GROUPS = _PRODUCT_COMPACT
    else:
        _mlp.sa_mlp_cd, _mlp.sa_mlp_plain = (sa_mlp_cd, sa_mlp_plain) if name == "torch" else (sa_mlp_cd_bf16, sa_mlp_plain_bf16)
        _mlp.umbrella_mlp, _mlp.umbrella_mlp2 = umbrella_mlp, umbrella_mlp2
        _mlp.fp_front, _mlp.fp_front_usable = fp_front, fp_front_usable
        _mlp.lazy_rows_usable = lambda bn_mods: False
        _mlp.prepack = lambda convs: None
        _mlp.deferred_counters = contextlib.nullcontext
        _mlp.COMPACT_GROUPS = False
