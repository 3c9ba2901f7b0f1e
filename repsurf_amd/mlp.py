"""Grouped shared-MLP stacks (1x1 conv -> BatchNorm -> ReLU ... -> pool over the group).

Two interchangeable executors over the same nn.Conv2d / nn.BatchNorm2d parameter modules
(state-dict compatible with the reference):

  "hip"    hand-written fp32-MFMA kernels (repsurf_amd/csrc/mlp.hip) with fused BN-statistics
           epilogues, BN+ReLU prologues and the group pool folded into the last layer; the
           product path.
  "torch"  plain PyTorch fp32 ops (F.linear / F.batch_norm / relu / max): the floating-point
           REFERENCE the MFMA kernels are tested against (tests/test_mlp_gpu.py) and a debugging
           aid (`REPSURF_MLP=torch`).  Never selected implicitly.

Rows are (group, sample) pairs, channels last: x is (groups*nsample, C).
"""
import os

import torch
import torch.nn.functional as F

BACKEND = os.environ.get("REPSURF_MLP", "hip")
# Build the grouped operand for the DISTINCT ball-query slots only (padding copies of the first neighbour are
# carried as a per-row multiplicity): exact, and 3-7x fewer rows at the model's radii.  HIP executor only.
COMPACT_GROUPS = os.environ.get("REPSURF_COMPACT", "1") != "0"


# Arithmetic of the MFMA row GEMMs (forward and data gradient) of the HIP executor:
#   "fp32"  v_mfma_f32_32x32x2_f32 -- the parity path (1e-5 against the reference), the default and what bench.py reports;
#   "bf16"  BASELINE configs[4]: operands rounded to bf16 at the LDS commit, v_mfma_f32_32x32x16_bf16, fp32 accumulation
#           and fp32 tensors in HBM (rs_mlp_gemm_rows_bf16, rs_mlp_wgrad_bf16; weight-gradient slabs are summed in fp32).
#           BatchNorm, pooling, the narrow first-layer kernels, the constructor MLP and the classifier head stay fp32.  Tolerance: tests/test_mlp_gpu.py (bf16 section).
PRECISION = os.environ.get("REPSURF_MLP_DTYPE", "fp32")
if PRECISION not in ("fp32", "bf16"):
    raise ValueError(f"REPSURF_MLP_DTYPE={PRECISION!r}: expected fp32 or bf16")


def set_precision(name):
    """fp32 | bf16 for the row GEMMs of the HIP executor; read at every launch (set it before capturing a graph)."""
    global PRECISION
    if name not in ("fp32", "bf16"):
        raise ValueError(name)
    PRECISION = name


def set_backend(name):
    global BACKEND
    if name not in ("hip", "torch"):
        raise ValueError(name)
    BACKEND = name


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


# ------------------------------------------------------------------ torch executor (reference)
def _torch_sa_cd(x, pos_channel, mlp_l0, bn_l0, mlp_f0, bn_f0, convs, bns, nsample):
    loc = _bn(F.linear(x[:, :pos_channel], _w2d(mlp_l0), mlp_l0.bias), bn_l0)
    feat = _bn(F.linear(x[:, pos_channel:], _w2d(mlp_f0), mlp_f0.bias), bn_f0)
    h = F.relu(loc + feat)
    for conv, bn in zip(convs, bns):
        h = F.relu(_bn(F.linear(h, _w2d(conv), conv.bias), bn))
    return h.view(-1, nsample, h.shape[1]).max(dim=1)[0]


def _torch_sa_plain(x, convs, bns, nsample):
    h = x
    for conv, bn in zip(convs, bns):
        h = F.relu(_bn(F.linear(h, _w2d(conv), conv.bias), bn))
    return h.view(-1, nsample, h.shape[1]).max(dim=1)[0]


def _torch_umbrella(x, mlps, group, aggr):
    conv0, bn0, _, conv1, bn1, _, conv2 = mlps
    h = F.relu(_bn(F.linear(x, _w2d(conv0), conv0.bias), bn0))
    h = F.relu(_bn(F.linear(h, _w2d(conv1), conv1.bias), bn1))
    h = F.linear(h, _w2d(conv2), conv2.bias).view(-1, group, conv2.weight.shape[0])
    if aggr == "max":
        return h.max(dim=1)[0]
    if aggr == "avg":
        return h.mean(dim=1)
    return h.sum(dim=1)


# ------------------------------------------------------------------ dispatch
def sa_mlp_cd(x, pos_channel, mlp_l0, bn_l0, mlp_f0, bn_f0, convs, bns, nsample, compact=None, feat_off=None, feat_k=None):
    """SurfaceAbstractionCD body (classification/modules/repsurface_utils.py:236-244):
    relu(bn_l0(mlp_l0(x[:, :pos])) + bn_f0(mlp_f0(x[:, pos:]))) -> [conv, bn, relu]* -> max over nsample.
    x (G*nsample, pos+feat) -> (G, mlp[-1])."""
    if BACKEND == "torch":
        assert compact is None, "the torch reference executor works on dense groups"
        if feat_off is not None:       # aligned (padded) rows: back to the tight layout for the reference executor
            x = torch.cat([x[:, :pos_channel], x[:, feat_off:feat_off + feat_k]], dim=1)
        return _torch_sa_cd(x, pos_channel, mlp_l0, bn_l0, mlp_f0, bn_f0, convs, bns, nsample)
    from . import mlp_hip
    return mlp_hip.sa_mlp_cd(x, pos_channel, mlp_l0, bn_l0, mlp_f0, bn_f0, convs, bns, nsample, compact=compact,
                             feat_off=feat_off, feat_k=feat_k)


def sa_mlp_plain(x, convs, bns, nsample, relu_last=True):
    """SurfaceAbstraction body (repsurface_utils.py:178-181)."""
    if BACKEND == "torch":
        if not relu_last:
            raise NotImplementedError("relu_last=False is a HIP-executor form; the torch reference of it is bn(linear(x))")
        return _torch_sa_plain(x, convs, bns, nsample)
    from . import mlp_hip
    return mlp_hip.sa_mlp_plain(x, convs, bns, nsample, relu_last)


def umbrella_mlp(x, mlps, group, aggr):
    """UmbrellaSurfaceConstructor.mlps + aggregation (repsurface_utils.py:296-305):
    conv-bn-relu-conv-bn-relu-conv then sum/max/avg over the `group` fan triangles.
    x (P*group, C) -> (P, C)."""
    if BACKEND == "torch":
        return _torch_umbrella(x, mlps, group, aggr)
    from . import mlp_hip
    return mlp_hip.umbrella_mlp(x, mlps, group, aggr)


def _torch_umbrella2(x, mlps, group):
    conv0, bn0, _, conv1 = mlps
    h = F.relu(_bn(F.linear(x, _w2d(conv0), conv0.bias), bn0))
    return F.linear(h, _w2d(conv1), conv1.bias).view(-1, group, conv1.weight.shape[0]).sum(dim=1)


def umbrella_mlp2(x, mlps, group):
    """Segmentation UmbrellaSurfaceConstructor.mlps + aggregation
    (segmentation/modules/repsurface_utils.py:298-303,323-327): conv-bn-relu-conv, sum over the `group`
    fan triangles.  x (P*group, C) -> (P, Cout)."""
    if BACKEND == "torch":
        return _torch_umbrella2(x, mlps, group)
    from . import mlp_hip
    return mlp_hip.umbrella_mlp2(x, mlps, group)


def prepack(convs):
    """One launch that makes every weight copy the SA stacks on these 1x1 convolutions need in this step."""
    if BACKEND != "torch":
        from . import mlp_hip
        mlp_hip.prepack(convs)


def deferred_counters():
    """Context in which the BatchNorm `num_batches_tracked` updates of all stacks are batched into one launch."""
    if BACKEND == "torch":
        import contextlib
        return contextlib.nullcontext()
    from . import mlp_hip
    return mlp_hip.deferred_counters()
