"""Grouped shared-MLP stacks (1x1 conv -> BatchNorm -> ReLU ... -> pool over the group) over the reference's
nn.Conv2d / nn.BatchNorm2d parameter modules (state-dict compatible with the reference), executed by the hand-written
MFMA kernels (repsurf_amd/csrc/mlp.hip, repsurf_amd/mlp_hip.py): fused BN-statistics epilogues, BN+ReLU prologues and
the group pool folded into the last layer.  There is no other executor in this package; the plain-PyTorch fp32
reference the kernels are tested against lives in tests/torch_executor.py.

Rows are (group, sample) pairs, channels last: x is (groups*nsample, C).

This module is deliberately thin: it is the SEAM between the reference-named modules and the executor -- the two run-time
switches (compacted groups, fp32 / bf16 arithmetic) live here, and the parity tests swap the stack functions below for the
plain-PyTorch executor of tests/torch_executor.py (`set_backend`) to compare the two on identical modules.
"""
import os

from . import mlp_hip

# Build the grouped operand for the DISTINCT ball-query slots only (padding copies of the first neighbour are
# carried as a per-row multiplicity): exact, and 3-7x fewer rows at the model's radii.
COMPACT_GROUPS = os.environ.get("REPSURF_COMPACT", "1") != "0"


# Arithmetic of the MFMA row GEMMs (forward and data gradient) and weight gradients:
#   "fp32"  fp32 operands, fp32 results within 1e-5 of the reference -- the parity path, the default and what bench.py reports.
#           HOW the fp32 products are formed is a property of the library, read once per process (RS_GEMM_SPLIT3,
#           include/repsurf_hip.h: rs_mlp_gemm_split3): 1 (default) = each operand split into three bf16 parts, six
#           v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation -- as close to the fp64 product as the fp32 MFMA
#           (tests/test_mlp_gpu.py::test_gemm_products_are_fp32_accurate); 0 = v_mfma_f32_32x32x2_f32.  bench.py names the
#           one that ran in its `arithmetic` key.
#   "bf16"  BASELINE configs[4]: operands rounded to bf16 at the LDS commit, v_mfma_f32_32x32x16_bf16, fp32 accumulation
#           (rs_mlp_gemm_rows_bf16, rs_mlp_wgrad_bf16; weight-gradient slabs are summed in fp32).
#           BatchNorm, pooling, the narrow first-layer kernels, the constructor MLP and the classifier head stay fp32.  Tolerance: tests/test_mlp_gpu.py (bf16 section).
PRECISION = os.environ.get("REPSURF_MLP_DTYPE", "fp32")
if PRECISION not in ("fp32", "bf16"):
    raise ValueError(f"REPSURF_MLP_DTYPE={PRECISION!r}: expected fp32 or bf16")


def set_precision(name):
    """fp32 | bf16 for the row GEMMs; read at every launch (set it before capturing a graph)."""
    global PRECISION
    if name not in ("fp32", "bf16"):
        raise ValueError(name)
    PRECISION = name


def sa_mlp_cd(x, pos_channel, mlp_l0, bn_l0, mlp_f0, bn_f0, convs, bns, nsample, compact=None, feat_off=None, feat_k=None):
    """SurfaceAbstractionCD body (classification/modules/repsurface_utils.py:236-244):
    relu(bn_l0(mlp_l0(x[:, :pos])) + bn_f0(mlp_f0(x[:, pos:]))) -> [conv, bn, relu]* -> max over nsample.
    x (G*nsample, pos+feat) -> (G, mlp[-1])."""
    return mlp_hip.sa_mlp_cd(x, pos_channel, mlp_l0, bn_l0, mlp_f0, bn_f0, convs, bns, nsample, compact=compact,
                             feat_off=feat_off, feat_k=feat_k)


def sa_mlp_plain(x, convs, bns, nsample, relu_last=True, lazy_out=False):
    """SurfaceAbstraction body (repsurface_utils.py:178-181).  lazy_out / a LazyRows input (nsample = 1): the last BatchNorm +
    ReLU is left to the consumer's operand prologue (mlp_hip.LazyRows)."""
    return mlp_hip.sa_mlp_plain(x, convs, bns, nsample, relu_last, lazy_out=lazy_out)


def lazy_rows_usable(bn_mods):
    """May the row stacks over these BatchNorms hand their last activation over unmaterialised (training, batch statistics, fp32)?"""
    return mlp_hip.lazy_rows_usable(bn_mods)


def umbrella_mlp(x, mlps, group, aggr, moments=None):
    """UmbrellaSurfaceConstructor.mlps + aggregation (repsurface_utils.py:296-305):
    conv-bn-relu-conv-bn-relu-conv then sum/max/avg over the `group` fan triangles.
    x (P*group, C) -> (P, C).  moments: `umbrella_moments(x)` computed ahead of time (geometry stage), optional."""
    return mlp_hip.umbrella_mlp(x, mlps, group, aggr, moments)


def umbrella_mlp2(x, mlps, group, moments=None):
    """Segmentation UmbrellaSurfaceConstructor.mlps + aggregation
    (segmentation/modules/repsurface_utils.py:298-303,323-327): conv-bn-relu-conv, sum over the `group`
    fan triangles.  x (P*group, C) -> (P, Cout).  moments: as umbrella_mlp."""
    return mlp_hip.umbrella_mlp2(x, mlps, group, moments)


def umbrella_moments_wanted(layers):
    """Does the constructor MLP with this many conv layers (3: classification, 2: segmentation) run on the path that reads the
    moments?  (The geometry stage computes them only then.)"""
    return bool(mlp_hip.FUSED_UMBRELLA and mlp_hip.UMB_MFMA and (layers == 2 or mlp_hip.UMB_MFMA_FWD3))


def umbrella_moments(x):
    """First and second moments of the (rows, 10) constructor features, (11, 16) fp64 (csrc/umbrella_mfma.hip): what BatchNorm 0
    of the constructor MLP and the linear part of its first weight gradient are computed from.  Geometry-only."""
    return mlp_hip.umbrella_moments(x)


def fp_front_usable(lin_f, bn_f, lin_s, bn_s):
    """Can `fp_front` serve these layers (training mode, batch statistics, fp32, <= 256 channels)?"""
    return mlp_hip.fp_front_usable(lin_f, bn_f, lin_s, bn_s)


def fp_front(points2, points1, idx, weight, lin_f, bn_f, lin_s, bn_s, csr=None):
    """Feature propagation in front of its [Linear, BN, ReLU]* chain (segmentation/modules/repsurface_utils.py:256-270) as one
    node: relu(interpolate(bn_f(lin_f(points2)), idx, weight) + bn_s(lin_s(points1)))."""
    return mlp_hip.fp_front(points2, points1, idx, weight, lin_f, bn_f, lin_s, bn_s, csr=csr)


def row_linear(x, linear):
    """A plain nn.Linear on ungrouped rows (no BatchNorm): y = x . W^T + b, forward and backward on the row GEMM /
    weight-gradient kernels (the segmentation classifier's 13-class output layer)."""
    return mlp_hip.row_linear(x, linear)


def prepack(convs):
    """One launch that makes every weight copy the SA stacks on these 1x1 convolutions need in this step."""
    mlp_hip.prepack(convs)


def deferred_counters():
    """Context in which the BatchNorm `num_batches_tracked` updates of all stacks are batched into one launch."""
    return mlp_hip.deferred_counters()
