"""modules.pointnet2_utils — same names and signatures as the reference
(classification/modules/pointnet2_utils.py), HIP kernels underneath.

The `cuda` flag is accepted for signature compatibility.  Both values run the HIP kernels, which
implement the *cuda=False* (CPU/PyTorch) semantics of the reference bit-exactly — random FPS
start drawn from the CPU generator, expanded-formula distances, lowest-index tie rule — because
that path is the parity oracle.  Tensors must be on a HIP device; there is no CPU path.
Index tensors are int32 (the reference's CUDA path; its CPU path returns int64).
"""
import torch

from repsurf_amd import ops, rng


def square_distance(src, dst):
    """(B,N,3),(B,M,3) -> (B,N,M) squared distances by the expanded formula (reference :15-25).
    Utility only: no kernel of the hot path materialises this matrix."""
    inner = torch.matmul(src, dst.transpose(1, 2))
    return (-2 * inner + src.pow(2).sum(-1).unsqueeze(2)) + dst.pow(2).sum(-1).unsqueeze(1)


def index_points(points, idx, cuda=False, is_group=False):
    """points (B,N,C), idx (B,S) | (B,S,K) -> (B,S,C) | (B,S,K,C)   (reference :28-44)."""
    return ops.gather_rows(points, idx)


def draw_fps_start(batch, n):
    """The first FPS pick, drawn exactly like the reference CPU path does (reference :66):
    torch.randint on the CPU default generator."""
    return torch.randint(0, n, (batch,), dtype=torch.long)


def farthest_point_sample(xyz, npoint, cuda=False, start=None):
    """xyz (B,N,3) -> (B,npoint) int32 sampled indices (reference :47-75).
    `start` (optional, (B,) ints) overrides the random first pick."""
    if start is None:
        start = rng.draw("fps", xyz.shape[0], xyz.shape[1], xyz.device)
    else:
        start = start.to(device=xyz.device, dtype=torch.int32, non_blocking=True)
    return ops.furthestsampling(xyz, npoint, start)


def query_ball_point(radius, nsample, xyz, new_xyz, debug=False, cuda=False):
    """-> (B,S,nsample) int32 neighbour lists (reference :78-99)."""
    idx = ops.ballquery(radius, nsample, xyz, new_xyz)
    if debug:
        d = square_distance(new_xyz, xyz)
        inside = (~(d > radius ** 2)).sum(-1)
        return (nsample - inside).clamp(min=0).sum(), (inside - nsample).clamp(min=0).sum()
    return idx


def query_knn_point(k, xyz, new_xyz, cuda=False):
    """-> (B,S,k) int32, ascending by (distance, index) (reference :102-111)."""
    return ops.knnquery(k, xyz, new_xyz)


def sample(nsample, feature, cuda=False):
    """feature (B,C,N) with xyz in the first 3 channels -> FPS-subsampled (B,C,nsample)
    (reference :114-124; the train loop's pre-model down-sampling)."""
    rows = feature.permute(0, 2, 1).contiguous()
    fps_idx = farthest_point_sample(rows[:, :, :3].contiguous(), nsample)
    return index_points(rows, fps_idx).permute(0, 2, 1)
