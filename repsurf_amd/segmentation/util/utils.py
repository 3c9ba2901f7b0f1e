"""util.utils — the pieces of the reference's segmentation/util/utils.py that call into the hot path:
the loss factory and the whole-scene kNN median filter used at test time (reference :225-245)."""
import torch
import torch.nn as nn

from repsurf_amd import ops


def get_loss(weight=None, ignore_label=None):
    return nn.CrossEntropyLoss(weight=weight, ignore_index=ignore_label)


def pc_median_filter_gpu(coord, label, group_size=16):
    """Median of the predicted labels over each point's `group_size` nearest neighbours (itself included) of one
    whole scene: coord (N,3), label (N,) -> numpy (N,)  (reference :233-245).  The kNN runs through
    repsurf_amd.ops.knn_scene: a uniform-grid search at scene scale (exactly the lists of the tiled scan), the scan itself
    for small clouds."""
    group_idx, _ = ops.knn_scene(group_size, coord)        # grid search for scenes, the tiled scan below 32 768 rows
    group_label = label[group_idx.view(-1).long()].view(coord.shape[0], group_size)
    return torch.median(group_label, 1)[0].cpu().numpy()
