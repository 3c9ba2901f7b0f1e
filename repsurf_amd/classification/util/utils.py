"""util.utils — the pieces of the reference's classification/util/utils.py that sit on the timed
step: model/loss factories and the label-smoothing loss (reference :55-78)."""
import importlib
import random

import numpy as np
import torch
import torch.nn as nn

from repsurf_amd import head as _head


def set_seed(seed):
    """Seed python / numpy / torch (CPU generator drives FPS starts and normal flips)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class SmoothClsLoss(nn.Module):
    """Label-smoothed NLL over log-probabilities (reference :55-69): the target distribution puts
    1-eps on the label and eps/(C-1) elsewhere."""

    def __init__(self, smoothing_ratio=0.1):
        super().__init__()
        self.smoothing_ratio = smoothing_ratio

    def forward(self, pred, target):
        eps, n_class = self.smoothing_ratio, pred.size(1)
        if pred.is_cuda and pred.dtype == torch.float32:       # the training step's case: ONE HIP launch, always
            return _head.smooth_cls_loss(pred, target, eps)
        # Not a fallback of the device path: the loss of log-probabilities that are NOT fp32 device tensors -- float64 checks and
        # host-side evaluation scripts call the criterion on CPU tensors -- by the reference's definition (:60-68).
        soft = torch.full_like(pred, eps / (n_class - 1)).scatter_(1, target.view(-1, 1), 1 - eps)
        return -(soft * pred).sum(dim=1).mean()


def get_model(args):
    return importlib.import_module('models.%s' % args.model).Model(args)


def get_loss():
    return SmoothClsLoss()
