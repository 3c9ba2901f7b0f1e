"""Fused classifier head + label-smoothing loss (csrc/head.hip) against the nn modules they replace."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def make_head(c0=1024, classes=15, p=0.4, seed=0):
    torch.manual_seed(seed)
    seq = nn.Sequential(nn.Linear(c0, 512), nn.BatchNorm1d(512), nn.ReLU(True), nn.Dropout(p),
                        nn.Linear(512, 256), nn.BatchNorm1d(256), nn.ReLU(True), nn.Dropout(p),
                        nn.Linear(256, classes)).cuda().train()
    with torch.no_grad():
        for m in seq:
            if isinstance(m, nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
    return seq


def make_odd_head(c0, n1, n2, classes, seed=3):
    torch.manual_seed(seed)
    seq = nn.Sequential(nn.Linear(c0, n1), nn.BatchNorm1d(n1), nn.ReLU(True), nn.Dropout(0.0),
                        nn.Linear(n1, n2), nn.BatchNorm1d(n2), nn.ReLU(True), nn.Dropout(0.0),
                        nn.Linear(n2, classes)).cuda().train()
    with torch.no_grad():
        for m in seq:
            if isinstance(m, nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
    return seq


@pytest.mark.parametrize("rows,c0,n1,n2,classes", [(7, 70, 48, 40, 7), (33, 100, 96, 33, 3), (64, 36, 64, 72, 40)])
def test_head_with_ragged_sizes(rows, c0, n1, n2, classes):
    """Widths that are not multiples of 32 / 4: the partial-chunk and unaligned paths of the staged layer kernels (the aligned
    whole-chunk path is what the classifier's own sizes take, test above)."""
    import copy
    from repsurf_amd import head
    ref = make_odd_head(c0, n1, n2, classes)
    mine = copy.deepcopy(ref)
    x = torch.randn(rows, c0, device="cuda")
    w = torch.randn(rows, classes, device="cuda")
    xr, xm = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    assert head.usable(mine, xm)
    lp_ref = F.log_softmax(ref(xr), -1)
    (lp_ref * w).sum().backward()
    lp = head.classifier_logprobs(mine, xm)
    (lp * w).sum().backward()
    assert (lp - lp_ref).abs().max().item() <= 2e-5
    assert (xm.grad - xr.grad).abs().max().item() <= 2e-5 * max(1.0, xr.grad.abs().max().item())
    for (n, pr), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        if n in ("0.bias", "4.bias"):
            continue
        rel = (pm.grad - pr.grad).norm().item() / max(pr.grad.norm().item(), 1e-12)
        assert rel <= 5e-4, (n, rel)


@pytest.mark.parametrize("rows", [32, 5, 64])
def test_head_matches_modules_without_dropout(rows):
    import copy
    from repsurf_amd import head
    ref = make_head(p=0.0)
    mine = copy.deepcopy(ref)
    x = torch.randn(rows, 1024, device="cuda")
    label = torch.randint(0, 15, (rows,), device="cuda")
    xr = x.clone().requires_grad_(True)
    xm = x.clone().requires_grad_(True)
    assert head.usable(mine, xm)
    lp_ref = F.log_softmax(ref(xr), -1)
    soft = torch.full_like(lp_ref, 0.1 / 14).scatter_(1, label.view(-1, 1), 0.9)
    loss_ref = -(soft * lp_ref).sum(1).mean()
    loss_ref.backward()
    lp = head.classifier_logprobs(mine, xm)
    loss = head.smooth_cls_loss(lp, label, 0.1)
    loss.backward()
    assert (lp - lp_ref).abs().max().item() <= 2e-5
    assert abs(loss.item() - loss_ref.item()) <= 1e-5
    assert (xm.grad - xr.grad).abs().max().item() <= 1e-5 * max(1.0, xr.grad.abs().max().item()) + 2e-6
    for (n, pr), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        if n in ("0.bias", "4.bias"):                      # in front of a BatchNorm: analytically zero
            assert pm.grad.abs().max().item() == 0.0
            continue
        rel = (pm.grad - pr.grad).norm().item() / max(pr.grad.norm().item(), 1e-12)
        assert rel <= 2e-4, (n, rel)
    for (n, br), (_, bm) in zip(ref.named_buffers(), mine.named_buffers()):
        assert torch.allclose(br.float(), bm.float(), rtol=1e-5, atol=1e-6), n


def test_head_dropout_is_consistent_and_fresh():
    """p = 0.4: the backward uses the forward's mask (checked against autograd on the same mask), about 40 % of the
    positive activations are dropped, and two forwards draw different masks."""
    from repsurf_amd import head
    seq = make_head(p=0.4)
    x = torch.randn(32, 1024, device="cuda", requires_grad=True)
    label = torch.randint(0, 15, (32,), device="cuda")
    dbg = {}
    head.DEBUG = dbg
    lp = head.classifier_logprobs(seq, x)
    head.DEBUG = None
    loss = head.smooth_cls_loss(lp, label, 0.1)
    loss.backward()
    got = {n: p.grad.clone() for n, p in seq.named_parameters()}
    gx = x.grad.clone()
    h1, h2 = dbg["h1"].clone(), dbg["h2"].clone()
    # torch replica with the masks read off the stored activations
    l1, bn1, _, _, l2, bn2, _, _, l3 = seq
    for p in seq.parameters():
        p.grad = None
    xr = x.detach().clone().requires_grad_(True)
    a1 = F.relu(F.batch_norm(l1(xr), None, None, bn1.weight, bn1.bias, True, 0.0, bn1.eps))
    m1 = torch.where(a1 > 0, h1 / a1.detach().clamp_min(1e-30), torch.zeros_like(a1))
    a2 = F.relu(F.batch_norm(l2(a1 * m1), None, None, bn2.weight, bn2.bias, True, 0.0, bn2.eps))
    m2 = torch.where(a2 > 0, h2 / a2.detach().clamp_min(1e-30), torch.zeros_like(a2))
    lp_ref = F.log_softmax(l3(a2 * m2), -1)
    soft = torch.full_like(lp_ref, 0.1 / 14).scatter_(1, label.view(-1, 1), 0.9)
    (-(soft * lp_ref).sum(1).mean()).backward()
    assert (lp - lp_ref).abs().max().item() <= 5e-5
    for n, p in seq.named_parameters():
        if n in ("0.bias", "4.bias"):
            continue
        rel = (got[n] - p.grad).norm().item() / max(p.grad.norm().item(), 1e-12)
        assert rel <= 5e-4, (n, rel)
    assert (gx - xr.grad).norm().item() / xr.grad.norm().item() <= 5e-4
    keep1 = ((m1 > 0) & (a1 > 0)).float().sum() / (a1 > 0).float().sum()
    assert 0.55 <= keep1.item() <= 0.65
    big = (m1 > 0) & (a1 > 1e-3)
    assert torch.allclose(m1[big], torch.full_like(m1[big], 1 / 0.6), rtol=1e-3)
    dbg2 = {}
    head.DEBUG = dbg2
    head.classifier_logprobs(seq, x.detach())
    head.DEBUG = None
    assert ((dbg2["h1"] > 0) != (h1 > 0)).float().mean().item() > 0.1       # a new mask on the next forward


def test_smooth_loss_matches_reference_formula():
    from repsurf_amd import head
    lp = F.log_softmax(torch.randn(48, 40, device="cuda"), -1).requires_grad_(True)
    t = torch.randint(0, 40, (48,), device="cuda")
    loss = head.smooth_cls_loss(lp, t, 0.1)
    loss.backward()
    lr = lp.detach().clone().requires_grad_(True)
    soft = torch.full_like(lr, 0.1 / 39).scatter_(1, t.view(-1, 1), 0.9)
    ref = -(soft * lr).sum(1).mean()
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-6 and (lp.grad - lr.grad).abs().max().item() <= 1e-8
