"""Row counts as device data (repsurf_amd.ragged, `rows_dev` of include/repsurf_hip.h): every building block of the segmentation network,
run on capacity-sized tensors whose rows beyond the batch's count hold NaN, gives the results of the same block on the truncated
tensors -- outputs, input gradients, parameter gradients, running statistics.  NaN padding makes any reduction that reads a row it
must not read visible.  (What lets ONE captured graph serve the reference's ragged packed batches: segmentation/util/data_util.py:15-23.)"""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

from tests import torch_executor

pytestmark = pytest.mark.gpu

NAN = float("nan")


def pad_rows(t, cap, fill=NAN):
    out = torch.full((cap,) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=t.device)
    out[:t.shape[0]] = t
    return out


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def grads_of(mods):
    return {f"{i}.{n}": (None if p.grad is None else p.grad.clone()) for i, m in enumerate(mods) for n, p in m.named_parameters()}


def check(ref, got, n, what, tol=1e-5):
    (o0, g0, p0, s0), (o1, g1, p1, s1) = ref, got
    assert torch.isfinite(o1[:n]).all(), what
    assert rel_l2(o1[:n], o0) <= tol, (what, "output", rel_l2(o1[:n], o0))
    for i, (a, b) in enumerate(zip(g0, g1)):
        if a is None:
            continue
        assert torch.isfinite(b[:a.shape[0]]).all(), (what, "input gradient", i)
        assert rel_l2(b[:a.shape[0]], a) <= 50 * tol, (what, "input gradient", i, rel_l2(b[:a.shape[0]], a))
    for k, a in p0.items():
        if a is None or float(a.norm()) < 1e-6:
            continue
        assert torch.isfinite(p1[k]).all(), (what, k)
        assert rel_l2(p1[k], a) <= 50 * tol, (what, k, rel_l2(p1[k], a))
    for a, b in zip(s0, s1):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), (what, "running statistics")


def bn_init(bns):
    for bn in bns:
        nn.init.uniform_(bn.weight, 0.5, 1.5)
        nn.init.uniform_(bn.bias, -0.3, 0.3)


@pytest.mark.parametrize("n,cap,cin,widths,relu_last", [(3000, 4096, 64, [128, 64], True), (777, 1024, 256, [256], False), (40, 64, 128, [128], True)])
def test_row_stack_under_a_capacity(n, cap, cin, widths, relu_last):
    from repsurf_amd import mlp, ragged
    torch_executor.set_backend("hip")
    torch.manual_seed(n)
    lins = nn.ModuleList([nn.Linear(a, b) for a, b in zip([cin] + widths[:-1], widths)]).cuda()
    bns = nn.ModuleList([nn.BatchNorm1d(b) for b in widths]).cuda().train()
    bn_init(bns)
    x0, w0 = torch.randn(n, cin).cuda(), torch.randn(n, widths[-1]).cuda()
    res = []
    for ragged_run in (False, True):
        l, b = copy.deepcopy(lins), copy.deepcopy(bns)
        x = (pad_rows(x0, cap) if ragged_run else x0.clone()).requires_grad_()
        w = pad_rows(w0, cap) if ragged_run else w0
        if ragged_run:
            capy = ragged.Capacity([cap], 32, 9, x.device)
            capy.fill([n])
            with capy:
                out = mlp.sa_mlp_plain(x, l, b, 1, relu_last)
                out.backward(w)
        else:
            out = mlp.sa_mlp_plain(x, l, b, 1, relu_last)
            out.backward(w)
        torch.cuda.synchronize()
        res.append((out.detach(), [x.grad], grads_of([l, b]), [bn.running_var.clone() for bn in b] + [bn.running_mean.clone() for bn in b]))
    check(res[0], res[1], n, "row stack")


@pytest.mark.parametrize("groups,capg,pos,feat,widths", [(100, 128, 3, 13, [32, 32, 64]), (11, 16, 3, 266, [256, 256, 512]), (53, 64, 6, 138, [128, 128, 256])])
def test_grouped_cd_stack_under_a_capacity(groups, capg, pos, feat, widths):
    """groups of 32 rows: the fused max-pool of the row GEMM's epilogue under a device row count"""
    from repsurf_amd import mlp, ragged
    from tests.test_mlp_gpu import make_cd
    torch_executor.set_backend("hip")
    ns = 32
    mod0 = make_cd(pos, feat, widths, groups)
    torch.manual_seed(groups)
    x0, w0 = torch.randn(groups * ns, pos + feat).cuda(), torch.randn(groups, widths[-1]).cuda()
    res = []
    for ragged_run in (False, True):
        mod = copy.deepcopy(mod0)
        x = (pad_rows(x0, capg * ns) if ragged_run else x0.clone()).requires_grad_()
        w = pad_rows(w0, capg) if ragged_run else w0

        def run():
            out = mlp.sa_mlp_cd(x, pos, mod.mlp_l0, mod.bn_l0, mod.mlp_f0, mod.bn_f0, mod.convs, mod.bns, ns)
            out.backward(w)
            return out
        if ragged_run:
            capy = ragged.Capacity([capg * 4 * 256, capg], ns, 9, x.device)
            capy.fill([0, groups])
            with capy:
                out = run()
        else:
            out = run()
        torch.cuda.synchronize()
        bns = [mod.bn_l0, mod.bn_f0] + list(mod.bns)
        res.append((out.detach(), [x.grad[:, pos:]], grads_of([mod]), [bn.running_var.clone() for bn in bns]))
    check(res[0], res[1], groups, "grouped stack")


@pytest.mark.parametrize("n,capn,m,capm,c2,c1,c", [(700, 1024, 170, 256, 256, 128, 256), (53, 64, 13, 16, 512, 256, 256)])
def test_feature_propagation_front_under_a_capacity(n, capn, m, capm, c2, c1, c):
    from repsurf_amd import mlp, ragged
    torch_executor.set_backend("hip")
    torch.manual_seed(n)
    lf, ls = nn.Linear(c2, c).cuda(), nn.Linear(c1, c).cuda()
    bf, bs = nn.BatchNorm1d(c).cuda().train(), nn.BatchNorm1d(c).cuda().train()
    bn_init([bf, bs])
    p2, p1, w0 = torch.randn(m, c2).cuda(), torch.randn(n, c1).cuda(), torch.randn(n, c).cuda()
    idx0 = torch.randint(0, m, (n, 3), dtype=torch.int32).cuda()
    wt0 = torch.rand(n, 3).cuda()
    wt0 = wt0 / wt0.sum(1, keepdim=True)
    res = []
    for ragged_run in (False, True):
        mods = [copy.deepcopy(t) for t in (lf, bf, ls, bs)]
        a = (pad_rows(p2, capm) if ragged_run else p2.clone()).requires_grad_()
        b = (pad_rows(p1, capn) if ragged_run else p1.clone()).requires_grad_()
        idx = pad_rows(idx0, capn, 0) if ragged_run else idx0
        wt = pad_rows(wt0, capn) if ragged_run else wt0
        w = pad_rows(w0, capn) if ragged_run else w0

        def run():
            out = mlp.fp_front(a, b, idx, wt, mods[0], mods[1], mods[2], mods[3])
            out.backward(w)
            return out
        if ragged_run:
            capy = ragged.Capacity([capn, capm], 32, 9, a.device)
            capy.fill([n, m])
            with capy:
                out = run()
        else:
            out = run()
        torch.cuda.synchronize()
        res.append((out.detach(), [a.grad, b.grad], grads_of(mods), [mods[1].running_var.clone(), mods[3].running_var.clone()]))
    check(res[0], res[1], n, "feature propagation front")


@pytest.mark.parametrize("n,cap", [(3000, 4096), (100, 256)])
def test_seg_constructor_mlp_under_a_capacity(n, cap):
    from repsurf_amd import mlp, ragged
    torch_executor.set_backend("hip")
    torch.manual_seed(n)
    k = 9
    mlps = nn.Sequential(nn.Conv1d(10, 10, 1), nn.BatchNorm1d(10), nn.ReLU(True), nn.Conv1d(10, 10, 1)).cuda().train()
    bn_init([mlps[1]])
    x0, w0 = torch.randn(n * k, 10).cuda(), torch.randn(n, 10).cuda()
    res = []
    for ragged_run in (False, True):
        mm = copy.deepcopy(mlps)
        x = pad_rows(x0, cap * k) if ragged_run else x0
        w = pad_rows(w0, cap) if ragged_run else w0
        mom = mlp.umbrella_moments(x0) if mlp.umbrella_moments_wanted(2) else None      # geometry: always over the batch's rows
        if ragged_run:
            capy = ragged.Capacity([cap], 32, k, x.device)
            capy.fill([n])
            with capy:
                out = mlp.umbrella_mlp2(x, mm, k, moments=mom)
                out.backward(w)
        else:
            out = mlp.umbrella_mlp2(x, mm, k, moments=mom)
            out.backward(w)
        torch.cuda.synchronize()
        res.append((out.detach(), [], grads_of([mm]), [mm[1].running_var.clone(), mm[1].running_mean.clone()]))
    check(res[0], res[1], n, "constructor MLP")


def test_output_linear_and_cross_entropy_under_a_capacity():
    from repsurf_amd import mlp, ragged
    from repsurf_amd.head import CrossEntropyLoss
    torch_executor.set_backend("hip")
    torch.manual_seed(5)
    n, cap = 3000, 4096
    lin = nn.Linear(128, 13).cuda()
    x0 = torch.randn(n, 128).cuda()
    lab0 = torch.randint(0, 13, (n,)).cuda()
    lab0[::17] = 255
    crit = CrossEntropyLoss(ignore_index=255)
    res = []
    for ragged_run in (False, True):
        ll = copy.deepcopy(lin)
        x = (pad_rows(x0, cap) if ragged_run else x0.clone()).requires_grad_()
        lab = pad_rows(lab0, cap, 255) if ragged_run else lab0
        if ragged_run:
            capy = ragged.Capacity([cap], 32, 9, x.device)
            capy.fill([n])
            with capy:
                loss = crit(mlp.row_linear(x, ll), lab)
                loss.backward()
        else:
            loss = crit(mlp.row_linear(x, ll), lab)
            loss.backward()
        torch.cuda.synchronize()
        res.append((loss.detach().reshape(1), [x.grad], grads_of([ll]), []))
    check(res[0], res[1], 1, "output layer + loss")


def test_gather_and_interpolation_backward_under_a_capacity():
    """the scatter-add backward passes stop at the device count (rows beyond it hold NaN gradients and index 0)"""
    from repsurf_amd import ops, ragged
    torch.manual_seed(3)
    n, capn, m, capm, c = 700, 1024, 170, 256, 64
    pts0 = torch.randn(1, n, c).cuda()
    idx0 = torch.randint(0, n, (1, m), dtype=torch.int32).cuda()
    g0 = torch.randn(1, m, c).cuda()
    a = pts0.clone().requires_grad_()
    ops.gather_rows(a, idx0).backward(g0)
    b = pad_rows(pts0[0], capn).unsqueeze(0).requires_grad_()
    capy = ragged.Capacity([capn, capm], 32, 9, b.device)
    capy.fill([n, m])
    with capy:
        ops.gather_rows(b, pad_rows(idx0[0], capm, 0).unsqueeze(0)).backward(pad_rows(g0[0], capm).unsqueeze(0))
    torch.cuda.synchronize()
    assert torch.allclose(b.grad[0, :n], a.grad[0], rtol=1e-5, atol=1e-6)
    # relu(three_interpolate(points, idx, weight)): coarse rows m, fine rows n
    p0 = torch.randn(1, m, c).cuda()
    i3 = torch.randint(0, m, (1, n, 3), dtype=torch.int32).cuda()
    w3 = torch.rand(1, n, 3).cuda()
    go = torch.randn(1, n, c).cuda()
    a = p0.clone().requires_grad_()
    ops.three_interpolate_add_relu(a, i3, w3).backward(go)
    b = pad_rows(p0[0], capm).unsqueeze(0).requires_grad_()
    with capy:
        ops.three_interpolate_add_relu(b, pad_rows(i3[0], capn, 0).unsqueeze(0), pad_rows(w3[0], capn).unsqueeze(0)).backward(pad_rows(go[0], capn).unsqueeze(0))
    torch.cuda.synchronize()
    assert torch.allclose(b.grad[0, :m], a.grad[0], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("skip", [False, True])
@pytest.mark.parametrize("lazy", [False, True])
@pytest.mark.parametrize("widths", [[128, 128], [128, 128, 128]])
def test_feature_propagation_stage_under_a_capacity(skip, lazy, widths):
    """A whole SurfaceFeaturePropagationCD stage (segmentation/modules/repsurface_utils.py:233-284): with and without the skip branch,
    with the coarse rows handed over unmaterialised (LazyRows) and the output left lazy, as the decoder chains its stages."""
    from repsurf_amd import mlp, mlp_hip, ragged
    from tests.util import subproject
    torch_executor.set_backend("hip")
    torch.manual_seed(11)
    n, capn, m, capm, c2, c1 = 2564, 4096, 641, 1024, 128, 64
    with subproject("segmentation"):
        from modules.repsurface_utils import SurfaceFeaturePropagationCD, row_mlp
        fp0 = SurfaceFeaturePropagationCD(c2, c1 if skip else None, widths).cuda().train()
        pre0 = (nn.ModuleList([nn.Linear(96, c2)]).cuda(), nn.ModuleList([nn.BatchNorm1d(c2)]).cuda().train())      # the producer of the coarse rows
        post0 = (nn.ModuleList([nn.Linear(128, 128)]).cuda(), nn.ModuleList([nn.BatchNorm1d(128)]).cuda().train())     # ... and the consumer of this stage's rows
        bn_init([b for b in fp0.modules() if isinstance(b, nn.BatchNorm1d)] + list(pre0[1]) + list(post0[1]))
        src0 = torch.randn(m, 96).cuda()
        p1 = torch.randn(n, c1).cuda()
        idx0 = torch.randint(0, m, (n, 3), dtype=torch.int32).cuda()
        wt0 = torch.rand(n, 3).cuda()
        wt0 = wt0 / wt0.sum(1, keepdim=True)
        w0 = torch.randn(n, 128).cuda()
        res = []
        for ragged_run in (False, True):
            fp, pre, post = copy.deepcopy(fp0), copy.deepcopy(pre0), copy.deepcopy(post0)
            src = (pad_rows(src0, capm) if ragged_run else src0.clone()).requires_grad_()
            fine = (pad_rows(p1, capn) if ragged_run else p1.clone()).requires_grad_()
            idx = pad_rows(idx0, capn, 0) if ragged_run else idx0
            wt = pad_rows(wt0, capn) if ragged_run else wt0
            w = pad_rows(w0, capn) if ragged_run else w0

            def run():
                coarse = row_mlp(src, pre[0], pre[1], lazy_out=lazy)
                out = fp([None, fine if skip else None, None], [None, coarse, None], geometry=(idx, wt, None), lazy_out=lazy)
                y = row_mlp(out, post[0], post[1]) if lazy else out      # (a lazy activation needs a consumer that applies it: the next stage / the classifier)
                y.backward(w)
                return y
            if ragged_run:
                capy = ragged.Capacity([capn, capm], 32, 9, src.device)
                capy.fill([n, m])
                with capy:
                    out = run()
            else:
                out = run()
            torch.cuda.synchronize()
            res.append((out.detach(), [src.grad] + ([fine.grad] if skip else []), grads_of([fp, pre[0], pre[1], post[0], post[1]]), []))
        check(res[0], res[1], n, f"feature propagation stage skip={skip} lazy={lazy}")
