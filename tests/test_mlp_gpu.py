"""The fp32-MFMA shared-MLP kernels (repsurf_amd/csrc/mlp.hip via repsurf_amd.mlp_hip) against a
plain PyTorch fp32 reference of the same op (repsurf_amd.mlp "torch" executor: F.linear /
F.batch_norm / relu / max): forward activations and every gradient.  Tolerance: 1e-5 of the
tensor's scale for activations (north-star), 2e-4 for gradients (BatchNorm-backward cancellation
in fp32 on both sides; the fp64 check below shows the HIP path is the closer of the two)."""
import copy

import numpy as np
import pytest
import torch

from tests import torch_executor
import torch.nn as nn

pytestmark = pytest.mark.gpu


def make_cd(pos, feat, mlp, seed):
    torch.manual_seed(seed)
    m = nn.Module()
    m.mlp_l0, m.mlp_f0 = nn.Conv2d(pos, mlp[0], 1), nn.Conv2d(feat, mlp[0], 1)
    m.bn_l0, m.bn_f0 = nn.BatchNorm2d(mlp[0]), nn.BatchNorm2d(mlp[0])
    m.convs = nn.ModuleList([nn.Conv2d(a, b, 1) for a, b in zip(mlp[:-1], mlp[1:])])
    m.bns = nn.ModuleList([nn.BatchNorm2d(b) for b in mlp[1:]])
    for bn in [m.bn_l0, m.bn_f0] + list(m.bns):
        nn.init.uniform_(bn.weight, 0.5, 1.5)
        nn.init.uniform_(bn.bias, -0.3, 0.3)
    return m.cuda().train()


def run_cd(mod, x, ns, pos, backend, w):
    from repsurf_amd import mlp
    torch_executor.set_backend(backend)
    mod.zero_grad()
    x = x.clone().requires_grad_()
    out = mlp.sa_mlp_cd(x, pos, mod.mlp_l0, mod.bn_l0, mod.mlp_f0, mod.bn_f0, mod.convs, mod.bns, ns)
    (out * w).sum().backward()
    grads = {n: p.grad.clone() for n, p in mod.named_parameters()}
    grads["x"] = x.grad[:, pos:].clone()      # position channels carry no gradient in the model (coordinates)
    return out.detach(), grads


def rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def rel_l2(a, b):
    """Relative L2 error: robust to the one-in-a-million ReLU-mask flip (|z| ~ 1e-8 changes sign between
    two roundings of the same BatchNorm affine), which moves single rows of a gradient by O(1)."""
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


CASES = [  # groups, nsample, pos, feat, mlp
    (96, 32, 6, 10, [64, 64, 128]),        # sa1 shape
    (40, 64, 6, 138, [128, 128, 256]),     # sa2 shape
    (4, 128, 6, 266, [256, 512, 1024]),    # sa3 (group_all) shape
    (37, 24, 6, 10, [128, 128, 256]),      # 2x model: rows not a multiple of the tile, nsample 24
    (5, 7, 3, 5, [16, 40]),                # odd everything
]


@pytest.mark.parametrize("groups,ns,pos,feat,widths", CASES)
def test_sa_cd_stack_matches_torch(groups, ns, pos, feat, widths):
    mod = make_cd(pos, feat, widths, 1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(groups * ns, pos + feat, generator=g).cuda()
    w = torch.randn(groups, widths[-1], generator=g).cuda()
    ref_mod = copy.deepcopy(mod)
    out_t, g_t = run_cd(ref_mod, x, ns, pos, "torch", w)
    out_h, g_h = run_cd(mod, x, ns, pos, "hip", w)
    assert rel(out_h, out_t) < 1e-5, rel(out_h, out_t)      # measured <= 1.0e-6 on the five shapes under both product arithmetics (profiles/r05/mlp_rel_cases.txt)
    for name in g_t:
        if ".bias" in name and ("mlp_l0" in name or "mlp_f0" in name or "convs" in name):
            assert g_h[name].abs().max() == 0          # bias before BatchNorm: analytic zero
            continue
        assert rel_l2(g_h[name], g_t[name]) < 3e-3, (name, rel_l2(g_h[name], g_t[name]))
    # running statistics follow nn.BatchNorm2d
    for a, b in zip([mod.bn_l0, mod.bn_f0] + list(mod.bns), [ref_mod.bn_l0, ref_mod.bn_f0] + list(ref_mod.bns)):
        assert torch.allclose(a.running_mean, b.running_mean, atol=1e-5)
        assert torch.allclose(a.running_var, b.running_var, rtol=1e-4, atol=1e-6)
        assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == 1


def test_sa_cd_stack_against_fp64():
    """who is right when hip and torch-fp32 disagree in the 5th digit: compare both with fp64"""
    groups, ns, pos, feat, widths = 64, 32, 6, 10, [64, 64, 128]
    mod = make_cd(pos, feat, widths, 3)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(groups * ns, pos + feat, generator=g).cuda()
    w = torch.randn(groups, widths[-1], generator=g).cuda()
    out_h, g_h = run_cd(copy.deepcopy(mod), x, ns, pos, "hip", w)
    out_t, g_t = run_cd(copy.deepcopy(mod), x, ns, pos, "torch", w)
    out_d, g_d = run_cd(copy.deepcopy(mod).double(), x.double(), ns, pos, "torch", w.double())
    assert rel(out_h.double(), out_d) < 1e-5
    for name in ("mlp_l0.weight", "mlp_f0.weight", "convs.0.weight", "convs.1.weight", "bns.1.weight", "x"):
        eh, et = rel_l2(g_h[name].double(), g_d[name]), rel_l2(g_t[name].double(), g_d[name])
        assert eh < 1e-3, (name, eh, et)


@pytest.mark.parametrize("aggr", ["sum", "max", "avg"])
def test_umbrella_stack_matches_torch(aggr):
    from repsurf_amd import mlp
    torch.manual_seed(5)
    mlps = nn.Sequential(nn.Conv2d(10, 10, 1, bias=False), nn.BatchNorm2d(10), nn.ReLU(True), nn.Conv2d(10, 10, 1),
                         nn.BatchNorm2d(10), nn.ReLU(True), nn.Conv2d(10, 10, 1)).cuda().train()
    x = torch.randn(300 * 8, 10).cuda()
    w = torch.randn(300, 10).cuda()
    res = {}
    for backend in ("torch", "hip"):
        torch_executor.set_backend(backend)
        m = copy.deepcopy(mlps)
        out = mlp.umbrella_mlp(x, m, 8, aggr)
        (out * w).sum().backward()
        res[backend] = (out.detach(), {n: p.grad.clone() for n, p in m.named_parameters()})
    from tests.util import parity_report
    parity_report(f"umbrella_stack_{aggr}_vs_torch", out_rel=rel(res["hip"][0], res["torch"][0]))
    assert rel(res["hip"][0], res["torch"][0]) < 1e-5       # the north-star's bound for fp32 activations (2e-5 until round 6)
    for name, gt in res["torch"][1].items():
        if name == "3.bias":
            continue
        assert rel_l2(res["hip"][1][name], gt) < 3e-3, name


def test_plain_stack_matches_torch():
    from repsurf_amd import mlp
    torch.manual_seed(6)
    convs = nn.ModuleList([nn.Conv2d(19, 32, 1), nn.Conv2d(32, 64, 1)]).cuda()
    bns = nn.ModuleList([nn.BatchNorm2d(32), nn.BatchNorm2d(64)]).cuda().train()
    x0 = torch.randn(50 * 16, 19).cuda()
    w = torch.randn(50, 64).cuda()
    res = {}
    for backend in ("torch", "hip"):
        torch_executor.set_backend(backend)
        c, b = copy.deepcopy(convs), copy.deepcopy(bns)
        x = x0.clone().requires_grad_()
        out = mlp.sa_mlp_plain(x, c, b, 16)
        (out * w).sum().backward()
        res[backend] = (out.detach(), x.grad.clone(), [p.grad.clone() for p in c.parameters()])
    from tests.util import parity_report
    parity_report("plain_stack_vs_torch", out_rel=rel(res["hip"][0], res["torch"][0]))
    assert rel(res["hip"][0], res["torch"][0]) < 1e-5
    assert rel_l2(res["hip"][1], res["torch"][1]) < 3e-3
    for gh, gt in zip(res["hip"][2][::2], res["torch"][2][::2]):
        assert rel_l2(gh, gt) < 3e-3


@pytest.mark.parametrize("radius,ns", [(0.25, 16), (0.5, 32), (2.0, 8)])
def test_compacted_groups_match_dense(radius, ns):
    """The compacted operand (distinct ball-query slots + multiplicities) gives the same stage output and the
    same gradients as the dense one: padding copies are exact duplicates, so this is an identity, not an
    approximation.  radius 2.0 fills every ball (no padding at all), 0.25 leaves mostly padding."""
    from repsurf_amd import mlp, ops
    from tests.util import cloud
    torch_executor.set_backend("hip")
    b, n, s, cn, cf = 2, 256, 64, 10, 12
    xyz = torch.from_numpy(cloud(77, b, n)).cuda()
    fps = ops.furthestsampling(xyz, s)
    centres = ops.gather_rows(xyz, fps)
    idx, cnt = ops.ballquery(radius, ns, xyz, centres, return_count=True)
    assert ((idx[:, :, 1:] == idx[:, :, :1]).sum(-1) == ns - cnt).all()      # cnt = distinct slots
    g = torch.Generator().manual_seed(3)
    normal0, feature0 = torch.randn(b, n, cn, generator=g).cuda(), torch.randn(b, n, cf, generator=g).cuda()
    w = torch.randn(b * s, 48, generator=g).cuda()
    mod = make_cd(6, cn + cf, [32, 32, 48], 9)
    res = {}
    for kind in ("dense", "compact"):
        m = copy.deepcopy(mod)
        normal, feature = normal0.clone().requires_grad_(), feature0.clone().requires_grad_()
        if kind == "dense":
            x = ops.group_features(xyz, centres, normal, feature, idx, polar=True)
            out = mlp.sa_mlp_cd(x, 6, m.mlp_l0, m.bn_l0, m.mlp_f0, m.bn_f0, m.convs, m.bns, ns)
        else:
            cg = ops.group_features_compact(xyz, centres, normal, feature, idx, cnt, polar=True)
            assert int(cg.offsets[-1]) == int(cnt.sum())
            out = mlp.sa_mlp_cd(cg.x, 6, m.mlp_l0, m.bn_l0, m.mlp_f0, m.bn_f0, m.convs, m.bns, ns, compact=cg)
        (out * w).sum().backward()
        res[kind] = (out.detach(), normal.grad.clone(), feature.grad.clone(),
                     {k: p.grad.clone() for k, p in m.named_parameters()},
                     [bn.running_var.clone() for bn in [m.bn_l0, m.bn_f0] + list(m.bns)])
    d, c = res["dense"], res["compact"]
    from tests.util import parity_report
    parity_report(f"compacted_vs_dense_r{radius}_ns{ns}", out_rel=rel(c[0], d[0]))
    assert rel(c[0], d[0]) < 1e-5
    assert rel_l2(c[1], d[1]) < 3e-3 and rel_l2(c[2], d[2]) < 3e-3
    for k in d[3]:
        if ".bias" in k and ("mlp_l0" in k or "mlp_f0" in k or "convs" in k):
            continue
        assert rel_l2(c[3][k], d[3][k]) < 3e-3, k
    for a, bb in zip(c[4], d[4]):
        assert torch.allclose(a, bb, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("groups,ns,c", [(32, 128, 1024), (5, 64, 70), (3, 200, 64), (4096, 32, 128)])
def test_pool_max_first_maximum(groups, ns, c):
    """rs_pool_max: max over the nsample rows of a group of relu(scale*y+shift) and the FIRST row attaining it
    (torch.max's tie rule, classification/modules/repsurface_utils.py:245); values are quantised so that ties occur.
    Both the per-(group, channel) kernel and the sliced kernel for few long groups are hit.  Bit-exact."""
    import ctypes
    from repsurf_amd import _lib
    g = torch.Generator().manual_seed(groups + ns)
    y = (torch.randint(-3, 4, (groups * ns, c), generator=g).float() * 0.5).cuda()
    scale = (torch.randint(-20, 44, (c,), generator=g).float() / 64).cuda()      # few mantissa bits: fma == mul + add exactly
    shift = (torch.randint(-8, 8, (c,), generator=g).float() / 64).cuda()
    out = torch.empty((groups, c), dtype=torch.float32, device="cuda")
    arg = torch.empty((groups, c), dtype=torch.int32, device="cuda")
    _lib.call("rs_pool_max", groups, ns, c, 1, None, y.data_ptr(), 0, scale.data_ptr(), shift.data_ptr(), out.data_ptr(),
              arg.data_ptr(), torch.cuda.current_stream().cuda_stream)
    z = torch.relu(torch.addcmul(shift, y, scale)).view(groups, ns, c)      # fma(scale, y, shift) like the kernel
    ref = z.max(dim=1)
    assert torch.equal(out, ref.values)
    first = (z == ref.values.unsqueeze(1)).float().argmax(dim=1)
    assert torch.equal(arg.long(), first)


# ---------------------------------------------------------------------------------------------------------------
# bf16 mixed precision (BASELINE configs[4]): rs_mlp_gemm_rows_bf16 = operands rounded to bf16 at the LDS commit,
# v_mfma_f32_32x32x16_bf16, fp32 accumulation; fp32 or (activation storage, below) bf16 output.
def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("rows,k,n", [(300, 128, 128), (4096, 512, 1024), (1000, 64, 64), (777, 40, 200),
                                      (130, 268, 256), (65, 16, 32), (512, 6, 64)])
def test_bf16_row_gemm_is_the_rounded_operand_product(rows, k, n):
    """The kernel's contract, exactly: out = bf16(x) . bf16(w)^T + bias accumulated in fp32.  Products of two
    bf16 values are exact in fp32, so against the same product in fp64 only the fp32 accumulation order is left
    (<= 1e-5 of the output scale) -- a wrong fragment layout, a dropped k-step or an unrounded operand would be O(1) / 4e-3.
    Asymmetric random operands; ragged K (40, 268, 6), rows and columns that do not fill a tile."""
    from repsurf_amd import mlp_hip as H, mlp
    g = torch.Generator().manual_seed(rows + k)
    x = torch.randn(rows, k, generator=g).cuda()
    w = (torch.randn(n, k, generator=g) / k ** 0.5).cuda()
    bias = torch.randn(n, generator=g).cuda()
    out = torch.full((rows, n), float("nan"), device="cuda")
    wk = H.w_fwd(w)
    epi = H.Epilogue(bias=H._ptr(bias), out=H._ptr(out), ldo=n, mode=H.EPI_STORE)
    mlp.set_precision("bf16")
    try:
        H.gemm_rows(rows, k, n, H.operand(H.OP_ID, x, k), wk, epi)
    finally:
        mlp.set_precision("fp32")
    ref = _bf16_round(x) @ _bf16_round(w).T + bias.double()
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-5, err
    # and it is NOT the fp32 product (the bf16 pipe really ran), unless the layout forced the fp32 instance
    full = x.double() @ w.double().T + bias.double()
    if k % 2 == 0 and k > 16:
        assert (out.double() - full).abs().max().item() / full.abs().max().item() > 1e-4


def test_bf16_row_gemm_fused_prologue_and_statistics():
    """BN+ReLU prologue in fp32 BEFORE the rounding, BatchNorm column sums of the fp32 output in the epilogue."""
    from repsurf_amd import mlp_hip as H, mlp
    rows, k, n = 5000, 128, 256
    g = torch.Generator().manual_seed(11)
    # few mantissa bits: s*y + t is exact in fp32 whether fused or not, so the value that gets rounded to bf16 is
    # the same in the kernel and in the reference below
    y = (torch.randint(-32, 33, (rows, k), generator=g).float() / 8).cuda()
    s, t = (torch.randint(32, 96, (k,), generator=g).float() / 64).cuda(), (torch.randint(-8, 8, (k,), generator=g).float() / 64).cuda()
    w = (torch.randn(n, k, generator=g) / k ** 0.5).cuda()
    out = torch.empty(rows, n, device="cuda")
    part = torch.empty((H.PARTIAL_BLOCKS, 2, n), dtype=torch.float64, device="cuda")
    epi = H.Epilogue(bias=None, out=H._ptr(out), ldo=n, mode=H.EPI_STATS, partial=part.data_ptr(),
                     partial_blocks=H.PARTIAL_BLOCKS)
    mlp.set_precision("bf16")
    try:
        H.gemm_rows(rows, k, n, H.operand(H.OP_RELU1, y, k, s1=s, t1=t), H.w_fwd(w), epi)
    finally:
        mlp.set_precision("fp32")
    act = torch.relu(torch.addcmul(t, y, s))                       # fma(s, y, t) like the kernel
    ref = _bf16_round(act) @ _bf16_round(w).T
    assert (out.double() - ref).abs().max().item() / ref.abs().max().item() < 1e-5
    sums = part.sum(0)
    assert torch.allclose(sums[0], out.double().sum(0), rtol=1e-5, atol=2e-3)      # per-tile fp32 column sums, fp64 across tiles
    assert torch.allclose(sums[1], (out.double() ** 2).sum(0), rtol=1e-5, atol=2e-3)


@pytest.mark.parametrize("groups,ns,pos,feat,widths", CASES[:4])
def test_bf16_sa_stack_within_restated_tolerance(groups, ns, pos, feat, widths):
    """SURVEY.md §8(d) C5 ("tolerance restated vs the fp32 reference ... to be tightened empirically").  The yardstick is
    what the reference itself would do in bf16: the torch executor (F.linear / F.batch_norm / relu / max, i.e. the
    reference's Conv2d-BN-ReLU stack) under torch.autocast(bfloat16).  Like autocast, this path rounds the operands AND
    stores the conv outputs as bf16 (fp32 accumulation, fp32 BatchNorm statistics of the stored values).  Measured
    (MI355X): gradient cosine against fp32 0.987-0.996 here, 0.980-0.992 for autocast -- the max-pool argmax moves for
    near-tied rows in any bf16 forward, which re-routes whole gradient rows, so 0.999 is not reachable by either.
    Asserted: activations <= 2e-2 of the output scale and no worse than 1.1 x autocast; every gradient's cosine >= 0.98
    and >= autocast's - 5e-3."""
    from repsurf_amd import mlp
    mod = make_cd(pos, feat, widths, 1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(groups * ns, pos + feat, generator=g).cuda()
    w = torch.randn(groups, widths[-1], generator=g).cuda()
    out_f, g_f = run_cd(copy.deepcopy(mod), x, ns, pos, "hip", w)
    mlp.set_precision("bf16")
    try:
        out_b, g_b = run_cd(copy.deepcopy(mod), x, ns, pos, "hip", w)
    finally:
        mlp.set_precision("fp32")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out_a, g_a = run_cd(copy.deepcopy(mod), x, ns, pos, "torch", w)
    torch_executor.set_backend("hip")
    err_b, err_a = (out_b - out_f).abs().max().item(), (out_a.float() - out_f).abs().max().item()
    assert err_b <= 2e-2 * max(1.0, out_f.abs().max().item()), err_b
    assert err_b <= 1.1 * err_a, (err_b, err_a)
    assert not torch.equal(out_b, out_f)

    def cos(a, b):
        return torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()
    for name in g_f:
        if g_f[name].abs().max() == 0:
            continue
        cb, ca = cos(g_b[name], g_f[name]), cos(g_a[name].float(), g_f[name])
        assert cb >= 0.98 and cb >= min(0.999, ca - 5e-3), (name, cb, ca)


@pytest.mark.parametrize("groups,ns,pos,feat,widths", CASES[:4])
def test_bf16_sa_stack_equals_the_rounded_operand_stack(groups, ns, pos, feat, widths):
    """the bf16 stack against a RESTATEMENT of itself (tests/torch_executor.py, backend "torch_bf16": torch fp32 ops
    with the kernels' roundings at the same points -- both GEMM operands after their fp32 prologue, the stored conv outputs, the
    incoming gradient as the P operand of data and weight gradient), not against "what autocast does".  bf16 x bf16 products are exact
    in fp32: on identical inputs only the summation order differs, plus the rare element a 1e-7 difference pushes across a bf16
    rounding boundary (one ulp = 4e-3).  Measured: every gradient cosine >= 0.99999 (0.98-0.99 against the fp32 stack), rel-L2 <= 5e-3,
    outputs within one bf16 ulp of the output scale."""
    from repsurf_amd import mlp
    mod = make_cd(pos, feat, widths, 1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(groups * ns, pos + feat, generator=g).cuda()
    w = torch.randn(groups, widths[-1], generator=g).cuda()
    mlp.set_precision("bf16")
    try:
        out_b, g_b = run_cd(copy.deepcopy(mod), x, ns, pos, "hip", w)
    finally:
        mlp.set_precision("fp32")
    out_r, g_r = run_cd(copy.deepcopy(mod), x, ns, pos, "torch_bf16", w)
    torch_executor.set_backend("hip")
    scale = out_r.abs().max().item()
    assert (out_b - out_r).abs().max().item() <= 2.0 ** -7 * scale                    # (two ulps of the largest value)
    assert ((out_b - out_r).abs() > 1e-4 * scale).float().mean().item() <= 0.05      # ... and only few elements at all
    for name in g_r:
        if g_r[name].abs().max() == 0 or (name.endswith(".bias") and not name.startswith(("bn", "bns"))):
            continue                                   # (a conv bias in front of a BatchNorm: analytic zero in the kernels, fp32 noise in the restatement)
        c = torch.nn.functional.cosine_similarity(g_b[name].flatten().double(), g_r[name].flatten().double(), dim=0).item()
        assert c >= 0.9999 and rel_l2(g_b[name], g_r[name]) <= 1e-2, (name, c, rel_l2(g_b[name], g_r[name]))


@pytest.mark.parametrize("rows,n,k", [(5000, 128, 128), (4096, 1024, 512), (1000, 128, 64), (777, 200, 40), (33, 64, 66),
                                      (20000, 256, 138)])
def test_bf16_weight_gradient_is_the_rounded_operand_product(rows, n, k):
    """rs_mlp_wgrad_bf16: dw[n][k] = sum_r bf16(P[r][n]) * bf16(Q[r][k]), fp32 accumulation (row pairs packed along the
    reduction index; rows that do not fill a 32-row stage or a pair; 138 = float2 operand)."""
    from repsurf_amd import mlp_hip as H, mlp
    g = torch.Generator().manual_seed(rows + n)
    p = torch.randn(rows, n, generator=g).cuda()
    q = torch.randn(rows, k, generator=g).cuda()
    mlp.set_precision("bf16")
    try:
        dw = H.wgrad(rows, n, k, H.operand(H.OP_ID, p, n), H.operand(H.OP_ID, q, k), p.device)
    finally:
        mlp.set_precision("fp32")
    ref = _bf16_round(p).T @ _bf16_round(q)
    err = (dw.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-5, err
    full = p.double().T @ q.double()
    assert (dw.double() - full).abs().max().item() / full.abs().max().item() > 1e-4      # the bf16 pipe really ran


# bf16 activation storage (configs[4]): the conv outputs y a stack saves for backward are written as bf16 by the producing
# GEMM and read back as bf16 by every consumer (the kernels fix WHICH tensors of a launch are bf16 by its operand mode:
# csrc/mlp.hip "storage roles").  Reading a bf16 tensor is exact, so a consumer must give BIT-IDENTICAL results on bf16
# tensors and on fp32 tensors holding the same (bf16-representable) values through the fp32-storage kernels; a producer must
# store exactly the nearest-even rounding of what the fp32-storage launch stores, and sum the statistics of the rounded values.
def _bf16_vals(shape, g, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("rows,k,n", [(5000, 128, 256), (333, 64, 64), (4096, 138, 128), (70, 32, 32)])
@pytest.mark.parametrize("mode", ["relu1", "relu2", "aff2", "pooled"])
def test_bf16_stored_operands_read_exactly(rows, k, n, mode):
    from repsurf_amd import mlp_hip as H, mlp
    ns = 5 if rows % 5 == 0 else 1
    g = torch.Generator().manual_seed(rows + k + len(mode))
    y_a, y_b = _bf16_vals((rows, k), g).cuda(), _bf16_vals((rows, k), g).cuda()
    dz = torch.randn(rows, k, generator=g).cuda()                          # masked gradient (fp32 in both storage modes)
    s1, t1, s2, t2 = (torch.randn(k, generator=g).cuda() for _ in range(4))
    w = (torch.randn(n, k, generator=g) / k ** 0.5).cuda()
    grad = torch.randn(rows // ns, k, generator=g).cuda()                 # pooled gradient (fp32, small)
    arg = torch.randint(0, ns, (rows // ns, k), generator=g, dtype=torch.int32).cuda()
    other = torch.randn(rows, n, generator=g).cuda()

    y_a32, y_b32 = y_a.float(), y_b.float()            # (kept alive here: an operand descriptor holds raw pointers only)

    def op(store):
        ya, yb = (y_a, y_b) if store else (y_a32, y_b32)
        if mode == "relu1":
            return H.operand(H.OP_RELU1, ya, k, s1=s1, t1=t1)
        if mode == "relu2":
            return H.operand(H.OP_RELU2, ya, k, yb, k, s1, t1, s2, t2)
        if mode == "aff2":
            return H.operand(H.OP_AFF2, dz, k, yb, k, s1=s1, t1=t1, s2=s2)
        return H.operand(H.OP_POOLED, grad, k, yb, k, s1=s1, t1=t1, s2=s2, arg=arg, ns=ns)
    res = []
    mlp.set_precision("bf16")
    try:
        for store in (True, False):
            fwd = mode in ("relu1", "relu2")                               # a forward operand's output is a y: bf16 when stored
            out = torch.zeros((rows, n), dtype=torch.bfloat16 if (store and fwd) else torch.float32, device="cuda")
            H.gemm_rows(rows, k, n, op(store), H.w_fwd(w), H.Epilogue(bias=None, out=H._ptr(out), ldo=n, mode=H.EPI_STORE, out_bf16=H._bf(out)))
            as_p = H.wgrad(rows, k, n, op(store), H.operand(H.OP_ID, other, n), dz.device)
            as_q = H.wgrad(rows, n, k, H.operand(H.OP_ID, other, n), op(store), dz.device) if fwd else as_p
            res.append((out, as_p, as_q))
    finally:
        mlp.set_precision("fp32")
    assert all(torch.isfinite(t.float()).all() for t in res[0])
    assert torch.equal(res[0][0], res[1][0].to(res[0][0].dtype))          # (forward: the rounded fp32-storage output)
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


@pytest.mark.parametrize("rows,k,n", [(5000, 128, 256), (333, 64, 64), (4096, 10, 64), (4099, 3, 64), (70, 256, 32), (1024, 128, 128)])
@pytest.mark.parametrize("epi", ["store", "stats", "stats_pool"])
def test_bf16_stored_output_is_the_rounded_fp32_output(rows, k, n, epi):
    from repsurf_amd import mlp_hip as H, mlp
    if epi == "stats_pool" and (rows % 8 or not H.fused_pool_ok(n, 8)):
        pytest.skip("fused pooling needs whole groups per thread")
    g = torch.Generator().manual_seed(rows + n + len(epi))
    x = torch.randn(rows, k + 1, generator=g).cuda()[:, 1:] if k == 3 else torch.randn(rows, k, generator=g).cuda()   # k = 3: odd base -> scalar loads
    ldx = x.stride(0)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).cuda()
    bias = torch.randn(n, generator=g).cuda()
    res = {}
    mlp.set_precision("bf16")
    try:
        for store in ("fp32", "bf16"):
            out = torch.zeros((rows, n), dtype=torch.float32 if store == "fp32" else torch.bfloat16, device="cuda")
            part = torch.zeros((H.PARTIAL_BLOCKS, 2, n), dtype=torch.float64, device="cuda")
            e = H.Epilogue(bias=H._ptr(bias), out=H._ptr(out), ldo=n, out_bf16=H._bf(out), mode=H.EPI_STORE if epi == "store" else H.EPI_STATS,
                           partial=part.data_ptr(), partial_blocks=H.PARTIAL_BLOCKS)
            pool = None
            if epi == "stats_pool":
                ext = torch.zeros((2, rows // 8, n), device="cuda")
                pos = torch.zeros((2, rows // 8, n), dtype=torch.int32, device="cuda")
                e.pool_ns, e.pool_max, e.pool_min, e.pool_amax, e.pool_amin = 8, H._ptr(ext[0]), H._ptr(ext[1]), pos[0].data_ptr(), pos[1].data_ptr()
                pool = (ext, pos)
            H.gemm_rows(rows, k, n, H.operand(H.OP_ID, x, ldx), H.w_fwd(w), e)
            res[store] = (out, part.sum(0), pool)
    finally:
        mlp.set_precision("fp32")
    assert torch.equal(res["bf16"][0], res["fp32"][0].to(torch.bfloat16))            # round to nearest even, nothing else
    o = res["bf16"][0].double()
    if epi != "store":      # sums of the STORED values
        assert torch.allclose(res["bf16"][1][0], o.sum(0), rtol=1e-5, atol=2e-3)
        assert torch.allclose(res["bf16"][1][1], (o * o).sum(0), rtol=1e-5, atol=2e-3)
    if epi == "stats_pool":  # extremes of the STORED values, first position attaining them
        ext, pos = res["bf16"][2]
        grp = res["bf16"][0].float().view(rows // 8, 8, n)
        assert torch.equal(ext[0], grp.max(1).values) and torch.equal(ext[1], grp.min(1).values)


@pytest.mark.parametrize("rows,k,n", [(5000, 128, 256), (333, 64, 64), (70, 256, 32), (1024, 128, 128)])
@pytest.mark.parametrize("two", [False, True])
def test_bf16_stored_mask_tensors_read_exactly(rows, k, n, two):
    """Data-gradient GEMM: the ReLU masks and the BatchNorm-backward sums rebuilt from bf16-stored y tensors equal the ones
    from fp32 tensors holding the same values, bit for bit (output dz and the fp64 partial sums)."""
    from repsurf_amd import mlp_hip as H, mlp
    g = torch.Generator().manual_seed(rows + n + two)
    dz_in, y_in = torch.randn(rows, k, generator=g).cuda(), _bf16_vals((rows, k), g).cuda()
    p, q, r = (torch.randn(k, generator=g).cuda() for _ in range(3))
    w = (torch.randn(k, n, generator=g) / k ** 0.5).cuda()
    y1, y2 = _bf16_vals((rows, n), g).cuda(), _bf16_vals((rows, n), g).cuda()
    v1, v2 = H.BNVec(n, dz_in.device), H.BNVec(n, dz_in.device)
    for v in (v1, v2):
        for t in (v.scale, v.shift, v.mean, v.invstd):
            t.copy_(torch.randn(n, generator=g))
    res = []
    f32 = {id(t): t.float() for t in (y_in, y1, y2)}     # (kept alive here: the descriptors hold raw pointers only)
    mlp.set_precision("bf16")
    try:
        for store in (True, False):
            cv = (lambda t: t) if store else (lambda t: f32[id(t)])
            p_op = H.operand(H.OP_AFF2, dz_in, k, cv(y_in), k, s1=p, t1=r, s2=q)
            dz, part, nstat = H.dgrad_masked(rows, k, n, p_op, w, cv(y1), v1, cv(y2) if two else None, v2 if two else None, device=dz_in.device)
            res.append((dz, part.sum(0)))
    finally:
        mlp.set_precision("fp32")
    assert torch.isfinite(res[0][0]).all() and res[0][0].abs().max() > 0
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("groups,ns,c,relu", [(2048, 32, 128, 1), (32, 128, 1024, 1), (700, 24, 64, 0)])
def test_bf16_stored_activation_pools_exactly(groups, ns, c, relu):
    from repsurf_amd import mlp_hip as H, _lib
    g = torch.Generator().manual_seed(groups + c)
    y16 = _bf16_vals((groups * ns, c), g).cuda()
    scale, shift, mean, invstd = (torch.randn(c, generator=g).cuda() for _ in range(4))
    dout = torch.randn(groups, c, generator=g).cuda()
    res = []
    for y in (y16, y16.float()):
        out = torch.empty((groups, c), device="cuda")
        arg = torch.empty((groups, c), dtype=torch.int32, device="cuda")
        _lib.call("rs_pool_max", groups, ns, c, relu, None, H._ptr(y), H._bf(y), H._ptr(scale), H._ptr(shift), H._ptr(out), arg.data_ptr(), H._stream())
        v = torch.empty_like(dout)
        part = torch.empty((H.PARTIAL_BLOCKS, 2, c), dtype=torch.float64, device="cuda")
        wide = torch.cat([torch.zeros(groups, 3, device="cuda"), dout, torch.ones(groups, 2, device="cuda")], 1)      # dout as a column slice
        v2, part2 = torch.empty_like(dout), torch.empty_like(part)
        _lib.call("rs_pool_max_backward", groups, ns, c, None, wide.data_ptr() + 12, c + 5, H._ptr(out) if relu else None, arg.data_ptr(), H._ptr(y), H._bf(y),
                  H._ptr(mean), H._ptr(invstd), H._ptr(v2), part2.data_ptr(), H.PARTIAL_BLOCKS, None, H._stream())
        _lib.call("rs_pool_max_backward", groups, ns, c, None, H._ptr(dout), 0, H._ptr(out) if relu else None, arg.data_ptr(), H._ptr(y), H._bf(y),
                  H._ptr(mean), H._ptr(invstd), H._ptr(v), part.data_ptr(), H.PARTIAL_BLOCKS, None, H._stream())
        assert torch.equal(v, v2) and torch.equal(part, part2)            # row pitch of dout; spare partial rows zeroed by the kernel
        res.append((out, arg, v, part.sum(0)))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    z = y16.float().view(groups, ns, c) * scale + shift
    ref = (torch.relu(z) if relu else z).max(1).values
    assert torch.allclose(res[0][0], ref, rtol=1e-6, atol=1e-6)


def test_batched_small_launches_equal_the_single_ones():
    """rs_bn_finalize_batch / rs_backward_tail (several BatchNorm finalizes and weight-gradient reductions in one launch)
    against rs_bn_finalize / rs_bn_backward_finalize / rs_reduce_partials one at a time: bit-identical outputs."""
    import ctypes
    from repsurf_amd import mlp_hip as H, _lib
    g = torch.Generator().manual_seed(21)
    dev = torch.device("cuda")
    rows = 70000
    # forward statistics of three layers of different widths
    single, items = [], []
    for c in (64, 10, 200):
        part = torch.randn(H.PARTIAL_BLOCKS, 2, c, generator=g, dtype=torch.float64).cuda()
        part[:, 1].abs_().mul_(rows / H.PARTIAL_BLOCKS).add_(part[:, 0] ** 2)
        gamma, beta = torch.randn(c, generator=g).cuda(), torch.randn(c, generator=g).cuda()
        rm, rv = torch.randn(c, generator=g).cuda(), torch.rand(c, generator=g).cuda()
        outs = []
        for _ in range(2):
            vec, rm_, rv_ = H.BNVec(c, dev), rm.clone(), rv.clone()
            outs.append((vec, rm_, rv_))
        vec, rm_, rv_ = outs[0]
        _lib.call("rs_bn_finalize", c, rows, H.PARTIAL_BLOCKS, part.data_ptr(), H._ptr(gamma), H._ptr(beta), 1e-5, 0.1, H._ptr(vec.scale),
                  H._ptr(vec.shift), H._ptr(vec.mean), H._ptr(vec.invstd), H._ptr(rm_), H._ptr(rv_), H._stream())
        vec, rm_, rv_ = outs[1]
        items.append((H.BnItem(c=c, nblk=H.PARTIAL_BLOCKS, rows=rows, partial=part.data_ptr(), gamma=H._ptr(gamma), beta=H._ptr(beta), eps=1e-5,
                               momentum=0.1, scale=H._ptr(vec.scale), shift=H._ptr(vec.shift), save_mean=H._ptr(vec.mean),
                               save_invstd=H._ptr(vec.invstd), running_mean=H._ptr(rm_), running_var=H._ptr(rv_)), part, gamma, beta))
        single.append(outs)
    H.bn_finalize_batch(items)
    for (a, arm, arv), (b, brm, brv) in single:
        for x, y in ((a.scale, b.scale), (a.shift, b.shift), (a.mean, b.mean), (a.invstd, b.invstd), (arm, brm), (arv, brv)):
            assert torch.equal(x, y)
        assert torch.isfinite(a.scale).all()
    # backward: two finalizes from one three-statistic partial + three pending reductions
    c = 128
    part = torch.randn(H.PARTIAL_BLOCKS, 3, c, generator=g, dtype=torch.float64).cuda()
    v1, v2 = H.BNVec(c, dev), H.BNVec(c, dev)
    for v in (v1, v2):
        for t in (v.scale, v.shift, v.mean, v.invstd):
            t.copy_(torch.randn(c, generator=g))
    reds = [(torch.randn(ch, n, generator=g).cuda(), ch, n) for ch, n in ((512, 64 * 6), (16, 512 * 1024), (37, 1000))]
    ref_c = []
    for which, v in ((1, v1), (2, v2)):
        buf = torch.empty((5, c), device=dev)
        _lib.call("rs_bn_backward_finalize", c, rows, H.PARTIAL_BLOCKS, 3, which, part.data_ptr(), H._ptr(v.scale), H._ptr(v.mean),
                  H._ptr(v.invstd), H._ptr(buf[0]), H._ptr(buf[1]), H._ptr(buf[2]), H._ptr(buf[3]), H._ptr(buf[4]), H._stream())
        ref_c.append(buf)
    ref_r = []
    for p_, ch, n in reds:
        out = torch.empty(n, device=dev)
        _lib.call("rs_reduce_partials", ch, n, H._ptr(p_), H._ptr(out), H._stream())
        ref_r.append(out)
    outs_r = [torch.empty(n, device=dev) for _, _, n in reds]
    assert not H._pending_reduce
    for (p_, ch, n), o in zip(reds, outs_r):
        H._pending_reduce.append((p_, ch, n, o))
    got = H.bwd_coeffs_multi([(c, rows, part, 3, 1, v1, None, False), (c, rows, part, 3, 2, v2, None, False)], dev)
    assert not H._pending_reduce
    for ref, tup in zip(ref_c, got):
        for i in range(5):
            assert torch.equal(ref[i], tup[i])
    for a, b in zip(ref_r, outs_r):
        assert torch.equal(a, b)
    fin = H.bwd_coeffs(c, rows, part, 3, 2, v2, dev)                         # single finalize, nothing pending
    assert all(torch.equal(ref_c[1][i], fin[i]) for i in range(5))


@pytest.mark.parametrize("relu_last", [True, False])
@pytest.mark.parametrize("rows,cin,widths", [(5000, 64, [128]), (777, 256, [256, 128]), (65536, 128, [128])])
def test_row_stack_of_single_row_groups_matches_torch(rows, cin, widths, relu_last):
    """[Linear, BatchNorm1d, ReLU]* on ungrouped rows through the fused kernels (groups of ONE row: segmentation feature
    propagation and classifier, segmentation/modules/repsurface_utils.py row_mlp), with and without the last ReLU,
    against the framework's Linear / BatchNorm1d / relu: output, input gradient, every parameter gradient."""
    from repsurf_amd import mlp
    torch_executor.set_backend("hip")
    torch.manual_seed(rows)
    lins = nn.ModuleList([nn.Linear(a, b) for a, b in zip([cin] + widths[:-1], widths)]).cuda()
    bns = nn.ModuleList([nn.BatchNorm1d(b) for b in widths]).cuda().train()
    for bn in bns:
        nn.init.uniform_(bn.weight, 0.5, 1.5)
        nn.init.uniform_(bn.bias, -0.3, 0.3)
    x0 = torch.randn(rows, cin).cuda()
    w = torch.randn(rows, widths[-1]).cuda()
    res = {}
    for kind in ("torch", "hip"):
        l, b = copy.deepcopy(lins), copy.deepcopy(bns)
        x = x0.clone().requires_grad_()
        if kind == "hip":
            out = mlp.sa_mlp_plain(x, l, b, 1, relu_last)
        else:
            out = x
            for i, (lin, bn) in enumerate(zip(l, b)):
                out = bn(lin(out))
                if relu_last or i + 1 < len(l):
                    out = torch.relu(out)
        (out * w).sum().backward()
        res[kind] = (out.detach(), x.grad.clone(), {n: p.grad.clone() for n, p in list(l.named_parameters()) + [("bn." + k, v) for k, v in b.named_parameters()]},
                     [bn.running_var.clone() for bn in b])
    from tests.util import parity_report
    parity_report("plain_stack_vs_torch", out_rel=rel(res["hip"][0], res["torch"][0]))
    assert rel(res["hip"][0], res["torch"][0]) < 1e-5
    assert rel_l2(res["hip"][1], res["torch"][1]) < 3e-3
    for name, gt in res["torch"][2].items():
        if name.endswith(".bias") and not name.startswith("bn."):
            continue                                   # Linear bias before BatchNorm: analytic zero here, fp32 noise there
        assert rel_l2(res["hip"][2][name], gt) < 3e-3, name
    for a, bb in zip(res["hip"][3], res["torch"][3]):
        assert torch.allclose(a, bb, rtol=1e-4, atol=1e-6)


def test_prepacked_weight_copies_do_not_outlive_their_model():
    """The ahead-of-time weight copies are looked up by address + version: a dead model's entry must not be found by a new
    model's weight that the allocator placed at the same address (regression: wrong data gradients in a segmentation
    model built after a classifier had run in the same process)."""
    from repsurf_amd import mlp_hip as H
    torch.manual_seed(0)
    for trial in range(4):
        a = nn.Conv2d(8, 12, 1).cuda()
        H.prepack([a])
        del a
        b = nn.Conv2d(8, 12, 1).cuda()                     # same size: the caching allocator likes to reuse the block
        with torch.no_grad():
            b.weight.mul_(1.0)                             # (version 1, like a parameter after its first optimizer step)
        w2d = H._w2d(b.weight)
        wt = H.pack_weights([w2d], True, w2d.device)[0]
        torch.cuda.synchronize()
        assert torch.equal(wt[:, :12], w2d.t()), trial


@pytest.mark.parametrize("groups,ns,pos,feat,widths", [CASES[0], CASES[4]])
def test_sa_cd_stack_eval_mode_forward_and_backward(groups, ns, pos, feat, widths):
    """model.eval(): BatchNorm normalises with its running statistics (constants).  Forward and every gradient of the HIP
    stack against the PyTorch executor in eval mode (round 1 raised NotImplementedError in this backward)."""
    mod = make_cd(pos, feat, widths, 3)
    g = torch.Generator().manual_seed(5)
    for bn in [mod.bn_l0, mod.bn_f0] + list(mod.bns):
        bn.running_mean.copy_(torch.randn(bn.num_features, generator=g).cuda() * 0.2)
        bn.running_var.copy_(torch.rand(bn.num_features, generator=g).cuda() + 0.5)
    mod.eval()
    x = torch.randn(groups * ns, pos + feat, generator=g).cuda()
    w = torch.randn(groups, widths[-1], generator=g).cuda()
    ref_mod = copy.deepcopy(mod)
    out_t, g_t = run_cd(ref_mod, x, ns, pos, "torch", w)
    out_h, g_h = run_cd(mod, x, ns, pos, "hip", w)
    assert rel(out_h, out_t) < 1e-5, rel(out_h, out_t)      # measured <= 1.0e-6 on the five shapes under both product arithmetics (profiles/r05/mlp_rel_cases.txt)
    for name in g_t:
        assert rel_l2(g_h[name], g_t[name]) < 3e-3, (name, rel_l2(g_h[name], g_t[name]))
    assert torch.equal(mod.bn_l0.running_mean, ref_mod.bn_l0.running_mean)          # eval: statistics untouched


@pytest.mark.parametrize("groups,ns,pos,feat,widths", [CASES[0], CASES[1], (5, 7, 3, 5, [18, 42])])
@pytest.mark.parametrize("training", [True, False])
def test_bf16_storage_against_fp32_storage_of_the_same_stack(groups, ns, pos, feat, widths, training):
    """bf16 mode with and without bf16 activation storage (train and eval mode; the last case has widths that are no multiple of
    4: such a stack keeps fp32 storage and both runs must be IDENTICAL).  Storage rounds each conv output once more (to the
    format its consumer would have rounded it to anyway after BN+ReLU): the two runs stay within the bf16 tolerance of
    each other, and the stored run is really a different one."""
    from repsurf_amd import mlp, mlp_hip as H
    mod = make_cd(pos, feat, widths, 3)
    g = torch.Generator().manual_seed(9)
    for bn in [mod.bn_l0, mod.bn_f0] + list(mod.bns):
        bn.running_mean.copy_(torch.randn(bn.num_features, generator=g).cuda() * 0.2)
        bn.running_var.copy_(torch.rand(bn.num_features, generator=g).cuda() + 0.5)
    mod.train(training)
    x = torch.randn(groups * ns, pos + feat, generator=g).cuda()
    w = torch.randn(groups, widths[-1], generator=g).cuda()
    res = {}
    mlp.set_precision("bf16")
    try:
        for store in (True, False):
            H.BF16_STORE = store
            res[store] = run_cd(copy.deepcopy(mod), x, ns, pos, "hip", w)
    finally:
        H.BF16_STORE = True
        mlp.set_precision("fp32")
    (out_s, g_s), (out_f, g_f) = res[True], res[False]
    if any(c % 4 for c in widths):
        assert torch.equal(out_s, out_f) and all(torch.equal(g_s[n], g_f[n]) for n in g_f)
        return
    assert not torch.equal(out_s, out_f)
    assert torch.isfinite(out_s).all() and (out_s - out_f).abs().max().item() <= 2e-2 * max(1.0, out_f.abs().max().item())
    for name in g_f:
        if g_f[name].abs().max() == 0:
            continue
        cos = torch.nn.functional.cosine_similarity(g_s[name].flatten().double(), g_f[name].flatten().double(), dim=0).item()
        assert cos >= 0.98, (name, cos)


@pytest.mark.parametrize("kind", ["cls3", "seg2"])
def test_constructor_stacks_eval_mode_forward_and_backward(kind):
    """The umbrella constructors' MLPs in eval mode (running statistics): forward and every gradient, bias gradients of the
    convolutions in front of a BatchNorm included (non-zero through frozen statistics), against the PyTorch executor."""
    from repsurf_amd import mlp
    torch.manual_seed(7)
    if kind == "cls3":
        mlps = nn.Sequential(nn.Conv2d(10, 10, 1, bias=False), nn.BatchNorm2d(10), nn.ReLU(True), nn.Conv2d(10, 10, 1),
                             nn.BatchNorm2d(10), nn.ReLU(True), nn.Conv2d(10, 10, 1)).cuda()
    else:
        mlps = nn.Sequential(nn.Conv1d(10, 10, 1), nn.BatchNorm1d(10), nn.ReLU(True), nn.Conv1d(10, 10, 1)).cuda()
    for m in mlps:
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 1.5)
            nn.init.uniform_(m.weight, 0.5, 1.5)
            nn.init.uniform_(m.bias, -0.3, 0.3)
    mlps.eval()
    group = 8 if kind == "cls3" else 9
    x = torch.randn(200 * group, 10).cuda()
    w = torch.randn(200, 10).cuda()
    res = {}
    for backend in ("torch", "hip"):
        torch_executor.set_backend(backend)
        m = copy.deepcopy(mlps)
        out = mlp.umbrella_mlp(x, m, group, "sum") if kind == "cls3" else mlp.umbrella_mlp2(x, m, group)
        (out * w).sum().backward()
        res[backend] = (out.detach(), {n: p.grad.clone() for n, p in m.named_parameters()})
    torch_executor.set_backend("hip")
    assert rel(res["hip"][0], res["torch"][0]) < 2e-5
    for name, gt in res["torch"][1].items():
        assert rel_l2(res["hip"][1][name], gt) < 3e-3, name


@pytest.mark.parametrize("points", [200, 7000])
def test_seg_constructor_fused_two_layer_mlp_training(points):
    """mlp.umbrella_mlp2 in training mode (the fused two-layer passes of csrc/umbrella_mlp.hip) against the PyTorch executor AND
    against the generic row-GEMM path it replaces: output, running statistics and every gradient (the bias in front of the
    BatchNorm: exactly 0 here, rounding noise in torch)."""
    from repsurf_amd import mlp, mlp_hip as H
    torch.manual_seed(11)
    mlps = nn.Sequential(nn.Conv1d(10, 10, 1), nn.BatchNorm1d(10), nn.ReLU(True), nn.Conv1d(10, 10, 1)).cuda()
    nn.init.uniform_(mlps[1].weight, 0.5, 1.5)
    nn.init.uniform_(mlps[1].bias, -0.3, 0.3)
    nn.init.uniform_(mlps[0].bias, -0.5, 0.5)
    mlps.train()
    group = 9
    x = torch.randn(points * group, 10).cuda()
    w = torch.randn(points, 10).cuda()
    res = {}
    for name in ("torch", "generic", "fused"):
        torch_executor.set_backend("torch" if name == "torch" else "hip")
        H.FUSED_UMBRELLA = name == "fused"
        try:
            m = copy.deepcopy(mlps)
            out = mlp.umbrella_mlp2(x, m, group)
            (out * w).sum().backward()
        finally:
            H.FUSED_UMBRELLA = True
        res[name] = (out.detach(), {n: p.grad.clone() for n, p in m.named_parameters()}, m[1].running_mean.clone(), m[1].running_var.clone())
    torch_executor.set_backend("hip")
    for other in ("torch", "generic"):
        assert rel(res["fused"][0], res[other][0]) < 2e-5, other
        assert torch.allclose(res["fused"][2], res[other][2], rtol=1e-5, atol=1e-6) and torch.allclose(res["fused"][3], res[other][3], rtol=1e-5, atol=1e-6)
        for pname, gt in res[other][1].items():
            if pname == "0.bias":
                assert res["fused"][1][pname].abs().max() == 0 and gt.abs().max() < 1e-3
                continue
            assert rel_l2(res["fused"][1][pname], gt) < 3e-3, (other, pname)


def test_owned_pass_carries_pending_reductions_across_stacks_with_identical_gradients():
    """mlp_hip.owned_pass(): the weight-gradient partials a stack leaves pending are summed by the next stack's first finalize
    launch or by the end-of-pass callback -- every gradient bit-identical to the default (each stack completes its own), and
    nothing is left pending when backward returns."""
    from repsurf_amd import mlp, mlp_hip as H
    torch_executor.set_backend("hip")
    g = torch.Generator().manual_seed(31)
    m1, m2 = make_cd(6, 10, [32, 32, 64], 4), make_cd(6, 64, [64, 128], 5)
    groups, ns = 48, 16
    x1 = torch.randn(groups * ns, 16, generator=g).cuda()
    pos2 = torch.randn(groups, 6, generator=g).cuda()
    w = torch.randn(groups // 8, 128, generator=g).cuda()

    def run(owned):
        a, b = copy.deepcopy(m1), copy.deepcopy(m2)
        ctx = H.owned_pass() if owned else contextlib.nullcontext()
        with ctx:
            h1 = mlp.sa_mlp_cd(x1, 6, a.mlp_l0, a.bn_l0, a.mlp_f0, a.bn_f0, a.convs, a.bns, ns)            # (groups, 64)
            x2 = torch.cat([pos2, h1], 1)                                                                    # second stack: 8 rows per group
            out = mlp.sa_mlp_cd(x2, 6, b.mlp_l0, b.bn_l0, b.mlp_f0, b.bn_f0, b.convs, b.bns, 8)
            (out * w).sum().backward()
            assert not H._pending_reduce and not H._flush_armed
        return [p.grad.clone() for p in list(a.parameters()) + list(b.parameters())]
    import contextlib
    ga, gb = run(False), run(True)
    assert len(ga) == len(gb) and all(torch.equal(p, q) for p, q in zip(ga, gb))
    assert H.OWNED_PASS == 0


@pytest.mark.parametrize("layers,points,group", [(3, 300, 8), (3, 2048, 8), (3, 33, 9), (3, 40000, 8), (2, 200, 9), (2, 7000, 9), (2, 16, 8), (3, 1, 8), (2, 70000, 9)])
def test_constructor_mlp_on_the_matrix_pipe(layers, points, group):
    """csrc/umbrella_mfma.hip (v_mfma_f32_16x16x4_f32 tiles of 16 points, BatchNorm 0 from the moments of x, finalizes folded into the
    consuming passes) against (a) the fp64 evaluation of the same modules, (b) the register-resident VALU passes it replaces
    (csrc/umbrella_mlp.hip): output, both BatchNorms' running statistics, every gradient; point counts that are not a multiple of
    the 16-point tile, fans of 8 / 9, a single point, more tiles than waves; the moments handed in ahead of time (the geometry stage does) or not."""
    from repsurf_amd import mlp, mlp_hip as H
    torch.manual_seed(31 + points)
    if layers == 3:
        mlps = nn.Sequential(nn.Conv2d(10, 10, 1, bias=False), nn.BatchNorm2d(10), nn.ReLU(True), nn.Conv2d(10, 10, 1),
                             nn.BatchNorm2d(10), nn.ReLU(True), nn.Conv2d(10, 10, 1)).cuda()
        bn_idx = (1, 4)
    else:
        mlps = nn.Sequential(nn.Conv1d(10, 10, 1), nn.BatchNorm1d(10), nn.ReLU(True), nn.Conv1d(10, 10, 1)).cuda()
        nn.init.uniform_(mlps[0].bias, -0.5, 0.5)
        bn_idx = (1,)
    for i in bn_idx:
        nn.init.uniform_(mlps[i].weight, 0.5, 1.5)
        nn.init.uniform_(mlps[i].bias, -0.3, 0.3)
    mlps.train()
    x = (torch.randn(points * group, 10) * torch.linspace(0.3, 2.0, 10) + torch.linspace(-1, 1, 10)).cuda()
    w = torch.randn(points, 10).cuda()
    torch_executor.set_backend("hip")

    def run(kind):
        m = copy.deepcopy(mlps)
        xx = x
        if kind == "fp64":
            m, xx = m.double(), x.double()
            torch_executor.set_backend("torch")
        H.UMB_MFMA = kind in ("mfma", "mfma+moments")
        fwd3, H.UMB_MFMA_FWD3 = H.UMB_MFMA_FWD3, True          # (the three-layer constructor is on the VALU passes by default: measured, mlp_hip)
        try:
            mom = mlp.umbrella_moments(x) if kind == "mfma+moments" else None
            if layers == 3:
                out = mlp.umbrella_mlp(xx, m, group, "sum", moments=mom)
            else:
                out = mlp.umbrella_mlp2(xx, m, group, moments=mom)
            (out * w.to(out.dtype)).sum().backward()
        finally:
            H.UMB_MFMA, H.UMB_MFMA_FWD3 = True, fwd3
            torch_executor.set_backend("hip")
        stats = [t.clone() for i in bn_idx for t in (m[i].running_mean, m[i].running_var)]
        return out.detach(), {n: p.grad.clone() for n, p in m.named_parameters()}, stats

    ref, valu, new, new2 = run("fp64"), run("valu"), run("mfma"), run("mfma+moments")
    assert torch.equal(new[0], new2[0]) and all(torch.equal(new[1][k], new2[1][k]) for k in new[1])
    if layers == 3:      # forward on the matrix pipe, backward on the VALU passes (they read the vectors the forward published)
        H.UMB_MFMA_BWD3 = not H.UMB_MFMA_BWD3
        try:
            mixed = run("mfma")
        finally:
            H.UMB_MFMA_BWD3 = not H.UMB_MFMA_BWD3
        assert torch.equal(new[0], mixed[0])
        for k in new[1]:
            assert k == "3.bias" or rel_l2(mixed[1][k].double(), ref[1][k]) < 2e-3, k
    pre_bn_bias = "3.bias" if layers == 3 else "0.bias"
    single = points * group < 2           # one row: the batch variance is 0 and every BatchNorm gradient degenerates
    for name, got in (("valu", valu), ("mfma", new)):
        assert rel(got[0].double(), ref[0]) < (1e-5 if not single else 1e-3), (name, rel(got[0].double(), ref[0]))
        for a, b in zip(got[2], ref[2]):
            assert torch.allclose(a.double(), b, rtol=2e-5, atol=1e-6), name
        for pname, gt in ref[1].items():
            if pname == pre_bn_bias:
                assert got[1][pname].abs().max() == 0
                continue
            if single:
                continue
            assert rel_l2(got[1][pname].double(), gt) < 2e-3, (name, pname, rel_l2(got[1][pname].double(), gt))
    # the matrix-pipe path is at least as close to the fp64 evaluation as the path it replaces (up to noise)
    if not single:
        for pname, gt in ref[1].items():
            if pname != pre_bn_bias:
                assert rel_l2(new[1][pname].double(), gt) < 3 * rel_l2(valu[1][pname].double(), gt) + 1e-5, pname


def _products_against_fp64():
    """One row GEMM (BatchNorm + ReLU prologue) and one weight gradient against fp64 products: (forward max, forward rms,
    gradient rms relative to the gradient's rms)."""
    from repsurf_amd import mlp_hip as H
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(11)
    rows, k, n = 4096, 512, 1024
    x = torch.randn(rows, k, generator=g).to(dev)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
    s = (torch.rand(k, generator=g) + 0.5).to(dev)
    t = (torch.randn(k, generator=g) * 0.1).to(dev)
    out = torch.empty(rows, n, device=dev)
    H.gemm_rows(rows, k, n, H.operand(H.OP_RELU1, x, k, s1=s, t1=t), H.w_fwd(w), H.Epilogue(bias=None, out=H._ptr(out), ldo=n, mode=H.EPI_STORE))
    ref = torch.relu(x * s + t).double() @ w.double().t()
    err = (out.double() - ref).abs()
    p = torch.randn(rows, 64, generator=g).to(dev)
    dw = H.wgrad(rows, 64, k, H.operand(H.OP_ID, p, 64), H.operand(H.OP_ID, x, k), dev)
    dw64 = H.wgrad(rows, n, 64, H.operand(H.OP_ID, out, n), H.operand(H.OP_ID, p, 64), dev)      # <= 64 columns of Q: the split-product block
    refw, refw64 = p.double().t() @ x.double(), out.double().t() @ p.double()
    gw = max(((dw.double() - refw).pow(2).mean().sqrt() / refw.pow(2).mean().sqrt()).item(),
             ((dw64.double() - refw64).pow(2).mean().sqrt() / refw64.pow(2).mean().sqrt()).item())
    return err.max().item(), err.pow(2).mean().sqrt().item(), gw


def test_gemm_products_are_fp32_accurate():
    """The tiled kernels form their products as six bf16 MFMAs over three-part operands (include/repsurf_hip.h: rs_mlp_gemm_split3):
    the result must be as close to the fp64 product as the fp32 MFMA's -- measured 3.8e-6 max / 2.5e-7 rms on outputs of rms 0.75
    (fp32 MFMA: 3.6e-6 / 2.2e-7; the usual two-part, three-product split: 1.7e-5 / 3.1e-6, tools/probes/bf16_split_accuracy.py)."""
    from repsurf_amd import mlp_hip as H
    assert H.gemm_split3() == (__import__("os").environ.get("RS_GEMM_SPLIT3", "1") != "0")
    fmax, frms, gw = _products_against_fp64()
    assert fmax < 8e-6 and frms < 5e-7, (fmax, frms)
    assert gw < 2e-6, gw


@pytest.mark.parametrize("cout,cin", [(64, 128), (40, 19), (512, 1024), (37, 8), (130, 260)])
@pytest.mark.parametrize("transpose", [False, True])
def test_split_image_is_tile_ordered(cout, cin, transpose):
    """rs_pack_weights' three-part image (include/repsurf_hip.h: rs_pack_weights_args.dst3, round 6): dst3[q][k / 8][row][k % 8] of the
    n-major operand (the weight, or its transpose for the data gradient), parts = nearest-even bf16 of what the parts before left,
    zero beyond the inner dimension -- bit for bit against the same arithmetic in torch."""
    from repsurf_amd import mlp_hip as H
    if not H._presplit_on():
        pytest.skip("the fp32 MFMA instances read no image")
    g = torch.Generator(device="cpu").manual_seed(cout * 1000 + cin)
    w = (torch.randn(cout, cin, generator=g) * torch.logspace(-6, 3, cin)).cuda()
    op = H.w_bwd(w) if transpose else H.w_fwd(w)
    hit = H._split3_of(op)
    if hit is None:
        H._pack_items([(op, False)], op.device)
        hit = H._split3_of(op)
    img, ld3 = hit[0], hit[1]
    mat = (w.t() if transpose else w).contiguous().cpu()                   # (outer, inner)
    outer, inner = mat.shape
    assert ld3 % 32 == 0 and ld3 >= inner and tuple(img.shape) == (3, ld3 // 8, outer, 8)
    pad = torch.zeros(outer, ld3)
    pad[:, :inner] = mat
    h = pad.bfloat16()
    m = (pad - h.float()).bfloat16()
    lo = (pad - h.float() - m.float()).bfloat16()
    want = torch.stack([t.view(outer, ld3 // 8, 8).permute(1, 0, 2) for t in (h, m, lo)])
    assert torch.equal(img.cpu().view(torch.int16), want.contiguous().view(torch.int16))
    # and the row GEMM over it (columns beyond a tile clamp to the last row of the image): against fp64
    rows = 200
    x = torch.randn(rows, inner, generator=g)
    out = torch.empty(rows, outer, device="cuda")
    if inner % 2 == 0:                                                      # (vector operands: the tiled kernels)
        H.gemm_rows(rows, inner, outer, H.operand(H.OP_ID, x.cuda(), inner), op, H.Epilogue(bias=None, out=H._ptr(out), ldo=outer, mode=H.EPI_STORE))
        ref = x.double() @ mat.double().t()
        assert (out.cpu().double() - ref).abs().max() <= 1e-5 * max(ref.abs().max().item(), 1.0)


def test_gemm_products_with_non_finite_and_denormal_operands():
    """Edge operands of the row GEMM under whichever product arithmetic this process runs (the child of
    test_fp32_mfma_instances_still_pass repeats it under the fp32 MFMA):
      * a NaN operand makes its output row NaN in both arithmetics;
      * an infinite operand makes its output row NON-FINITE in both: the fp32 MFMA yields +-inf (NaN against a zero weight), the
        three-part split yields NaN (the residual x - bf16(x) of an infinity is inf - inf) -- a poisoned row either way, stated in
        include/repsurf_hip.h (rs_mlp_gemm_split3); the same holds for finite |x| >= 2^127 (2 - 2^-8) ~ 3.39e38, whose bf16 rounds
        to infinity;
      * every OTHER row of the same launch is untouched (bit-identical to the launch without the edge rows);
      * denormal operands (1e-40) contribute nothing measurable: the row equals the row with zeros there to 1e-30."""
    from repsurf_amd import mlp_hip as H
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(5)
    rows, k, n = 256, 128, 64
    x = torch.randn(rows, k, generator=g)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)

    def run(a):
        a = a.to(dev).contiguous()
        out = torch.empty(rows, n, device=dev)
        H.gemm_rows(rows, k, n, H.operand(H.OP_ID, a, k), H.w_fwd(w), H.Epilogue(bias=None, out=H._ptr(out), ldo=n, mode=H.EPI_STORE))
        return out.cpu()

    clean = run(x)
    edge = x.clone()
    edge[3, 7] = float("nan")
    edge[100, 0] = float("inf")
    edge[101, 5] = -float("inf")
    edge[200, 9] = 3.4e38
    edge[17, :4] = torch.tensor([1e-40, -1e-40, 3e-39, 1e-45])
    zeroed = x.clone()
    zeroed[17, :4] = 0
    out, out_z = run(edge), run(zeroed)
    assert torch.isnan(out[3]).all()
    assert not torch.isfinite(out[100]).any() and not torch.isfinite(out[101]).any()
    if H.gemm_split3():
        assert torch.isnan(out[100]).all() and not torch.isfinite(out[200]).any()
    else:
        assert torch.isinf(out[100]).all() and (out[100] == float("inf") * torch.sign(w[:, 0].cpu())).all()
        assert torch.isfinite(out[200]).all()
    others = [r for r in range(rows) if r not in (3, 17, 100, 101, 200)]
    assert torch.equal(out[others], clean[others])
    assert (out[17] - out_z[17]).abs().max() < 1e-30
    assert (clean - (x.double() @ w.double().cpu().t()).float()).abs().max() < 5e-6


def test_fp32_mfma_instances_still_pass():
    """RS_GEMM_SPLIT3=0 (the fp32 MFMA instances of the same kernels, read once per process): this file's tests in a child process."""
    import os
    import subprocess
    import sys
    from tests.conftest import ROOT
    if os.environ.get("RS_GEMM_SPLIT3", "1") == "0":
        pytest.skip("this process already runs the fp32 MFMA instances")
    env = dict(os.environ, RS_GEMM_SPLIT3="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_mlp_gpu.py", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
