#!/usr/bin/env python3
"""Locate the first divergent backward intermediate of the HIP SA stack (case: 40 groups x 64, 6|138 -> 128,128,256)."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repsurf_amd import mlp_hip as H, mlp
from tests.test_mlp_gpu import make_cd, rel

groups, ns, pos, feat, widths = 40, 64, 6, 138, [128, 128, 256]
mod = make_cd(pos, feat, widths, 1)
g = torch.Generator().manual_seed(2)
x = torch.randn(groups * ns, pos + feat, generator=g).cuda()
w = torch.randn(groups, widths[-1], generator=g).cuda()
H.DEBUG = {}
xh = x.clone().requires_grad_()
out = mlp.sa_mlp_cd(xh, pos, mod.mlp_l0, mod.bn_l0, mod.mlp_f0, mod.bn_f0, mod.convs, mod.bns, ns)
(out * w).sum().backward()
D = H.DEBUG
# fp64 reference with autograd on the intermediates
md = copy.deepcopy(mod).double()
xd = x.double().requires_grad_()
import torch.nn.functional as F
def bn(y, m): return F.batch_norm(y, None, None, m.weight, m.bias, True, 0.1, m.eps)
yl = F.linear(xd[:, :pos], md.mlp_l0.weight.view(128, -1), md.mlp_l0.bias); yl.retain_grad()
yf = F.linear(xd[:, pos:], md.mlp_f0.weight.view(128, -1), md.mlp_f0.bias); yf.retain_grad()
z0 = bn(yl, md.bn_l0) + bn(yf, md.bn_f0); z0.retain_grad()
a0 = torch.relu(z0)
y1 = F.linear(a0, md.convs[0].weight.view(128, -1), md.convs[0].bias); y1.retain_grad()
z1 = bn(y1, md.bns[0]); z1.retain_grad()
a1 = torch.relu(z1)
y2 = F.linear(a1, md.convs[1].weight.view(256, -1), md.convs[1].bias)
o = torch.relu(bn(y2, md.bns[1])).view(groups, ns, -1).max(1)[0]
(o * w.double()).sum().backward()
print("out", rel(out.detach().double(), o.detach()))
print("yl fwd", rel(D["yl"].double(), yl.detach()), "yf fwd", rel(D["yf"].double(), yf.detach()))
L1 = D["layer1"]
print("dz1 (grad wrt z1)", rel(L1["dz"].double(), z1.grad))
dY1 = L1["p"].double() * L1["dz"].double() + L1["q"].double() * L1["y"].double() + L1["r"].double()
print("dY1", rel(dY1, y1.grad), "dg1", rel(L1["dg"].double(), md.bns[0].weight.grad), "db1", rel(L1["db"].double(), md.bns[0].bias.grad))
print("dz0 (grad wrt z0)", rel(D["dz0"].double(), z0.grad))
ps = D["part0"].sum(0)
print("part0 s0", rel(ps[0], z0.grad.sum(0)))
dYl = D["pl"].double() * D["dz0"].double() + D["ql"].double() * D["yl"].double() + D["rl"].double()
dYf = D["pf"].double() * D["dz0"].double() + D["qf"].double() * D["yf"].double() + D["rf"].double()
print("dYl", rel(dYl, yl.grad), "dYf", rel(dYf, yf.grad))
print("dgl", rel(D["dgl"].double(), md.bn_l0.weight.grad), "dbl", rel(D["dbl"].double(), md.bn_l0.bias.grad))
print("x grad feat", rel(xh.grad[:, pos:].double(), xd.grad[:, pos:]), "x grad pos (expect hip=0)", float(xh.grad[:, :pos].abs().max()))
print("Wl", rel(mod.mlp_l0.weight.grad.double(), md.mlp_l0.weight.grad), "Wf", rel(mod.mlp_f0.weight.grad.double(), md.mlp_f0.weight.grad))
# mask disagreement count
zh = D["vl"].scale * D["yl"] + D["vl"].shift + D["vf"].scale * D["yf"] + D["vf"].shift
print("mask mismatches", int(((zh > 0) != (z0.detach() > 0)).sum()), "of", zh.numel())
bad = (D["dz0"].double() - z0.grad).abs()
print("dz0 max abs err", float(bad.max()), "at", divmod(int(bad.argmax()), 128), "ref max", float(z0.grad.abs().max()))
rows_bad = (bad.max(1)[0] > 1e-6 * float(z0.grad.abs().max())).nonzero().flatten()
print("bad rows", rows_bad.numel(), rows_bad[:20].tolist())
