import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repsurf_amd import mlp_hip as H, mlp
def run(rows, k, n, two):
    g = torch.Generator().manual_seed(rows + n + two)
    bfv = lambda shape: torch.randn(*shape, generator=g).to(torch.bfloat16)
    dz_in, y_in = torch.randn(rows, k, generator=g).cuda(), bfv((rows, k)).cuda()
    p, q, r = (torch.randn(k, generator=g).cuda() for _ in range(3))
    w = (torch.randn(k, n, generator=g) / k ** 0.5).cuda()
    y1, y2 = bfv((rows, n)).cuda(), bfv((rows, n)).cuda()
    v1, v2 = H.BNVec(n, dz_in.device), H.BNVec(n, dz_in.device)
    for v in (v1, v2):
        for t in (v.scale, v.shift, v.mean, v.invstd):
            t.copy_(torch.randn(n, generator=g))
    res = []
    mlp.set_precision("bf16")
    for store in (True, False):
        cv = (lambda t: t) if store else (lambda t: t.float())
        p_op = H.operand(H.OP_AFF2, dz_in, k, cv(y_in), k, s1=p, t1=r, s2=q)
        dz, part, nstat = H.dgrad_masked(rows, k, n, p_op, w, cv(y1), v1, cv(y2) if two else None, v2 if two else None, device=dz_in.device)
        res.append(dz)
    mlp.set_precision("fp32")
    E = (p * dz_in + (q * y_in.float() + r)).to(torch.bfloat16).double()
    ref = E @ w.to(torch.bfloat16).double()
    z = v1.scale * y1.float() + v1.shift
    if two: z = z + v2.scale * y2.float() + v2.shift
    ref = torch.where(z > 0, ref, torch.zeros_like(ref))
    for name, d in zip(("store", "fp32"), res):
        e = (d.double() - ref).abs()
        print(rows, k, n, two, name, "max err", e.max().item(), "rows with err>1e-3:", (e.max(1).values > 1e-3).sum().item(), "first bad rows", (e.max(1).values > 1e-3).nonzero().flatten()[:8].tolist(),
              "bad cols", (e.max(0).values > 1e-3).nonzero().flatten()[:8].tolist())
for a in [(333, 64, 64, False), (5000, 128, 256, False), (1024, 128, 128, True), (70, 256, 32, False), (320, 64, 64, False), (333, 64, 128, False), (333, 128, 64, False)]:
    run(*a)
