#!/usr/bin/env python3
"""GPU diagnostic of the low-level shared-MLP kernels against direct torch formulas (fp64).
Prints one line per (kernel mode, shape); never asserts."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repsurf_amd import mlp_hip as H, _lib

dev = "cuda"
torch.manual_seed(0)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def rnd(*s):
    return torch.randn(*s, device=dev)


def build_operand(mode, rows, cols, ns):
    """returns (RowOperand, fp64 reference matrix, keepalive)"""
    a, b = rnd(rows, cols), rnd(rows, cols)
    s1, t1, s2, t2 = rnd(cols), rnd(cols), rnd(cols), rnd(cols)
    if mode == H.OP_ID:
        return H.operand(mode, a, cols), a.double(), (a,)
    if mode == H.OP_RELU1:
        return H.operand(mode, a, cols, s1=s1, t1=t1), torch.relu(s1.double() * a.double() + t1.double()), (a, s1, t1)
    if mode == H.OP_RELU2:
        ref = torch.relu(s1.double() * a.double() + t1.double() + s2.double() * b.double() + t2.double())
        return H.operand(mode, a, cols, b, cols, s1, t1, s2, t2), ref, (a, b, s1, t1, s2, t2)
    if mode == H.OP_AFF2:
        ref = s1.double() * a.double() + s2.double() * b.double() + t1.double()
        return H.operand(mode, a, cols, b, cols, s1=s1, t1=t1, s2=s2), ref, (a, b, s1, t1, s2)
    g = rows // ns
    v = rnd(g, cols)
    if mode == H.OP_POOLED:
        arg = torch.randint(0, ns, (g, cols), device=dev, dtype=torch.int32)
        dz = torch.zeros(g, ns, cols, device=dev, dtype=torch.float64)
        dz.scatter_(1, arg.long().unsqueeze(1), v.double().unsqueeze(1))
        ref = s1.double() * dz.view(rows, cols) + s2.double() * b.double() + t1.double()
        return H.operand(mode, v, cols, b, cols, s1=s1, t1=t1, s2=s2, arg=arg, ns=ns), ref, (v, b, s1, t1, s2, arg)
    ref = v.double().unsqueeze(1).expand(g, ns, cols).reshape(rows, cols)
    return H.operand(mode, v, cols, ns=ns), ref, (v,)


names = {0: "ID", 1: "RELU1", 2: "RELU2", 3: "AFF2", 4: "POOLED", 5: "BCAST"}
results = []
shapes = [(2560, 128, 128, 64), (3000, 9, 5, 8), (1280, 266, 256, 128), (2560, 128, 138, 64), (888, 128, 128, 24), (2560, 64, 64, 32), (4096, 256, 128, 64),
          (640, 6, 64, 32), (640, 10, 10, 8), (512, 512, 1024, 128), (2560, 138, 128, 64), (2560, 128, 6, 64)]
for (rows, kdim, cols, ns) in shapes:
    for mode in range(6):
        for kbn in (1,):
            try:
                E, Eref, keep = build_operand(mode, rows, kdim, ns)
                w = rnd(kdim, cols)
                bias = rnd(cols)
                out = torch.full((rows, cols), float("nan"), device=dev)
                epi = H.Epilogue(bias=H._ptr(bias), out=H._ptr(out), ldo=cols, mode=H.EPI_STORE)
                H.gemm_rows(rows, kdim, cols, E, H._pad4(w), epi)
                ref = Eref @ w.double() + bias.double()
                r = rel(out, ref)
                results.append(("gemm_store", names[mode], kbn, rows, kdim, cols, r))
                if r > 1e-4 or r != r:
                    print("BAD gemm_store", names[mode], "kbn", kbn, rows, kdim, cols, r, flush=True)
            except Exception as e:
                print("EXC gemm_store", names[mode], kbn, rows, kdim, cols, repr(e)[:200], flush=True)
    # stats epilogue
    E, Eref, keep = build_operand(H.OP_RELU1, rows, kdim, ns)
    w, bias = rnd(cols, kdim), rnd(cols)
    out = torch.empty(rows, cols, device=dev)
    part = torch.full((H.PARTIAL_BLOCKS, 2, cols), float("nan"), device=dev, dtype=torch.float64)
    epi = H.Epilogue(bias=H._ptr(bias), out=H._ptr(out), ldo=cols, mode=H.EPI_STATS, partial=part.data_ptr(), partial_blocks=H.PARTIAL_BLOCKS)
    H.gemm_rows(rows, kdim, cols, E, H.w_fwd(w), epi)
    ref = Eref @ w.double().t() + bias.double()
    ps = part.sum(0)
    print("stats", rows, kdim, cols, "y", rel(out, ref), "sum", rel(ps[0], ref.sum(0)), "sumsq", rel(ps[1], (ref * ref).sum(0)), flush=True)
    # mask epilogue (dual and single)
    for dual in (0, 1):
        E, Eref, keep = build_operand(H.OP_AFF2, rows, kdim, ns)
        w = rnd(kdim, cols)
        y1, y2 = rnd(rows, cols), rnd(rows, cols)
        v1, v2 = H.BNVec(cols, dev), H.BNVec(cols, dev)
        for v in (v1, v2):
            v.scale.copy_(rnd(cols)); v.shift.copy_(rnd(cols)); v.mean.copy_(rnd(cols)); v.invstd.copy_(rnd(cols).abs() + 0.5)
        dz, part, nstat = H.dgrad_masked(rows, kdim, cols, E, w, y1, v1, y2 if dual else None, v2 if dual else None, device=dev)
        z = v1.scale.double() * y1.double() + v1.shift.double()
        if dual:
            z = z + v2.scale.double() * y2.double() + v2.shift.double()
        ref = (Eref @ w.double()) * (z > 0)
        ps = part.sum(0)
        yh1 = (y1.double() - v1.mean.double()) * v1.invstd.double()
        line = ["mask dual=%d" % dual, rows, kdim, cols, "dz", rel(dz, ref), "s0", rel(ps[0], ref.sum(0)), "s1", rel(ps[1], (ref * yh1).sum(0))]
        if dual:
            yh2 = (y2.double() - v2.mean.double()) * v2.invstd.double()
            line += ["s2", rel(ps[2], (ref * yh2).sum(0))]
        print(*line, flush=True)
    # wgrad
    for (pm, qm) in ((H.OP_AFF2, H.OP_RELU1), (H.OP_POOLED, H.OP_RELU2), (H.OP_BCAST, H.OP_ID), (H.OP_AFF2, H.OP_ID)):
        Pp, Pref, k1 = build_operand(pm, rows, cols, ns)
        Qq, Qref, k2 = build_operand(qm, rows, kdim, ns)
        dw = H.wgrad(rows, cols, kdim, Pp, Qq, dev)
        print("wgrad", names[pm], names[qm], rows, cols, kdim, rel(dw, Pref.t() @ Qref), flush=True)
bad = [r for r in results if not (r[-1] < 1e-4)]
print("gemm_store cases", len(results), "bad", len(bad))
