#!/usr/bin/env python3
"""Gradients of the intermediate activations: the network run eagerly under a ragged.Capacity (capacity-sized buffers) against the plain
eager pass on the same batch (GPU box)."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.test_seg_gpu import _seg_model, packed_cloud, dev
from tests.util import subproject
from repsurf_amd import ops as _ops, mlp_hip
from repsurf_amd.graph import RaggedSegStep
from repsurf_amd.head import CrossEntropyLoss
cuda = torch.device("cuda")
layouts = [[1024, 700, 513, 900], [512, 512, 900, 640]]
batches, labels = [], []
for seed, sizes in enumerate(layouts):
    xyz, _ = packed_cloud(20 + seed, sizes)
    r = np.random.RandomState(40 + seed)
    n = sum(sizes)
    batches.append([dev(xyz), dev(r.rand(n, 3).astype(np.float32)), _ops.offsets_tensor(np.cumsum(sizes).tolist(), cuda)])
    lab = r.randint(0, 13, n).astype(np.int64)
    lab[r.rand(n) < 0.05] = 255
    labels.append(dev(lab))
crit = CrossEntropyLoss(ignore_index=255)


def instrument(model, store):
    hooks = []
    for name in ("surface_constructor", "sa1", "sa2", "sa3", "sa4", "fp4", "fp3", "fp2", "fp1", "classifier.0", "classifier.3"):
        mod = model
        for part in name.split("."):
            mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]

        def fwd_hook(m, inp, out, name=name):
            t = out
            if isinstance(out, (list, tuple)):
                t = out[2]
            if isinstance(t, mlp_hip.LazyRows):
                t = t.y
            if torch.is_tensor(t) and t.requires_grad:
                store[name + ".out"] = t.detach()
                t.register_hook(lambda g, name=name: store.__setitem__(name + ".grad", g.detach().clone()))
        hooks.append(mod.register_forward_hook(fwd_hook))
    return hooks


_orig_interp = _ops.three_interpolate_add_relu
_cur = {"store": None}


def _wrapped(points, idx, weight, add=None, csr=None):
    out = _orig_interp(points, idx, weight, add, csr)
    st = _cur["store"]
    if st is not None:
        points.register_hook(lambda g: st.__setitem__("interp.points.grad", g.detach().clone()[0]))
        out.register_hook(lambda g: st.__setitem__("interp.out.grad", g.detach().clone()[0]))
        st["interp.idx"] = idx.detach()[0]
        st["interp.weight"] = weight.detach()[0]
    return out


_ops.three_interpolate_add_relu = _wrapped
with subproject("segmentation"):
    eager = _seg_model()
    eager.surface_constructor.random_inv = False
    twin = copy.deepcopy(eager)
    step = RaggedSegStep(twin, crit, None, batches[0], labels[0], capacity=4 * 1024)
    step(batches[1], labels[1])          # batch 1 now sits in parity 1
    torch.cuda.synchronize()
    q = 1
    se, sr = {}, {}
    instrument(eager, se)
    instrument(twin, sr)
    for p in list(eager.parameters()) + list(twin.parameters()):
        p.grad = None
    _cur["store"] = se
    mlp_hip.DEBUG = {"log": []}
    le = crit(eager(batches[1]), labels[1])
    le.backward()
    log_e = mlp_hip.DEBUG["log"]
    mlp_hip.DEBUG = {"log": []}
    _cur["store"] = sr
    if os.environ.get("POISON", "1") == "1":      # every torch.empty of the capacity run starts as NaN: a reduction that reads a row it must not read shows
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True
        n0 = step.counts[q][0]
        step.coord[q][n0:] = float("nan")
        step.feat[q][n0:] = float("nan")
        st = step.state[q]
        st.feat[n0:] = float("nan")
        for li, g in enumerate(st.stages):
            g.new_center[step.counts[q][li + 1]:] = float("nan")
        for (fine, _), f in zip(((3, 4), (2, 3), (1, 2), (0, 1)), st.fps):
            f[1][step.counts[q][fine]:] = float("nan")
    with step.caps[q]:
        lr = crit(twin([step.coord[q], step.feat[q], step.offset], geo=step.state[q]), step.label[q])
        lr.backward()
    torch.cuda.synchronize()
    print("loss", le.item(), lr.item(), "counts", step.counts[q])
    for k in se:
        a, b = se[k], sr.get(k)
        if b is None:
            print(k, "missing")
            continue
        n = a.shape[0]
        bb = b[:n]
        if a.dtype != torch.float32:
            print(f"{k:28s} rows {n}/{b.shape[0]} equal {bool((a == bb).all())}")
            continue
        if not torch.isfinite(bb).all():
            print(f"{k:28s} rows {n:6d}/{b.shape[0]:6d}  NON-FINITE in valid rows: {int((~torch.isfinite(bb)).any(-1).sum() if bb.dim() > 1 else (~torch.isfinite(bb)).sum())} rows")
            continue
        print(f"{k:28s} rows {n:6d}/{b.shape[0]:6d}  rel-L2 {float((a.double() - bb.double()).norm() / max(float(a.double().norm()), 1e-30)):.2e}  |a| {float(a.norm()):.3e}")
    a, b = se["interp.out.grad"], sr["interp.out.grad"][:se["interp.out.grad"].shape[0]]
    err = (a.double() - b.double()).norm(dim=1) / a.double().norm(dim=1).clamp_min(1e-12)
    print("interp.out.grad per-row relative error: median", float(err.median()), "max", float(err.max()), "rows > 1e-4:", int((err > 1e-4).sum()))
    ratio = (b.double() * a.double()).sum(1) / (a.double() * a.double()).sum(1).clamp_min(1e-30)
    print("   projection b.a / a.a: median", float(ratio.median()), "min", float(ratio.min()), "max", float(ratio.max()))
    d = (b - a).double()
    print("   column-wise mean of (b - a):", d.mean(0)[:6].tolist(), " std over rows:", d.std(0)[:6].tolist())
    names = [n for n, _ in eager.named_parameters()]
    for nm, pe, pt in zip(names, eager.parameters(), twin.parameters()):
        if nm.startswith(("fp1.", "classifier.")) and pe.grad is not None and float(pe.grad.norm()) > 0:
            print(f"   {nm:32s} {float((pe.grad.double() - pt.grad.double()).norm() / pe.grad.double().norm()):.2e}")
    bad = torch.nonzero(err > 1e-4).flatten().tolist()
    print("bad rows:", bad[:40], "...", bad[-10:])

    log_r = mlp_hip.DEBUG["log"]
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(log_e[:3], log_r[:3])):
        n = a["rows"]
        print(f"stack-layer log {i}: li {a['li']} rows {a['rows']} / {b['rows']} full {a['full']} / {b['full']}")
        for k in ("p", "q", "r", "dg", "db"):
            print(f"     {k}: rel {float((a[k].double() - b[k].double()).norm() / a[k].double().norm().clamp_min(1e-30)):.2e}   first: {a[k][:3].tolist()} | {b[k][:3].tolist()}")
        print(f"     dz rel {float((a['dz'].double() - b['dz'][:n].double()).norm() / a['dz'].double().norm()):.2e}; part sums rel {float((a['part'].sum(0) - b['part'].sum(0)).norm() / a['part'].sum(0).norm()):.2e}  part rows {a['part'].shape[0]} / {b['part'].shape[0]}")
    a, b = log_e[0]["dz"], log_r[0]["dz"][:log_e[0]["rows"]]
    rowerr = (a.double() - b.double()).norm(dim=1) / a.double().norm(dim=1).clamp_min(1e-20)
    bad = torch.nonzero(rowerr > 1e-4).flatten()
    print("dz bad rows:", bad.numel(), bad[:20].tolist())
    r0 = int(bad[0])
    za, zb = a[r0], b[r0]
    print("  row", r0, "nonzeros eager/capacity:", int((za != 0).sum()), int((zb != 0).sum()), " mask differs in", int(((za != 0) != (zb != 0)).sum()), "channels")
    both = (za != 0) & (zb != 0)
    print("  where both nonzero: max rel diff", float(((za[both] - zb[both]).abs() / za[both].abs().clamp_min(1e-20)).max()))
    print("  eager  :", za[:8].tolist())
    print("  capac. :", zb[:8].tolist())
    tot_mask = ((a != 0) != (b != 0)).sum().item()
    print("  total mask differences over the tensor:", tot_mask, "of", a.numel())
    d = (a - b)[(a != 0) & (b != 0)]
    print("  value differences where both nonzero: max abs", float(d.abs().max()), "vs typical |dz|", float(a.abs().mean()))

    for tag, log, model, store in (("eager", log_e, eager, se), ("capacity", log_r, twin, sr)):
        L = log[0]
        n = log_e[0]["rows"]
        W0 = model.fp1.mlp_convs[0].weight.detach().double()                      # (out, in)
        E = L["p"].double() * L["dz"][:n].double() + L["q"].double() * L["y"][:n].double() + L["r"].double()
        want = E @ W0
        got = store["interp.out.grad"][:n].double()
        e = (want - got).norm(dim=1) / want.norm(dim=1).clamp_min(1e-30)
        print(f"{tag}: dx against (p dz + q y + r) . W0 in fp64: rel-L2 {float((want - got).norm() / want.norm()):.2e}; rows > 1e-4: {int((e > 1e-4).sum())}")
