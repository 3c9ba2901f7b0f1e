#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REAL reference (hancyran/RepSurf).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
It imports the reference's classification CPU/PyTorch path (`cuda=False`, the oracle the
north-star names), after stubbing the CUDA extension module the reference insists on importing
(classification/modules/pointnet2_utils.py:8-12), runs it on seeded synthetic clouds and stores
inputs + outputs as small .npz files.  Nothing here is used at run time on the GPU box.

Fixtures:
  geom_seed{S}.npz   B=2 x 1024 uniform clouds: FPS / ball-query / kNN indices for the radii and
                     nsample of the shipped models, umbrella features (pre-MLP) for cloud 0.
  geom_real.npz      first 1024 rows of the reference's visualization/*.txt clouds (non-uniform
                     density: balls overflow nsample, duplicate-free).
  model_b4.npz       RepSurf-U repsurf_ssg_umb, B=4: logits, loss, stage outputs (subsampled),
                     parameter-gradient norms and three full gradients, with name-seeded weights
                     and dropout disabled.
  probe.json         torch build info + the CPU-kernel rounding-order probe the oracle relies on.
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/classification"


def import_reference():
    sys.modules.setdefault("pointops_cuda", types.ModuleType("pointops_cuda"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models.repsurf import repsurf_ssg_umb  # noqa
    from modules import pointnet2_utils, repsurface_utils  # noqa
    from util.utils import SmoothClsLoss  # noqa
    return repsurf_ssg_umb, pointnet2_utils, repsurface_utils, SmoothClsLoss


def ref_args(**over):
    ns = argparse.Namespace(num_point=1024, return_dist=True, return_center=True, return_polar=True,
                            group_size=8, umb_pool="sum", cuda_ops=False, num_class=15)
    for k, v in over.items():
        setattr(ns, k, v)
    return ns


def make_cloud(seed, b, n):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(b, n, 3, generator=g) * 2 - 1).contiguous()


def geom_fixture(P, R, xyz, seed):
    """indices through the reference's own functions, with the RNG draws recorded"""
    out = {"xyz": xyz.numpy()}
    b, n, _ = xyz.shape
    torch.manual_seed(seed)
    st1 = torch.randint(0, n, (b,), dtype=torch.long)        # what pointnet2_utils.py:66 will draw
    torch.manual_seed(seed)
    fps1 = P.farthest_point_sample(xyz, 512)
    assert torch.equal(fps1[:, 0], st1)
    out["fps1_start"] = st1.numpy().astype(np.int32)
    out["fps1"] = fps1.numpy().astype(np.int16)
    c1 = P.index_points(xyz, fps1)
    for tag, (r, ns) in {"ball_r02_ns32": (0.2, 32), "ball_r01_ns24": (0.1, 24)}.items():
        out[tag] = P.query_ball_point(r, ns, xyz, c1).numpy().astype(np.int16)
    torch.manual_seed(seed + 1000)
    st2 = torch.randint(0, 512, (b,), dtype=torch.long)
    torch.manual_seed(seed + 1000)
    fps2 = P.farthest_point_sample(c1, 128)
    out["fps2_start"] = st2.numpy().astype(np.int32)
    out["fps2"] = fps2.numpy().astype(np.int16)
    c2 = P.index_points(c1, fps2)
    out["ball2_r04_ns64"] = P.query_ball_point(0.4, 64, c1, c2).numpy().astype(np.int16)
    knn = P.query_knn_point(9, xyz, xyz)
    out["knn9"] = knn.numpy().astype(np.int16)
    # exact distance ties inside the top-10 make torch's (unstable) sort order arbitrary: flag them
    d = P.square_distance(xyz, xyz)
    top = d.sort(dim=-1)[0][:, :, :10]
    out["knn9_tie_rows"] = (top[:, :, 1:] == top[:, :, :-1]).any(-1).numpy()
    # umbrella features of cloud 0, captured at the input of UmbrellaSurfaceConstructor.mlps
    usc = R.UmbrellaSurfaceConstructor(9, 10, return_dist=True, aggr_type="sum", cuda=False)
    grabbed = {}
    usc.mlps.register_forward_pre_hook(lambda m, inp: grabbed.__setitem__("x", inp[0].detach().clone()))
    torch.manual_seed(seed + 2000)
    inv = torch.randint(0, 2, (1, 1, 1)).float() * 2 - 1     # what recons_utils.py:50 will draw
    torch.manual_seed(seed + 2000)
    with torch.no_grad():
        usc(xyz[:1].permute(0, 2, 1).contiguous())
    out["umb_inv_sign"] = inv.view(1).numpy()
    out["umb_feat"] = grabbed["x"].permute(0, 3, 2, 1).contiguous().numpy()   # (1, N, 8, 10)
    return out


def name_seeded_init(model):
    """Weights that depend only on parameter NAMES and shapes (so the reference model and ours get
    identical values without sharing construction order).  Mirrors tests/util.py."""
    import zlib
    with torch.no_grad():
        for name, p in sorted(model.named_parameters()):
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) / fan_in ** 0.5)
            elif name.endswith("weight"):      # BatchNorm gamma
                p.copy_(0.75 + 0.5 * torch.rand(p.shape, generator=g))
            else:                              # biases / BatchNorm beta
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.1)


def model_fixture(mod, SmoothClsLoss, b=4, seed=7):
    model = mod.Model(ref_args())
    name_seeded_init(model)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    xyz = make_cloud(seed, b, 1024)
    g = torch.Generator().manual_seed(seed + 1)
    label = torch.randint(0, 15, (b,), generator=g)
    grabbed = {}
    for nm in ("surface_constructor", "sa1", "sa2", "sa3"):
        getattr(model, nm).register_forward_hook(
            lambda m, i, o, nm=nm: grabbed.__setitem__(nm, o if torch.is_tensor(o) else o[2]))
    torch.manual_seed(seed + 2)
    rng_state = torch.get_rng_state()
    inv = torch.randint(0, 2, (b, 1, 1)).float() * 2 - 1
    st1 = torch.randint(0, 1024, (b,), dtype=torch.long)
    st2 = torch.randint(0, 512, (b,), dtype=torch.long)
    torch.set_rng_state(rng_state)
    pred = model(xyz.permute(0, 2, 1).contiguous())
    loss = SmoothClsLoss()(pred, label)
    loss.backward()
    out = {"xyz": xyz.numpy(), "label": label.numpy().astype(np.int32), "rng_seed": np.int64(seed + 2),
           "inv_sign": inv.view(b).numpy(), "fps1_start": st1.numpy().astype(np.int32),
           "fps2_start": st2.numpy().astype(np.int32),
           "logits": pred.detach().numpy(), "loss": np.float32(loss.item()),
           "normal": grabbed["surface_constructor"].detach().numpy(),            # (B,10,N)
           "sa1_feat_sub": grabbed["sa1"].detach()[:, :, ::8].contiguous().numpy(),   # (B,128,64)
           "sa2_feat_sub": grabbed["sa2"].detach()[:, :, ::4].contiguous().numpy(),   # (B,256,32)
           "sa3_feat": grabbed["sa3"].detach().numpy()}
    names, norms = [], []
    for name, p in sorted(model.named_parameters()):
        names.append(name)
        norms.append(float(p.grad.norm()))
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, np.float32)
    for name in ("surface_constructor.mlps.0.weight", "sa1.mlp_l0.weight", "sa2.mlp_convs.0.bias",
                 "sa3.bn_f0.weight", "classfier.8.weight"):
        out["grad::" + name] = dict(model.named_parameters())[name].grad.numpy()
    out["bn_running_mean::sa1.bn_l0"] = model.sa1.bn_l0.running_mean.numpy()
    out["bn_running_var::sa1.bn_l0"] = model.sa1.bn_l0.running_var.numpy()
    return out


def probe():
    def f32(a):
        return a.astype(np.float32)
    torch.manual_seed(0)
    x = torch.rand(4, 1024, 3) * 2 - 1
    xn = x.numpy()
    res = {}
    s = torch.sum(x ** 2, -1).numpy()
    res["sum3_is_(a+b)+c"] = float((s == (xn[..., 0] * xn[..., 0] + xn[..., 1] * xn[..., 1]) + xn[..., 2] * xn[..., 2]).mean())
    q = x[:, :512]
    mm = torch.matmul(q, x.permute(0, 2, 1)).numpy()
    qd, xd = q.numpy().astype(np.float64), xn.astype(np.float64)
    p0 = f32(qd[:, :, None, 0] * xd[:, None, :, 0])
    c1 = f32(qd[:, :, None, 1] * xd[:, None, :, 1] + p0.astype(np.float64))
    c2 = f32(qd[:, :, None, 2] * xd[:, None, :, 2] + c1.astype(np.float64))
    res["matmul_k3_is_fma_chain"] = float((mm == c2).mean())
    a, b = torch.rand(100000, 3) * 2 - 1, torch.rand(100000, 3) * 2 - 1
    cr = torch.cross(a, b, dim=-1).numpy()
    ad, bd = a.numpy().astype(np.float64), b.numpy().astype(np.float64)
    res["cross_is_fma(a1*b2,-(a2*b1))"] = float((cr[:, 0] == f32(ad[:, 1] * bd[:, 2] - f32(ad[:, 2] * bd[:, 1]).astype(np.float64))).mean())
    crd = cr.astype(np.float64)
    nr = torch.norm(torch.from_numpy(cr), dim=-1).numpy()
    fm = f32(np.sqrt(f32(crd[:, 2] ** 2 + f32(crd[:, 1] ** 2 + f32(crd[:, 0] ** 2).astype(np.float64)).astype(np.float64))))
    res["norm3_is_sqrt(fma(z,z,fma(y,y,x*x)))"] = float((nr == fm).mean())
    th = torch.rand(100000) * 3.14
    res["div_pi_is_div_by_float32(pi)"] = float(((th / np.pi).numpy() == th.numpy() / np.float32(np.pi)).mean())
    small = torch.randint(0, 3, (5000, 8)).float()
    res["argsort8_is_stable"] = bool((small.argsort(dim=-1).numpy() == small.sort(dim=-1, stable=True)[1].numpy()).all())
    big = torch.randint(0, 50, (64, 1024)).float()
    res["max_returns_first_index"] = bool((torch.max(big, -1)[1].numpy() == big.numpy().argmax(-1)).all())
    return {"torch": torch.__version__, "threads": torch.get_num_threads(),
            "config": torch.__config__.show().splitlines()[:12], "probe": res}


def main():
    mod, P, R, SmoothClsLoss = import_reference()
    for seed in (0, 1, 2, 3):
        fx = geom_fixture(P, R, make_cloud(seed, 2, 1024), seed)
        np.savez_compressed(os.path.join(HERE, f"geom_seed{seed}.npz"), **fx)
        print("geom seed", seed, "near-dup kNN tie rows:", int(fx["knn9_tie_rows"].sum()))
    clouds = []
    for nm in ("airplane_0001", "bed_0001", "cup_0001", "table_0250"):
        a = np.loadtxt(f"/root/reference/visualization/{nm}.txt", delimiter=",", dtype=np.float32)
        clouds.append(a[:1024, :3])
    real = torch.from_numpy(np.stack(clouds)).contiguous()
    fx = geom_fixture(P, R, real, 11)
    np.savez_compressed(os.path.join(HERE, "geom_real.npz"), **fx)
    print("geom real tie rows:", int(fx["knn9_tie_rows"].sum()))
    np.savez_compressed(os.path.join(HERE, "model_b4.npz"), **model_fixture(mod, SmoothClsLoss))
    with open(os.path.join(HERE, "probe.json"), "w") as f:
        json.dump(probe(), f, indent=1)
    print("done")


if __name__ == "__main__":
    main()
