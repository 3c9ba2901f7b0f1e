#!/usr/bin/env python3
"""Generate tests/golden/seg_*.npz from the REAL reference's segmentation code (hancyran/RepSurf).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_seg.py

The reference's segmentation path is CUDA-only: `pointops_cuda.furthestsampling_cuda` / `knnquery_cuda` are
compiled CUDA kernels and the Python side allocates with `torch.cuda.IntTensor/FloatTensor`
(segmentation/modules/pointops/functions/pointops.py:42-44,125-127, repsurface_utils.py:22,268).  Here
  * `pointops_cuda` is oracle/_ref: the reference's OWN `*_cuda_kernel.cu` files compiled unmodified as host code
    (oracle/Makefile.ref, oracle/ref_pointops.py) — the FPS tree reduction and the kNN heap run as written, and
  * `torch.cuda.IntTensor/FloatTensor` are replaced by CPU constructors,
so that every line of the reference's own code (pointops.furthestsampling / sectorized_fps / knnquery,
sample_and_group, group_by_umbrella_v2, cal_normal, check_nan_umb, SurfaceAbstractionCD,
SurfaceFeaturePropagationCD, Model.forward) executes unmodified on CPU.  The fixtures pin the whole path:
sampled rows, neighbour lists, feature order, rotation, NaN patching, numpy-RNG flips, the dual first layer,
interpolation weights, the decoder wiring.

Fixtures:
  seg_geom.npz    3 packed clouds of unequal size: umbrella features (pre-MLP) through the reference
                  functions, sample_and_group outputs of one stage, interpolation weights.
  seg_sector.npz  pointops.sectorized_fps (pointops.py:52-108) on 3 packed clouds, num_sectors 1/2/4, min_points lowered
                  so that the sector branch runs on small clouds.
  seg_model.npz   repsurf_umb_ssg on 2 packed clouds (2048 + 1536 points): logits, loss, stage outputs
                  (subsampled), parameter-gradient norms + subsampled gradients; name-seeded weights, dropout 0.
  seg_cfg3.npz    (`--configs3`, ~10 minutes) the reference's own model at BASELINE configs[3] size -- 16 x 4096 uniform clouds, the
                  batch tests/test_parity_full_gpu.py::test_segmentation_step_at_the_benchmark_configuration_three_way builds --
                  in fp32 AND in float64: logits of every 16th row, loss, gradient norms of both runs: pins "the reference's own fp32
                  run is further than 1e-5 from its float64 evaluation" at the benchmark size, not only on the 2-cloud fixture.
  seg_pointnet2.npz  the reference's PointNet++ baseline (models/pointnet2/pointnet2_ssg.py over modules/pointnet2_utils.py)
                  on the same clouds: logits, loss, gradient norms + subsampled gradients.
"""
import argparse
import os
import sys
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/segmentation"

from oracle import ref_pointops  # noqa: E402


def install_stubs():
    ref_pointops.build()
    sys.modules["pointops_cuda"] = ref_pointops.module("seg")

    def ctor(dtype):
        class _T:
            def __new__(cls, *a):
                if len(a) == 1 and isinstance(a[0], (list, tuple)):
                    return torch.tensor(a[0], dtype=dtype)
                # float buffers the MODULES allocate (interpolation accumulators, pointnet2_utils.py:114,
                # repsurface_utils.py:268) follow FLOAT_DTYPE -- float64 in the truth runs below; buffers the operator file
                # hands to the kernels (pointops.py: FPS distances, kNN dist2) are always float32
                import inspect
                mine = dtype
                if dtype == torch.float32 and "pointops" not in inspect.stack()[1].filename:
                    mine = FLOAT_DTYPE[0]
                return torch.empty(*a, dtype=mine)
        return _T
    torch.cuda.IntTensor = ctor(torch.int32)
    torch.cuda.FloatTensor = ctor(torch.float32)
    if REF not in sys.path:
        sys.path.insert(0, REF)


FLOAT_DTYPE = [torch.float32]


def truth_run(build, inputs, label):
    """The reference's OWN code evaluated in float64 (same indices: coordinates and the kernels stay float32; parameters,
    features and every dense operation in double): the truth leg of the three-way parity tests.  -> logits, {name: grad}"""
    FLOAT_DTYPE[0] = torch.float64
    try:
        model = build().double()
        for mod in model.modules():      # float32 geometry (fan features, coordinate offsets) enters the double network exactly
            if isinstance(mod, (torch.nn.Conv1d, torch.nn.Linear)):
                mod.register_forward_pre_hook(lambda m_, a: tuple(t.double() for t in a))
        logits = model(inputs(torch.float64))
        loss = torch.nn.functional.cross_entropy(logits, label)
        loss.backward()
        return logits.detach().numpy(), float(loss.item()), {n: p.grad.detach() for n, p in model.named_parameters()}
    finally:
        FLOAT_DTYPE[0] = torch.float32


def name_seeded_init(model):
    with torch.no_grad():
        for name, p in sorted(model.named_parameters()):
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
            if p.dim() >= 2:
                v = (torch.rand(p.shape, generator=g) * 2 - 1) / p[0].numel() ** 0.5
            elif name.endswith("weight"):
                v = 0.75 + 0.5 * torch.rand(p.shape, generator=g)
            else:
                v = (torch.rand(p.shape, generator=g) * 2 - 1) * 0.1
            p.copy_(v)


def packed(seed, sizes):
    g = torch.Generator().manual_seed(seed)
    n = sum(sizes)
    coord = (torch.rand(n, 3, generator=g) * 2 - 1).contiguous()
    rgb = torch.rand(n, 3, generator=g).contiguous()
    offset = torch.tensor(np.cumsum(sizes), dtype=torch.int32)
    return coord, rgb, offset


def sub(t, step=7):
    return t.detach().reshape(-1)[::step].numpy().copy()


def main():
    install_stubs()
    from modules import repsurface_utils as R
    from modules.recons_utils import cal_const, cal_normal, cal_center, check_nan_umb
    from modules.polar_utils import xyz2sphere
    from modules.pointops.functions import pointops
    from models.repsurf.repsurf_umb_ssg import Model

    # ---------------- geometry fixture
    coord, rgb, offset = packed(11, [300, 512, 217])
    out = {"coord": coord.numpy(), "offset": offset.numpy()}
    for tag, fn in (("fix", R.group_by_umbrella_v2), ("none", R.group_by_umbrella)):
        np.random.seed(5)
        flips = np.random.rand(3) < 0.5                      # what cal_normal will draw (recons_utils.py:29)
        np.random.seed(5)
        gx = fn(coord, coord, offset, offset, k=9)
        nor = cal_normal(gx, offset, random_inv=True, is_group=True)
        cen = cal_center(gx)
        pol = xyz2sphere(cen)
        pos = cal_const(nor, cen)
        nor, cen, pos = check_nan_umb(nor, cen, pos)
        out[f"umb_{tag}"] = torch.cat([pol, nor, pos, cen], dim=-1).numpy()
        out[f"umb_{tag}_sign"] = np.where(flips, 1.0, -1.0).astype(np.float32)
    normal = torch.rand(coord.shape[0], 10, generator=torch.Generator().manual_seed(3))
    feat = torch.cat([coord, rgb], 1)
    for polar in (False, True):
        nc, nn_, nf, no = R.sample_and_group(4, 32, coord, normal, feat, offset, return_polar=polar, num_sector=1)
        t = "p" if polar else "x"
        out[f"sg_{t}_center"], out[f"sg_{t}_normal"], out[f"sg_{t}_feat"] = nc.numpy(), nn_.numpy(), nf.numpy()
        out[f"sg_{t}_offset"] = no.numpy()
    out["sg_normal_in"] = normal.numpy()
    out["sg_rgb"] = rgb.numpy()
    idx, dist = pointops.knnquery(3, nc, coord, no, offset)
    dr = 1.0 / (dist + 1e-8)
    out["interp_dist"] = dist.numpy()
    out["interp_weight"] = (dr / torch.sum(dr, dim=1, keepdim=True)).numpy()
    # raw kernel outputs through the reference's own autograd Functions (pointops.py:31-49,114-130)
    new_off = torch.tensor(np.cumsum([300 // 4, 512 // 4, 217 // 4]), dtype=torch.int32)
    out["fps_new_offset"] = new_off.numpy()
    out["fps_idx"] = pointops.furthestsampling(coord, offset, new_off).numpy()
    for k in (9, 32):
        idx, dist = pointops.knnquery(k, coord, coord, offset, offset)
        out[f"knn{k}_idx"], out[f"knn{k}_dist"] = idx.numpy(), dist.numpy()
    np.savez_compressed(os.path.join(HERE, "seg_geom.npz"), **out)

    # ---------------- sectorized FPS fixture (pointops.py:52-108), min_points lowered to reach the sector branch
    coord, _, offset = packed(31, [600, 1500, 1000])
    new_off = torch.tensor(np.cumsum([150, 375, 250]), dtype=torch.int32)
    out = {"coord": coord.numpy(), "offset": offset.numpy(), "new_offset": new_off.numpy(), "min_points": np.int32(800)}
    for ns in (1, 2, 4):
        out[f"idx_s{ns}"] = pointops.sectorized_fps(coord, offset, new_off, ns, 800).numpy().astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "seg_sector.npz"), **out)

    # ---------------- model fixture
    args = argparse.Namespace(return_polar=False, in_channel=6, group_size=8, num_class=13)
    torch.manual_seed(0)
    model = Model(args).train()
    name_seeded_init(model)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    coord, rgb, offset = packed(21, [2048, 1536])
    label = torch.randint(0, 13, (coord.shape[0],), generator=torch.Generator().manual_seed(4))
    np.random.seed(9)
    flips = np.random.rand(2) < 0.5
    np.random.seed(9)
    stage = {}
    hooks = [getattr(model, n).register_forward_hook(lambda mod, i, o, n=n: stage.__setitem__(n, list(o) if isinstance(o, list) else o))
             for n in ("sa1", "sa2", "sa3", "sa4", "fp1", "surface_constructor")]
    logits = model([coord, rgb, offset])
    loss = torch.nn.functional.cross_entropy(logits, label)
    loss.backward()
    for h in hooks:
        h.remove()
    out = {"coord": coord.numpy(), "rgb": rgb.numpy(), "offset": offset.numpy(), "label": label.numpy().astype(np.int16),
           "inv_sign": np.where(flips, 1.0, -1.0).astype(np.float32),
           "logits": logits.detach().numpy(), "loss": np.float32(loss.item()),
           "normal": stage["surface_constructor"].detach().numpy().astype(np.float32)}
    for n in ("sa1", "sa2", "sa3", "sa4"):
        out[n + "_center"] = stage[n][0].numpy()
        out[n + "_offset"] = stage[n][3].numpy()
        out[n + "_feat_sub"] = sub(stage[n][2])
    out["fp1_sub"] = sub(stage["fp1"])
    for name, p in model.named_parameters():
        out["shape/" + name] = np.array(p.shape, np.int32)
        out["gnorm/" + name] = np.float32(p.grad.norm().item())
        out["gsub/" + name] = sub(p.grad, 7 if p.numel() > 4096 else 1)
    out["buffers"] = np.array(sorted(n for n, _ in model.named_buffers()))

    def build_repsurf():
        m_ = Model(args).train()
        name_seeded_init(m_)
        for mod in m_.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        np.random.seed(9)
        return m_
    l64, loss64, g64 = truth_run(build_repsurf, lambda dt: [coord, rgb.clone().to(dt), offset], label)
    truth = {"logits64": l64, "loss64": np.float64(loss64)}
    for name, g in g64.items():
        truth["gsub64/" + name] = sub(g, 7 if g.numel() > 4096 else 1)
        truth["gnorm64/" + name] = np.float64(g.norm().item())
    np.savez_compressed(os.path.join(HERE, "seg_model_fp64.npz"), **truth)
    print("reference fp32 vs its own fp64: logits", np.abs(out["logits"] - l64).max())
    np.savez_compressed(os.path.join(HERE, "seg_model.npz"), **out)
    print("wrote seg_geom.npz, seg_model.npz; loss", loss.item())

    # ---------------- PointNet++ baseline over the same pointops boundary (models/pointnet2/pointnet2_ssg.py over
    # modules/pointnet2_utils.py:13-135): same clouds / labels / weight rule as the model fixture above
    from models.pointnet2.pointnet2_ssg import Model as PointNet2
    torch.manual_seed(0)
    pn = PointNet2(args).train()
    name_seeded_init(pn)
    for m in pn.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    logits = pn([coord, rgb.clone(), offset])
    loss = torch.nn.functional.cross_entropy(logits, label)
    loss.backward()
    out = {"coord": coord.numpy(), "rgb": rgb.numpy(), "offset": offset.numpy(), "label": label.numpy().astype(np.int16),
           "logits": logits.detach().numpy(), "loss": np.float32(loss.item())}
    for name, p in pn.named_parameters():
        out["shape/" + name] = np.array(p.shape, np.int32)
        out["gnorm/" + name] = np.float32(p.grad.norm().item())
        out["gsub/" + name] = sub(p.grad, 7 if p.numel() > 4096 else 1)

    def build_pn2():
        m_ = PointNet2(args).train()
        name_seeded_init(m_)
        for mod in m_.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        return m_
    l64, loss64, g64 = truth_run(build_pn2, lambda dt: [coord, rgb.clone().to(dt), offset], label)
    out["logits64"], out["loss64"] = l64, np.float64(loss64)
    for name, g in g64.items():
        out["gsub64/" + name] = sub(g, 7 if g.numel() > 4096 else 1)
    print("reference PointNet++ fp32 vs its own fp64: logits", np.abs(out["logits"] - l64).max())
    np.savez_compressed(os.path.join(HERE, "seg_pointnet2.npz"), **out)
    print("wrote seg_pointnet2.npz; loss", loss.item())


def configs3():
    """The reference's own model, fp32 and float64, on the configs[3] batch of the GPU parity test (same generator calls)."""
    import time
    install_stubs()
    from models.repsurf.repsurf_umb_ssg import Model
    args = argparse.Namespace(return_polar=False, in_channel=6, group_size=8, num_class=13)
    r = np.random.RandomState(3)
    n = 16 * 4096
    coord = torch.from_numpy((r.rand(n, 3) * 2 - 1).astype(np.float32))
    rgb = torch.from_numpy(r.rand(n, 3).astype(np.float32))
    offset = torch.from_numpy((np.arange(1, 17) * 4096).astype(np.int32))
    label = torch.from_numpy(r.randint(0, 13, n).astype(np.int64))

    def build():
        m_ = Model(args).train()
        name_seeded_init(m_)
        for mod in m_.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        np.random.seed(17)                     # cal_normal draws the 16 flips from here (recons_utils.py:29)
        return m_
    t0 = time.time()
    model = build()
    logits = model([coord, rgb.clone(), offset])
    loss = torch.nn.functional.cross_entropy(logits, label)
    loss.backward()
    print("fp32 run %.0f s, loss %.6f" % (time.time() - t0, loss.item()), flush=True)
    out = {"rows": np.arange(0, n, 16, dtype=np.int32), "logits32": logits.detach().numpy()[::16].copy(), "loss32": np.float32(loss.item())}
    for name, p in model.named_parameters():
        out["gnorm32/" + name] = np.float32(p.grad.norm().item())
    t0 = time.time()
    l64, loss64, g64 = truth_run(build, lambda dt: [coord, rgb.clone().to(dt), offset], label)
    print("fp64 run %.0f s, loss %.9f" % (time.time() - t0, loss64), flush=True)
    out["logits64"], out["loss64"] = l64[::16].copy(), np.float64(loss64)
    rel = {}
    for name, p in model.named_parameters():
        g = g64[name]
        out["gnorm64/" + name] = np.float64(g.norm().item())
        if g.norm().item() > 1e-5:
            rel[name] = float((p.grad.double() - g).norm() / g.norm())
            out["grel32/" + name] = np.float64(rel[name])       # relative L2 distance of the reference's fp32 gradient from its fp64 one
    out["logits32_vs_64_max_abs_all_rows"] = np.float64(np.abs(logits.detach().numpy() - l64).max())
    print("reference fp32 vs its own fp64 at configs[3]: logits max abs", out["logits32_vs_64_max_abs_all_rows"], "scale", np.abs(l64).max(),
          "gradient rel-L2 median / max", np.median(list(rel.values())), max(rel.values()))
    np.savez_compressed(os.path.join(HERE, "seg_cfg3.npz"), **out)


if __name__ == "__main__":
    if "--configs3" in sys.argv:
        configs3()
    else:
        main()
