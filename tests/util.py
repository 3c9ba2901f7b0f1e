"""Shared test helpers."""
import argparse
import contextlib
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def ref_args(**over):
    """The canonical flag set of classification/scripts/scanobjectnn/repsurf_ssg_umb.sh."""
    ns = argparse.Namespace(num_point=1024, return_dist=True, return_center=True, return_polar=True,
                            group_size=8, umb_pool="sum", cuda_ops=True, num_class=15)
    for k, v in over.items():
        setattr(ns, k, v)
    return ns


def name_seeded_init(model):
    """Weights that depend only on parameter names/shapes (same rule as tests/golden/make_golden.py)."""
    with torch.no_grad():
        for name, p in sorted(model.named_parameters()):
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
            if p.dim() >= 2:
                fan_in = p[0].numel()
                v = (torch.rand(p.shape, generator=g) * 2 - 1) / fan_in ** 0.5
            elif name.endswith("weight"):
                v = 0.75 + 0.5 * torch.rand(p.shape, generator=g)
            else:
                v = (torch.rand(p.shape, generator=g) * 2 - 1) * 0.1
            p.copy_(v.to(p.device))


def disable_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0


def cloud(seed, b, n, kind="uniform"):
    r = np.random.RandomState(seed)
    if kind == "uniform":
        return (r.rand(b, n, 3) * 2 - 1).astype(np.float32)
    if kind == "clustered":   # dense blobs: balls overflow nsample
        c = r.rand(b, 8, 3) * 2 - 1
        pick = r.randint(0, 8, (b, n))
        return (np.take_along_axis(c, pick[..., None].repeat(3, -1), 1) + 0.05 * r.randn(b, n, 3)).astype(np.float32)
    if kind == "grid":        # lattice: many exactly equal distances (tie rules)
        side = int(round(n ** (1 / 3))) + 1
        g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
        out = np.stack([g[r.permutation(n)] for _ in range(b)]).astype(np.float32) / side
        return out
    if kind == "dup":         # duplicated points: zero distances, degenerate fan triangles
        base = (r.rand(b, n // 2, 3) * 2 - 1).astype(np.float32)
        return np.concatenate([base, base], 1)[:, r.permutation(2 * (n // 2))]
    raise ValueError(kind)


def take(points, idx):
    """numpy gather: points (B,N,C), idx (B,...) -> (B,...,C)"""
    b = points.shape[0]
    flat = idx.reshape(b, -1).astype(np.int64)
    out = np.take_along_axis(points, flat[..., None].repeat(points.shape[2], -1), 1)
    return out.reshape(*idx.shape, points.shape[2])


def is_pre_bn_bias(name):
    """Biases added right before a BatchNorm: their gradient is analytically zero."""
    if not name.endswith(".bias"):
        return False
    return (".mlp_l0." in name or ".mlp_f0." in name or ".mlp_convs." in name
            # mlps.6.bias shifts every normal channel by a constant, which each stage's bn_f0 removes
            or name in ("surface_constructor.mlps.3.bias", "surface_constructor.mlps.6.bias",
                        "classfier.0.bias", "classfier.4.bias"))


def seg_args(**over):
    """The flag set of segmentation/scripts/s3dis/train_repsurf_umb.sh as the model reads it."""
    ns = argparse.Namespace(return_polar=False, in_channel=6, group_size=8, num_class=13)
    for k, v in over.items():
        setattr(ns, k, v)
    return ns


def seg_param_shapes(args=None):
    """(name, shape) of every parameter of segmentation/models/repsurf/repsurf_umb_ssg.py:13-41."""
    a = args or seg_args()
    cc = 6 if a.return_polar else 3
    out = []

    def lin(name, cin, cout, conv=False):
        out.append((name + ".weight", (cout, cin, 1) if conv else (cout, cin)))
        out.append((name + ".bias", (cout,)))

    def bn(name, c):
        out.append((name + ".weight", (c,)))
        out.append((name + ".bias", (c,)))

    for i, (cin, mlp) in enumerate([(a.in_channel + 10, [32, 32, 64]), (64 + 10, [64, 64, 128]),
                                    (128 + 10, [128, 128, 256]), (256 + 10, [256, 256, 512])], 1):
        pre = f"sa{i}"
        lin(pre + ".mlp_l0", cc, mlp[0], True); lin(pre + ".mlp_f0", cin, mlp[0], True)
        bn(pre + ".bn_l0", mlp[0]); bn(pre + ".bn_f0", mlp[0])
        last = mlp[0]
        for j, w in enumerate(mlp[1:]):
            lin(f"{pre}.mlp_convs.{j}", last, w, True); bn(f"{pre}.mlp_bns.{j}", w)
            last = w
    for name, prev, skip, mlp in [("fp4", 512, 256, [256, 256]), ("fp3", 256, 128, [256, 256]),
                                  ("fp2", 256, 64, [256, 128]), ("fp1", 128, None, [128, 128, 128])]:
        lin(name + ".mlp_f0", prev, mlp[0]); bn(name + ".norm_f0", mlp[0])
        if skip is not None:
            lin(name + ".mlp_s0", skip, mlp[0]); bn(name + ".norm_s0", mlp[0])
        last = mlp[0]
        for j, w in enumerate(mlp[1:]):
            lin(f"{name}.mlp_convs.{j}", last, w); bn(f"{name}.mlp_bns.{j}", w)
            last = w
    lin("classifier.0", 128, 128); bn("classifier.1", 128); lin("classifier.4", 128, a.num_class)
    lin("surface_constructor.mlps.0", 10, 10, True); bn("surface_constructor.mlps.1", 10)
    lin("surface_constructor.mlps.3", 10, 10, True)
    return out


def seg_state(args=None):
    """Name-seeded weights of the segmentation model as a state dict (same rule as name_seeded_init /
    tests/golden/make_golden_seg.py), without instantiating any module."""
    state = {}
    for name, shape in sorted(seg_param_shapes(args)):
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = (torch.rand(shape, generator=g) * 2 - 1) / fan_in ** 0.5
        elif name.endswith("weight"):
            v = 0.75 + 0.5 * torch.rand(shape, generator=g)
        else:
            v = (torch.rand(shape, generator=g) * 2 - 1) * 0.1
        state[name] = v
    return state


_SUB_PACKAGES = ("modules", "models", "util")
_SUB_CACHE = {}


@contextlib.contextmanager
def subproject(name):
    """Import context of one of the reference's sub-projects (`classification` / `segmentation`): both ship
    top-level packages called `modules` and `models`, resolved relative to the sub-project root on sys.path
    (PYTHONPATH=./ in the reference's scripts), so only one can be live at a time."""
    root = os.path.join(ROOT, "repsurf_amd", name)
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in _SUB_PACKAGES}
    for k in saved:
        del sys.modules[k]
    sys.modules.update(_SUB_CACHE.setdefault(name, {}))
    sys.path.insert(0, root)
    try:
        yield root
    finally:
        sys.path.remove(root)
        now = {k: v for k, v in sys.modules.items() if k.split(".")[0] in _SUB_PACKAGES}
        _SUB_CACHE[name].update(now)
        for k in now:
            del sys.modules[k]
        sys.modules.update(saved)


def parity_report(test, **numbers):
    """Append the MEASURED errors of a parity test to gpurun_out/parity_report.jsonl (scratch; the round's copy is
    committed under profiles/): assertions say pass/fail, this says by how much."""
    import json
    path = os.environ.get("REPSURF_PARITY_REPORT", os.path.join(ROOT, "gpurun_out", "parity_report.jsonl"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps({"test": test, "gemm_products": "fp32_mfma" if os.environ.get("RS_GEMM_SPLIT3", "1") == "0" else "bf16x3_split",
                                **{k: _num(v) for k, v in numbers.items()}}) + "\n")
    except OSError:
        pass
    print("parity", test, {k: (float("%.3g" % v) if isinstance(_num(v), float) else v) for k, v in numbers.items()})


def _num(v):
    return v if isinstance(v, (str, list, dict, bool)) or v is None else float(v)


def staged_reference_file(sub, rel):
    """Path of an UNMODIFIED reference source file needed to execute the drop-in claim (the model definitions):
    /root/reference/<sub>/<rel> in the build container, else the copy oracle/Makefile.ref stages under the git-ignored
    oracle/_ref/dropin/ (it travels to the GPU box with the built .so files; it is never part of the history)."""
    for base in ("/root/reference", os.path.join(ROOT, "oracle", "_ref", "dropin")):
        p = os.path.join(base, sub, rel)
        if os.path.exists(p):
            return p
    return None


def load_by_path(name, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def three_way(got, ref32, ref64, rel_l2=False):
    """Three-way comparison of one tensor: the HIP result and the fp32 CPU oracle, each against the SAME network evaluated in
    float64 (the truth leg).  Returns (error of the HIP path, error of the fp32 oracle, scale of the truth).  With
    rel_l2 the errors are relative L2 norms (gradients), otherwise maximum absolute differences (activations)."""
    got, ref32, ref64 = (np.asarray(a, np.float64).reshape(-1) for a in (got, ref32, ref64))
    if rel_l2:
        nrm = max(np.linalg.norm(ref64), 1e-300)
        return np.linalg.norm(got - ref64) / nrm, np.linalg.norm(ref32 - ref64) / nrm, nrm
    return np.abs(got - ref64).max(), np.abs(ref32 - ref64).max(), np.abs(ref64).max()


def gradient_noise_check(table, per_tensor=4.0, median=2.0, floor=1e-4):
    """table: {parameter: (relative-L2 error of the HIP gradient, of the fp32 reference gradient)}, both against the float64
    evaluation of the same network.  Activations are continuous in the rounding noise and are held to 1.5 x the reference's
    error elsewhere; GRADIENTS are not: a ReLU input or a max-pool runner-up within ~1e-6 of the decision flips between two
    valid fp32 evaluations and re-routes a whole row, so a tensor's error is a handful of such quanta and varies by 2-4 x
    between two correct fp32 runs (profiles/r03/seg_fp32_spread.txt: the SAME CPU oracle on two hosts: median 0.0044 / 0.0078,
    max 0.012 / 0.018; with torch's fp32 BatchNorm kernel 0.020 / 0.10; single tensors move by 3x between the two hosts --
    surface_constructor.mlps.1.weight at configs[3]: oracle 0.0060 on the GPU host, 0.0184 on the build host, HIP 0.0194).
    The statement is therefore: every tensor within
    `per_tensor` x the larger of its own reference error and the network's typical (median) reference error, and the median
    over tensors within `median` x the reference's median.  -> list of offending (name, hip, ref)."""
    ref_med = float(np.median([v[1] for v in table.values()]))
    hip_med = float(np.median([v[0] for v in table.values()]))
    bad = [(k, v[0], v[1]) for k, v in table.items() if v[0] > per_tensor * max(v[1], ref_med) + floor]
    if hip_med > median * ref_med + floor:
        bad.append(("<median over tensors>", hip_med, ref_med))
    return bad
