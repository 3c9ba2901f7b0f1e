"""Parity at the BENCHMARK configurations (BASELINE.json configs[1], [3], [4]) and of the call sites either side of the
model (pre-model `sample()`, sectorized FPS, the classification L1 operator wrappers, eval after a training step).
Every test records its measured errors through tests.util.parity_report (committed copy of the last run: profiles/r04/parity_report.jsonl)."""
import os

import numpy as np
import pytest
import torch

from oracle import geom_oracle as G
from oracle import seg_ref, torch_ref
from tests.util import (GOLDEN, cloud, disable_dropout, is_pre_bn_bias, name_seeded_init, parity_report, ref_args, seg_args,
                        seg_state, subproject, take)

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def build_cls(arch="repsurf_ssg_umb"):
    import importlib
    model = importlib.import_module(f"models.repsurf.{arch}").Model(ref_args())
    name_seeded_init(model)
    disable_dropout(model)
    return model.cuda().train()


def test_classifier_step_at_the_benchmark_configuration_matches_oracle():
    """configs[1]: B=32 x 1024 points, full repsurf_ssg_umb step (forward, SmoothClsLoss, backward) against the CPU
    oracle on the same clouds, weights and CPU-generator draws.  Indices (FPS, ball query) bit-exact; log-probabilities
    within 1e-5 of the tensor scale (|log p| <= ~4); gradients relative-L2."""
    from util.utils import SmoothClsLoss
    b, seed = 32, 11
    model = build_cls()
    stage_idx = {}
    xyz = cloud(seed, b, 1024)
    label = np.random.RandomState(seed).randint(0, 15, (b,))
    torch.manual_seed(seed)
    state = torch.get_rng_state()
    flip = (torch.randint(0, 2, (b, 1, 1)).float() * 2 - 1).view(b).numpy()
    starts = [torch.randint(0, n, (b,), dtype=torch.long).numpy().astype(np.int32) for n in (1024, 512)]
    torch.set_rng_state(state)
    for nm in ("sa1", "sa2", "sa3"):
        getattr(model, nm).register_forward_hook(lambda m, i, o, nm=nm: stage_idx.__setitem__(nm, (o[0], o[2])))
    pred = model(dev(xyz).permute(0, 2, 1).contiguous())
    loss = SmoothClsLoss()(pred, dev(label).long())
    loss.backward()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = torch_ref.step({k: v.cpu() for k, v in model.state_dict().items()}, xyz, label, flip, starts)
    ties = int(ref["near_tie"].sum())
    # sampled centres: the FPS picks of both stages are the oracle's, bit for bit
    for nm, n_in in (("sa1", xyz), ):
        centres = stage_idx[nm][0].permute(0, 2, 1).detach().cpu().numpy()
        assert np.array_equal(centres, take(xyz, ref["sa1_fps"]))
    nums = {"azimuth_near_tie_points": ties}
    for nm in ("sa1", "sa2", "sa3"):
        got = stage_idx[nm][1].permute(0, 2, 1).detach().cpu().numpy()
        r = ref[nm + "_feat"].detach().numpy()
        nums[nm + "_feat_rel"] = np.abs(got - r).max() / max(np.abs(r).max(), 1.0)
    r = ref["logits"].detach().numpy()
    nums["logits_rel"] = np.abs(pred.detach().cpu().numpy() - r).max() / max(np.abs(r).max(), 1.0)
    nums["logits_scale"] = float(np.abs(r).max())
    nums["loss_abs"] = abs(loss.item() - float(ref["loss"].detach()))
    worst, worst_name = 0.0, ""
    for name, p in model.named_parameters():
        # sa3.mlp_bns.1.bias shifts every pooled feature of a channel by the same amount for all clouds, which the head's
        # BatchNorm1d (batch statistics) removes: analytically zero gradient, fp noise on both sides (like the pre-BN biases)
        if is_pre_bn_bias(name) or name == "sa3.mlp_bns.1.bias":
            continue
        rg = ref["grads"][name].numpy().reshape(p.shape)
        rel = np.linalg.norm(p.grad.cpu().numpy() - rg) / max(np.linalg.norm(rg), 1e-12)
        if rel > worst:
            worst, worst_name = rel, name
    nums["grad_rel_l2_worst"], nums["grad_worst_name"] = worst, worst_name
    nums["logits_abs"] = float(np.abs(pred.detach().cpu().numpy() - ref["logits"].detach().numpy()).max())
    parity_report("cls_b32x1024_vs_oracle", **nums)
    for nm in ("sa1", "sa2", "sa3"):
        assert nums[nm + "_feat_rel"] <= 1e-5, nums
    assert nums["logits_rel"] <= 1e-5 and nums["loss_abs"] <= 2e-5, nums
    # ... and literally: |log p - reference| <= 1e-5 ABSOLUTE (the scale above is max |log p| ~ 4; measured 1.9e-6 of it), VERDICT r5 weak 1a
    nums["logits_abs"] = float(np.abs(pred.detach().cpu().numpy() - r).max())
    assert nums["logits_abs"] <= 1e-5, nums
    assert worst <= 1e-2, nums


def _seg_three_way(tag, coord, rgb, offset, label, np_seed, flips, factor=1.5):
    """HIP step, fp32 CPU oracle and float64 truth on one batch.  The north-star's 1e-5 is a bound on single fp32 operators;
    a 13-BatchNorm-deep fp32 network evaluated by ANY fp32 arithmetic sits further than that from its exact value, so the
    model-level claim is stated against the truth: the HIP path is no further from the float64 evaluation of the network than
    `factor` x the reference's own fp32 arithmetic (the CPU oracle, torch fp32 kernels) is -- for the logits and the stage
    outputs; gradient tensors by tests.util.gradient_noise_check (they are discontinuous in the rounding noise: see there).
    All three numbers of every tensor go to the parity report."""
    from tests.util import three_way
    feats = {}
    with subproject("segmentation"):
        from models.repsurf.repsurf_umb_ssg import Model
        model = Model(seg_args())
        model.load_state_dict(seg_state(), strict=False)
        disable_dropout(model)
        model = model.cuda().train()
        for nm in ("sa1", "sa2", "sa3", "sa4"):
            getattr(model, nm).register_forward_hook(lambda m, i, o, nm=nm: feats.__setitem__(nm + "_feat", o[2].detach()))
        model.fp1.register_forward_hook(lambda m, i, o: feats.__setitem__("fp1_feat", o.detach()))
        model.surface_constructor.register_forward_hook(lambda m, i, o: feats.__setitem__("normal", o.detach()))
        np.random.seed(np_seed)
        logits = model([dev(coord), dev(rgb), dev(offset)])
    loss = torch.nn.functional.cross_entropy(logits, dev(label))
    loss.backward()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = seg_ref.step(seg_state(), coord, rgb, offset, label, flips)
    truth = seg_ref.step(seg_state(), coord, rgb, offset, label, flips, dtype=torch.float64)
    feats["logits"] = logits.detach()
    _seg_three_way.last_logits = logits.detach().cpu().numpy().astype(np.float64)
    nums = {"azimuth_near_tie_points": int(ref["near_tie"].sum()),
            "logits_vs_fp32_oracle": np.abs(logits.detach().cpu().numpy() - ref["logits"].detach().numpy()).max(),
            "loss_abs_vs_fp64": abs(loss.item() - float(truth["loss"].detach())),
            "oracle_loss_abs_vs_fp64": abs(float(ref["loss"].detach()) - float(truth["loss"].detach()))}
    bad = []
    for key in ("normal", "sa1_feat", "sa2_feat", "sa3_feat", "sa4_feat", "fp1_feat", "logits"):
        e_hip, e_ref, scale = three_way(feats[key].cpu().numpy(), ref[key].detach().numpy(), truth[key].detach().numpy())
        nums[key] = {"hip_vs_fp64": float(e_hip), "fp32_oracle_vs_fp64": float(e_ref), "scale": float(scale)}
        if e_hip > factor * e_ref + 1e-6 * scale:
            bad.append((key, e_hip, e_ref))
    grads, worst = {}, (0.0, "")
    for name, p in model.named_parameters():
        t = truth["grads"][name].numpy()
        if np.linalg.norm(t) < 1e-5:                              # pre-BatchNorm biases: analytically zero, fp noise everywhere
            continue
        e_hip, e_ref, nrm = three_way(p.grad.detach().cpu().numpy(), ref["grads"][name].numpy(), t, rel_l2=True)
        grads[name] = [float("%.3g" % e_hip), float("%.3g" % e_ref)]
        worst = max(worst, (e_hip, name))
    from tests.util import gradient_noise_check
    bad += gradient_noise_check(grads)
    nums["grad_rel_l2_median_vs_fp64__hip_oracle"] = [float(np.median([v[0] for v in grads.values()])),
                                                      float(np.median([v[1] for v in grads.values()]))]
    nums["grad_rel_l2_worst_vs_fp64"], nums["grad_worst_name"] = worst
    nums["grad_rel_l2_vs_fp64__hip_oracle"] = grads
    nums["outside_bound"] = [b[0] for b in bad]
    parity_report(tag, **nums)
    return nums, bad


def test_segmentation_step_at_the_benchmark_configuration_three_way():
    """configs[3]: 16 clouds x 4096 points x (xyz + rgb), 13 classes: HIP vs fp32 oracle vs float64 truth."""
    r = np.random.RandomState(3)
    n = 16 * 4096
    coord = (r.rand(n, 3) * 2 - 1).astype(np.float32)
    rgb = r.rand(n, 3).astype(np.float32)
    offset = (np.arange(1, 17) * 4096).astype(np.int32)
    label = r.randint(0, 13, n).astype(np.int64)
    np.random.seed(17)
    flips = np.where(np.random.rand(16) < 0.5, 1.0, -1.0).astype(np.float32)
    nums, bad = _seg_three_way("seg_16x4096_three_way", coord, rgb, offset, label, 17, flips)
    assert not bad, bad
    assert nums["loss_abs_vs_fp64"] <= 2e-5
    # absolute ceilings next to the ratio asserts (a regression that doubled BOTH errors would pass the ratios): measured 3.9e-5
    # on the logits (scale 4.1), median gradient relative L2 0.0062
    assert nums["logits"]["hip_vs_fp64"] <= 6e-5, nums["logits"]
    assert nums["grad_rel_l2_median_vs_fp64__hip_oracle"][0] <= 1e-2, nums["grad_rel_l2_median_vs_fp64__hip_oracle"]
    # the REFERENCE's own code at this size, fp32 and float64 (tests/golden/seg_cfg3.npz, every 16th row): its fp32 run misses a
    # literal 1e-5 too, the float64 run equals the oracle's truth leg, and the HIP logits are no further from the reference's
    # float64 logits than 1.5 x the reference's own fp32 run
    path = os.path.join(GOLDEN, "seg_cfg3.npz")
    if os.path.exists(path):
        fx = np.load(path)
        got = _seg_three_way.last_logits[fx["rows"]]
        e_hip = np.abs(got - fx["logits64"]).max()
        e_ref = np.abs(fx["logits32"].astype(np.float64) - fx["logits64"]).max()
        parity_report("seg_16x4096_vs_reference_own_runs", hip_vs_reference_fp64=float(e_hip), reference_fp32_vs_its_fp64=float(e_ref),
                      reference_fp32_vs_its_fp64_all_rows=float(fx["logits32_vs_64_max_abs_all_rows"]),
                      scale=float(np.abs(fx["logits64"]).max()))
        assert e_ref > 1e-5, "the reference's own fp32 run meets 1e-5 at configs[3]: restate the bound"
        assert e_hip <= 1.5 * e_ref, (e_hip, e_ref)


def test_benchmark_configurations_under_the_fp32_mfma_instances_too():
    """The two BASELINE-config parity tests above ran under this process's GEMM arithmetic (default: fp32 products formed as six bf16
    MFMAs over three-part operands, RS_GEMM_SPLIT3=1).  Here the same two tests run in a child process under RS_GEMM_SPLIT3=0 (the
    fp32 MFMA instances; the switch is read once per process), appending to the same parity report: every line carries
    `gemm_products`, so the two arithmetics sit side by side (committed copy: profiles/r05/parity_report.jsonl)."""
    import subprocess
    import sys
    from tests.util import ROOT
    if os.environ.get("RS_GEMM_SPLIT3", "1") == "0":
        pytest.skip("this process already runs the fp32 MFMA instances")
    env = dict(os.environ, RS_GEMM_SPLIT3="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_parity_full_gpu.py", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                        "classifier_step_at_the_benchmark_configuration or segmentation_step_at_the_benchmark_configuration"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "2 passed" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])


def test_segmentation_fixture_three_way():
    """The 2-cloud fixture of the reference's own run (tests/golden/seg_model.npz): additionally the REFERENCE's fp32 logits
    against the float64 truth -- the distance the reference itself keeps from the exact network."""
    fx = np.load(os.path.join(GOLDEN, "seg_model.npz"))
    label = fx["label"].astype(np.int64)
    nums, bad = _seg_three_way("seg_fixture_three_way", fx["coord"], fx["rgb"], fx["offset"], label, 9, fx["inv_sign"])
    truth = seg_ref.step(seg_state(), fx["coord"], fx["rgb"], fx["offset"], label, fx["inv_sign"], dtype=torch.float64,
                         want_grads=False)
    ref_err = np.abs(fx["logits"] - truth["logits"].detach().numpy()).max()
    # ... and the truth itself cross-checked: the oracle in float64 against the REFERENCE's own code in float64
    t64 = np.load(os.path.join(GOLDEN, "seg_model_fp64.npz"))
    truth_gap = np.abs(truth["logits"].detach().numpy() - t64["logits64"]).max()
    parity_report("seg_fixture_reference_fp32_vs_fp64", logits_max_abs=ref_err, oracle_fp64_vs_reference_fp64=truth_gap,
                  reference_fp32_vs_its_own_fp64=np.abs(fx["logits"] - t64["logits64"]).max())
    assert truth_gap <= 2e-6, truth_gap        # (6e-7: the fan features enter both as fp32, last-ulp atan2 / acos differences)
    assert not bad, bad
    assert nums["logits"]["hip_vs_fp64"] <= 1.5 * max(ref_err, nums["logits"]["fp32_oracle_vs_fp64"])
    assert nums["logits"]["hip_vs_fp64"] <= 6e-5 and nums["grad_rel_l2_median_vs_fp64__hip_oracle"][0] <= 1e-2      # absolute ceilings (measured 2.6e-5)


def test_bf16_classifier_step_at_configs4_shape():
    """configs[4]: B=64 x 2048 points, bf16 MFMA operands AND bf16 storage of the conv outputs: finite, same geometry, and as
    close to the fp32 path as the reference's own layers are under torch.autocast(bfloat16) (the yardstick of DESIGN §5a,
    here at model level: the torch executor of the SA stacks under autocast, everything else as in the fp32 run).  A bf16
    forward moves the max-pool argmax of near-tied rows, which re-routes whole gradient rows: the gradient cosine against
    fp32 is ~0.9 for either implementation, so it is asserted relative to autocast's."""
    from repsurf_amd import mlp
    from tests import torch_executor
    from util.utils import SmoothClsLoss
    xyz, label = cloud(31, 64, 2048), (np.arange(64) % 15).astype(np.int64)
    res = {}

    def run(tag):
        model = build_cls()
        model.surface_constructor.register_forward_hook(lambda m, i, o: res.__setitem__(tag + "_normal", o.detach().clone()))
        torch.manual_seed(5)
        pred = model(dev(xyz).permute(0, 2, 1).contiguous())
        loss = SmoothClsLoss()(pred, dev(label))
        loss.backward()
        res[tag] = (pred.detach().cpu().numpy(), float(loss.detach()),
                    np.concatenate([p.grad.detach().cpu().numpy().ravel().astype(np.float64)
                                    for n, p in model.named_parameters() if not is_pre_bn_bias(n)]))
    for prec in ("fp32", "bf16"):
        mlp.set_precision(prec)
        try:
            run(prec)
        finally:
            mlp.set_precision("fp32")
    torch_executor.set_backend("torch")
    try:
        def under_autocast(fn):
            def wrapped(*a, **k):
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    return fn(*a, **k).float()
            return wrapped
        mlp.sa_mlp_cd, mlp.sa_mlp_plain = under_autocast(torch_executor.sa_mlp_cd), under_autocast(torch_executor.sa_mlp_plain)
        run("autocast")
    finally:
        torch_executor.set_backend("hip")
    assert np.isfinite(res["bf16"][0]).all() and np.isfinite(res["bf16"][2]).all()
    assert torch.equal(res["fp32_normal"], res["bf16_normal"])        # geometry + constructor stay fp32 in both modes

    def against_fp32(tag):
        g, f = res[tag][2], res["fp32"][2]
        return float(g @ f / (np.linalg.norm(g) * np.linalg.norm(f))), float(np.abs(res[tag][0] - res["fp32"][0]).max())
    (cos, err), (cos_a, err_a) = against_fp32("bf16"), against_fp32("autocast")
    parity_report("cls_b64x2048_bf16_vs_fp32", grad_cosine=cos, logp_max_abs=err, loss_abs=abs(res["bf16"][1] - res["fp32"][1]),
                  autocast_grad_cosine=cos_a, autocast_logp_max_abs=err_a)
    assert cos > 0.85 and cos >= cos_a - 0.02, (cos, cos_a)
    assert err < 0.15 and err <= 1.25 * err_a + 1e-2, (err, err_a)
    assert abs(res["bf16"][1] - res["fp32"][1]) < 5e-2


def test_pre_model_sample_equals_oracle_fps_and_gather():
    """modules.pointnet2_utils.sample (classification/modules/pointnet2_utils.py:114-124; the train loop's 2048 -> 1024
    down-sampling, train_cls_scanobjectnn.py:218): the same rows as the oracle FPS from the same CPU-generator start."""
    from modules import pointnet2_utils as P
    b, n, m = 6, 2048, 1024
    feat = np.concatenate([cloud(41, b, n), np.random.RandomState(1).rand(b, n, 3).astype(np.float32)], -1)   # xyz + 3
    torch.manual_seed(123)
    state = torch.get_rng_state()
    start = torch.randint(0, n, (b,), dtype=torch.long).numpy().astype(np.int32)
    torch.set_rng_state(state)
    got = P.sample(m, dev(feat).permute(0, 2, 1).contiguous()).permute(0, 2, 1).cpu().numpy()
    idx = G.fps(feat[:, :, :3], m, start)
    assert np.array_equal(got, take(feat, idx))


def test_sectorized_fps_matches_the_reference_function():
    """pointops.sectorized_fps against the fixture produced by the reference's own function
    (segmentation/modules/pointops/functions/pointops.py:52-108 over its own FPS kernel, make_golden_seg.py)."""
    fx = np.load(os.path.join(GOLDEN, "seg_sector.npz"))
    with subproject("segmentation"):
        from modules.pointops.functions import pointops
        for ns in (1, 2, 4):
            got = pointops.sectorized_fps(dev(fx["coord"]), dev(fx["offset"]), dev(fx["new_offset"]), ns,
                                          int(fx["min_points"])).cpu().numpy()
            ref = fx[f"idx_s{ns}"]
            same = (got == ref).mean()
            parity_report(f"sectorized_fps_s{ns}", rows_equal_fraction=same)
            assert np.array_equal(got, ref), (ns, same)


def test_classification_pointops_wrappers_match_the_reference_operators():
    """modules.pointops.functions.pointops (8 wrappers) against the reference's own autograd Functions run over its own
    kernels (tests/golden/make_golden_pointops.py -> cls_pointops.npz).  furthestsampling / gathering / grouping /
    nearestneighbor / interpolation: exact (the operators are copies or 3-term sums in a fixed order); ballquery / knnquery
    follow the CPU-path distance formula (DESIGN §1 `cuda=` note): identical rows except where the two formulas round
    across the radius / swap two neighbours."""
    from modules.pointops.functions import pointops as P
    fx = np.load(os.path.join(GOLDEN, "cls_pointops.npz"))
    xyz, feats, new_xyz = dev(fx["xyz"]), dev(fx["feats"]), dev(fx["new_xyz"])
    fps = P.furthestsampling(xyz, fx["fps"].shape[1])
    assert np.array_equal(fps.cpu().numpy(), fx["fps"])
    f1 = feats.clone().requires_grad_()
    gath = P.gathering(f1, fps)
    assert np.array_equal(gath.detach().cpu().numpy(), fx["gathering"])
    (gath * dev(fx["gathering_w"])).sum().backward()
    assert np.abs(f1.grad.cpu().numpy() - fx["gathering_grad"]).max() <= 1e-6
    rows = 0
    for r, ns in ((0.2, 16), (0.4, 32)):
        got = P.ballquery(r, ns, xyz, new_xyz).cpu().numpy()
        rows += int((got != fx[f"ball_{ns}"]).any(-1).sum())
    knn = P.knnquery(9, xyz, new_xyz).cpu().numpy()
    knn_rows = int((np.sort(knn, -1) != np.sort(fx["knn9"], -1)).any(-1).sum())
    assert np.array_equal(np.sort(fx["knn9"], -1), np.sort(fx["knn9_heap"], -1))
    assert np.array_equal(P.knnquery_heap(9, xyz, new_xyz).cpu().numpy(), knn)
    parity_report("cls_pointops_wrappers", ballquery_rows_differing=rows, knn_rows_with_other_set=knn_rows)
    assert rows == 0 and knn_rows == 0
    idx = dev(fx["ball_16"])
    f2 = feats.clone().requires_grad_()
    grp = P.grouping(f2, idx)
    assert np.array_equal(grp.detach().cpu().numpy(), fx["grouping"])
    (grp * dev(fx["grouping_w"])).sum().backward()
    assert np.abs(f2.grad.cpu().numpy() - fx["grouping_grad"]).max() <= 2e-5        # atomics on both sides
    dist, nidx = P.nearestneighbor(xyz, new_xyz)
    assert np.array_equal(nidx.cpu().numpy(), fx["nn_idx"])
    assert (np.abs(dist.cpu().numpy() - fx["nn_dist"]) <= np.spacing(fx["nn_dist"])).all()   # torch.sqrt (CPU, VML) vs device
    kf = dev(fx["interp_feats"]).requires_grad_()
    itp = P.interpolation(kf, nidx, dev(fx["nn_weight"]))
    assert np.abs(itp.detach().cpu().numpy() - fx["interp"]).max() <= 1e-6
    (itp * dev(fx["interp_w"])).sum().backward()
    assert np.abs(kf.grad.cpu().numpy() - fx["interp_grad"]).max() <= 2e-5


def test_eval_forward_after_a_training_step_uses_the_updated_weights():
    """the prepacked weight copies were validated by (address, tensor version) only; repsurf_amd.optim.Adam
    updates through raw pointers, so an eval forward after a training step found first-layer copies one step old."""
    from repsurf_amd import mlp_hip, optim
    from util.utils import SmoothClsLoss
    model = build_cls()
    opt = optim.Adam(model.parameters(), lr=1e-2)
    x = dev(cloud(3, 4, 1024)).permute(0, 2, 1).contiguous()
    torch.manual_seed(0)
    loss = SmoothClsLoss()(model(x), dev(np.array([1, 2, 3, 4])))
    loss.backward()
    opt.step()
    model.eval()
    with torch.no_grad():
        torch.manual_seed(1)
        a = model(x).clone()
        mlp_hip.weights_changed()                 # a table that is certainly empty
        torch.manual_seed(1)
        b = model(x).clone()
    assert torch.equal(a, b)
