"""The segmentation oracle (oracle/seg_ref.py + oracle/geom_oracle.c) against fixtures produced by the
reference's own segmentation torch code (tests/golden/make_golden_seg.py).  CPU only.

What these fixtures pin: everything downstream of the packed FPS / kNN kernels (which restate CUDA sources
and stay parity-unpinned) -- umbrella feature order, the fixed rotation, NaN patching, numpy-RNG flips,
sample_and_group channel order, interpolation weights, SA / FP / classifier wiring, gradients."""
import os

import numpy as np
import pytest
import torch

from oracle import geom_oracle as G
from oracle import seg_ref
from tests.util import GOLDEN


@pytest.fixture(scope="module")
def geom():
    return np.load(os.path.join(GOLDEN, "seg_geom.npz"))


@pytest.fixture(scope="module")
def model_fx():
    return np.load(os.path.join(GOLDEN, "seg_model.npz"))


@pytest.mark.parametrize("tag,rotate", [("fix", True), ("none", False)])
def test_umbrella_features_match_reference(geom, tag, rotate):
    coord, offset = geom["coord"], geom["offset"]
    idx, _ = G.knn_offset(9, coord, coord, offset, offset)
    feat, tie = G.umbrella_fan_offset(coord, coord, idx, offset, geom[f"umb_{tag}_sign"], rotate)
    ref = geom[f"umb_{tag}"]
    assert feat.shape == ref.shape == (coord.shape[0], 9, 10)
    assert np.array_equal(np.isnan(feat), np.isnan(ref))
    err = np.nan_to_num(np.abs(feat - ref)).reshape(coord.shape[0], -1).max(-1)
    assert tie.sum() <= 3                       # near-tie azimuths: order decided by the exact predicate
    assert err[~tie].max() <= 5e-7, err[~tie].max()


@pytest.mark.parametrize("polar", [False, True])
def test_sample_and_group_matches_reference(geom, polar):
    t = "p" if polar else "x"
    coord, offset = geom["coord"], geom["offset"]
    normal = torch.from_numpy(geom["sg_normal_in"])
    feat = torch.cat([torch.from_numpy(coord), torch.from_numpy(geom["sg_rgb"])], 1)
    nc, nn_, rows, no, _ = seg_ref.sample_and_group(4, 32, coord, normal, feat, offset, return_polar=polar)
    assert np.array_equal(no, geom[f"sg_{t}_offset"])
    assert np.array_equal(nc, geom[f"sg_{t}_center"])
    assert np.array_equal(nn_.numpy(), geom[f"sg_{t}_normal"])
    ref = geom[f"sg_{t}_feat"]
    got = rows.numpy().reshape(ref.shape)
    assert np.abs(got - ref).max() <= 5e-7


def test_interp_weights_match_reference(geom):
    w = G.interp_weights(geom["interp_dist"] ** 2)      # not bit-identical input: sqrt(d2)^2 != d2; compare loosely
    assert np.abs(w - geom["interp_weight"]).max() <= 1e-6
    nc, no = geom["sg_x_center"], geom["sg_x_offset"]
    _, d2 = G.knn_offset(3, nc, geom["coord"], no, geom["offset"])
    # torch.sqrt on CPU goes through MKL VML (not correctly rounded: 25 of 3087 values differ in the last
    # bit from sqrtf); the oracle and the HIP kernel use the correctly rounded sqrt, like CUDA's sqrtf
    assert np.abs(np.sqrt(d2) - geom["interp_dist"]).max() <= 6e-8
    assert np.abs(G.interp_weights(d2) - geom["interp_weight"]).max() <= 2e-7


def test_strided_offset():
    assert seg_ref.strided_offset([300, 812, 1029], 4).tolist() == [75, 203, 257]
    assert seg_ref.strided_offset([7], 4).tolist() == [1]


def test_model_step_matches_reference(model_fx):
    from tests.util import seg_state
    state = seg_state()
    out = seg_ref.step(state, model_fx["coord"], model_fx["rgb"], model_fx["offset"], model_fx["label"].astype(np.int64),
                       model_fx["inv_sign"])
    assert not out["near_tie"].any()
    assert np.abs(out["normal"].detach().numpy() - model_fx["normal"]).max() <= 2e-5
    for n in ("sa1", "sa2", "sa3", "sa4"):
        assert np.array_equal(out[n + "_fps"].shape[0], model_fx[n + "_center"].shape[0])
        ref = model_fx[n + "_feat_sub"]
        got = out[n + "_feat"].detach().reshape(-1)[::7].numpy()
        assert np.abs(got - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max()), n
    ref = model_fx["fp1_sub"]
    got = out["fp1_feat"].detach().reshape(-1)[::7].numpy()
    assert np.abs(got - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(out["logits"].detach().numpy() - model_fx["logits"]).max() <= 1e-4
    assert abs(out["loss"].item() - float(model_fx["loss"])) <= 2e-5
    bad = []
    for name, g in out["grads"].items():
        ref_n = float(model_fx["gnorm/" + name])
        ref = model_fx["gsub/" + name]
        got = g.reshape(-1)[::(7 if g.numel() > 4096 else 1)].numpy()
        scale = max(ref_n, 1e-6)
        if seg_ref_pre_bn_bias(name):
            continue
        # Tolerance: run in float64, the reference's own code differs from its fp32 self by 4e-3..1e-2
        # (relative L2) on these gradients -- ReLU / max-pool selections flip on 1e-5 forward differences at
        # this small size -- and the oracle sits at the same distance from the fp64 result (probed).
        if np.linalg.norm(got - ref) > 3e-2 * max(np.linalg.norm(ref), 1e-3 * scale):
            bad.append((name, float(np.linalg.norm(got - ref)), float(np.linalg.norm(ref))))
    assert not bad, bad


def seg_ref_pre_bn_bias(name):
    """biases added right before a BatchNorm: analytically zero gradient (the reference holds fp noise)"""
    if not name.endswith(".bias"):
        return False
    return (".mlp_l0." in name or ".mlp_f0." in name or ".mlp_s0." in name or ".mlp_convs." in name
            or name in ("surface_constructor.mlps.0.bias", "surface_constructor.mlps.3.bias", "classifier.0.bias"))
