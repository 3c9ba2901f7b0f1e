"""Repository invariants the build contract depends on."""
import os
import re

import pytest
import torch

from tests.util import ROOT, ref_args


def py_files(sub):
    for d, _, fs in os.walk(os.path.join(ROOT, sub)):
        for f in fs:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_product_never_touches_the_oracle():
    for path in list(py_files("repsurf_amd")):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
        assert "/root/reference" not in src, path


def test_no_cpu_fallback():
    from repsurf_amd import ops, _lib
    with pytest.raises(_lib.RepSurfHipError, match="no CPU fallback"):
        ops.ballquery(0.2, 8, torch.zeros(1, 16, 3), torch.zeros(1, 4, 3))


def test_state_dict_matches_reference_names():
    import numpy as np
    from models.repsurf.repsurf_ssg_umb import Model
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_b4.npz"))
    model = Model(ref_args())
    assert sorted(n for n, _ in model.named_parameters()) == list(g["grad_names"])
    assert sum(p.numel() for p in model.parameters()) == 1476791


def test_rng_draw_order_matches_reference_cpu_path():
    """constructor flip, then one FPS start per sampling stage, all from the CPU generator."""
    import numpy as np
    from modules.pointnet2_utils import draw_fps_start
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_b4.npz"))
    torch.manual_seed(int(g["rng_seed"]))
    flip = torch.randint(0, 2, (4, 1, 1)).float() * 2. - 1.
    s1, s2 = draw_fps_start(4, 1024), draw_fps_start(4, 512)
    assert np.array_equal(flip.view(4).numpy(), g["inv_sign"])
    assert np.array_equal(s1.numpy(), g["fps1_start"]) and np.array_equal(s2.numpy(), g["fps2_start"])


def test_seg_state_dict_matches_reference_names():
    """names, shapes and buffers of the segmentation model equal the reference's (fixture written from the
    reference's own Model by tests/golden/make_golden_seg.py)."""
    import numpy as np
    from tests.util import seg_args, subproject
    g = np.load(os.path.join(ROOT, "tests", "golden", "seg_model.npz"))
    with subproject("segmentation"):
        from models.repsurf.repsurf_umb_ssg import Model
        model = Model(seg_args())
    ref = {k[6:]: tuple(g[k]) for k in g.files if k.startswith("shape/")}
    assert {n: tuple(p.shape) for n, p in model.named_parameters()} == ref
    assert sorted(n for n, _ in model.named_buffers()) == list(g["buffers"])


def test_collate_fn_wire_format():
    """util.data_util.collate_fn == segmentation/util/data_util.py:15-23: rows concatenated, running int32 ends."""
    import torch
    from tests.util import subproject
    with subproject("segmentation"):
        from util.data_util import collate_fn
    g = torch.Generator().manual_seed(0)
    batch = [(torch.rand(n, 3, generator=g), torch.rand(n, 3, generator=g), torch.randint(0, 13, (n,), generator=g))
             for n in (5, 1, 7)]
    coord, feat, label, offset = collate_fn(batch)
    assert coord.shape == (13, 3) and feat.shape == (13, 3) and label.shape == (13,)
    assert offset.dtype == torch.int32 and offset.tolist() == [5, 6, 13]
    assert torch.equal(coord[5:6], batch[1][0]) and torch.equal(label[6:], batch[2][2])
    c2, f2, l2, o2 = collate_fn([(b[0], b[1], None) for b in batch])
    assert l2 is None and o2.tolist() == [5, 6, 13]
