"""Host-side bookkeeping of the round-4 geometry / decoder paths: what needs no device (CPU tensors only)."""
import pytest
import torch

from repsurf_amd import mlp_hip, ops
from repsurf_amd.geometry import StageGeometry


def test_largest_cloud_comes_from_the_host_copy_only():
    cpu = torch.device("cpu")
    off = ops.offsets_tensor([3, 10, 12], cpu)
    assert ops._largest_cloud(off) == 7
    assert ops.host_offsets(off) == (3, 10, 12)
    bare = torch.tensor([3, 10, 12], dtype=torch.int32)          # no host copy travels with it: unknown, never a device read
    assert ops._largest_cloud(bare) == 0
    off2 = ops.offsets_tensor([5], cpu)
    assert ops._largest_cloud(off2) == 5


def test_inverse_index_declines_without_known_cloud_sizes():
    """ops.inverse_index builds nothing when the host does not hold the cloud sizes, or a cloud exceeds the 16 384 rows one
    workgroup's counters hold: the backward of the gather then takes the atomic scatter."""
    src = torch.zeros((8, 3), dtype=torch.int32)
    bare = torch.tensor([8], dtype=torch.int32)
    assert ops.inverse_index(src, 3, bare, bare) is None
    big = ops.offsets_tensor([20000], torch.device("cpu"))
    assert ops.inverse_index(src, 3, ops.offsets_tensor([8], torch.device("cpu")), big) is None


def test_compact_index_carries_the_inverse_through_clone():
    cpu = torch.device("cpu")
    plain = ops.CompactIndex.empty(4, 8, cpu)
    assert plain.meta.shape == (3, 32) and plain.pts is None and plain.csr(10) is None
    ci = ops.CompactIndex.empty(4, 8, cpu, points=10)
    assert ci.meta.shape == (4, 32) and ci.pts.numel() == 21 and ci.csr_rows.data_ptr() == ci.meta[3].data_ptr()
    assert ci.csr(10) is None                                     # nothing built yet
    ci.csr_ready = True
    off, centre, rows = ci.csr(10)
    assert off.numel() == 11 and centre.numel() == 10 and rows.numel() == 32
    assert ci.csr(11) is None                                     # another number of source points: not this inverse
    g = StageGeometry(torch.zeros((1, 4), dtype=torch.int32), torch.zeros((1, 4, 3)), torch.zeros((1, 4, 8), dtype=torch.int32),
                      torch.zeros((1, 4), dtype=torch.int32), ci)
    c = g.clone()
    assert c.index.csr_ready and c.index.csr(10) is not None and c.index.pts.data_ptr() != ci.pts.data_ptr()
    assert len(g.tensors()) == len(c.tensors()) == 8              # fps_idx, new_center, idx, cnt, offsets, mult, meta, pts


def test_lazy_rows_is_a_training_time_handover():
    lazy = mlp_hip.LazyRows()
    assert lazy.y is None and lazy.vec is None and lazy.part is None
    bn = torch.nn.BatchNorm1d(4)
    assert mlp_hip.lazy_rows_usable([bn]) == mlp_hip.LAZY_ROWS
    bn.eval()
    assert not mlp_hip.lazy_rows_usable([bn])                     # running statistics: the layer-by-layer route


def test_three_bf16_parts_carry_an_fp32_product():
    """The arithmetic behind the split-product GEMMs (csrc/mlp.hip unit 4; include/repsurf_hip.h: rs_mlp_gemm_split3), restated in
    numpy: x = h + m + l with every part the nearest-even bf16 of what the parts before it left.  Three parts reproduce x to fp32's
    last bit or better, and the six kept products (hh, hm, mh, hl, lh, mm) give a K = 512 dot product that is as close to the
    float64 one as a plain fp32 product; two parts / three products -- the usual bf16x3 -- miss the 1e-5 the north star allows."""
    import importlib.util
    import os
    import numpy as np
    from tests.conftest import ROOT
    spec = importlib.util.spec_from_file_location("bf16_split_accuracy", os.path.join(ROOT, "tools", "probes", "bf16_split_accuracy.py"))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(4096) * np.exp(rng.uniform(-20, 20, 4096))).astype(np.float32)
    h, m, l = probe.split(x, 3)
    for part in (h, m, l):                                   # every part IS a bf16: the low 16 bits of its fp32 pattern are zero
        assert not (part.view(np.uint32) & 0xFFFF).any()
    back = (h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64))
    assert np.all(np.abs(back - x.astype(np.float64)) <= np.abs(x.astype(np.float64)) * 2.0 ** -24)
    a = np.maximum(rng.standard_normal((64, 512)).astype(np.float32), 0)
    w = (rng.standard_normal((512, 48)) / np.sqrt(512)).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64)

    def product(parts, terms):
        pa, pw = probe.split(a, parts), probe.split(w, parts)
        acc = np.zeros(ref.shape, np.float32)
        for i, j in reversed(terms):
            acc = acc + (pa[i] @ pw[j]).astype(np.float32)
        return np.abs(acc - ref).max()

    six = product(3, [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)])
    three = product(2, [(0, 0), (0, 1), (1, 0)])
    plain = np.abs(a @ w - ref).max()
    assert six <= 2 * plain and six < 2e-6, (six, plain)
    assert three > 5e-6, three                               # what the third part buys


def test_ragged_capacity_table_maps_launch_sizes_to_counts():
    """repsurf_amd.ragged (host logic; CPU tensors stand in for the device table): the capacities of one step are pairwise distinct, the
    row count a launch is sized for finds its slot, `fill` writes level counts times the group sizes, counts above a capacity and
    colliding capacities are refused, and nothing is looked up outside a `with` block."""
    import torch
    from repsurf_amd import ragged
    cap = ragged.Capacity([4096, 1024, 256, 64, 16], 32, 9, "cpu")
    assert ragged.dev(4096) is None                                   # no active capacity: every kernel takes its scalar row count
    cap.fill([3000, 700, 170, 40, 9])
    with cap:
        base = cap.table.data_ptr()
        slots = {rows: (ragged.dev(rows) - base) // 4 for rows in (4096, 9 * 4096, 1024, 32 * 1024, 256, 32 * 256, 64, 32 * 64, 16, 32 * 16)}
        assert sorted(slots.values()) == list(range(10))
        assert ragged.dev(4095) is None and ragged.dev(128) is None   # not a capacity of this step
        want = {4096: 3000, 9 * 4096: 27000, 1024: 700, 32 * 1024: 22400, 256: 170, 32 * 256: 5440, 64: 40, 32 * 64: 1280, 16: 9, 32 * 16: 288}
        for rows, slot in slots.items():
            assert int(cap.table[slot]) == want[rows]
        with pytest.raises(RuntimeError):
            with cap:
                pass
    assert ragged.dev(4096) is None
    with pytest.raises(ValueError, match="captured for at most"):
        cap.fill([4097, 1, 1, 1, 1])
    with pytest.raises(ValueError, match="appears twice"):
        ragged.Capacity([288, 9], 32, 9, "cpu")                        # 9 x 32 = 288: a level-0 capacity that is not a multiple of 256


def test_device_key_resolves_an_indexless_cuda_device():
    """the per-device counter tables were keyed by str(device); a query with 'cuda' read a table stored under 'cuda:0' as 0."""
    from repsurf_amd import _lib
    assert _lib.device_key("cuda") == _lib.device_key(torch.device("cuda")) == _lib.device_key("cuda:0")
    assert _lib.device_key("cuda:3") == "cuda:3" and _lib.device_key("cpu") == "cpu"
