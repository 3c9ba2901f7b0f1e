"""Kernels of the geometry stream beside the network graph of the other stream (graph.PipelinedStep / RaggedSegStep run them that way).

Round 6: the fan-feature kernel, built with the compiler's SLP vectorizer, computed lanes 48..63 of a wave from other operands in 1-3 % of
its launches while split-product GEMMs of the other stream shared its compute units (profiles/r06/eager_beside_graph.txt); the geometry
translation units are compiled without the vectorizers since (Makefile).  A build that loses the flags fails here: 8 000 launches beside
the replaying graph, every output equal to the kernel's output alone (the same tree built with the vectorizers: 53 of 8 000 deviate, each in lanes 48..63 of a wave)."""
import os
import re

import numpy as np
import pytest
import torch

from tests.util import subproject

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_geometry_units_build_without_the_vectorizers():
    mk = open(os.path.join(ROOT, "Makefile")).read()
    units = re.search(r"^GEOM_TUS := (.*)$", mk, re.M).group(1).split()
    for tu in ("seg_geom", "knn_umbrella", "grid_knn", "ballquery", "fps", "scene_knn", "knn_wide"):      # everything the side stream launches
        assert tu in units, tu
    assert re.search(r"\$\(foreach t,\$\(GEOM_TUS\).*-fno-slp-vectorize -fno-vectorize", mk)


UNITS = ("seg_geom", "knn_umbrella", "grid_knn", "ballquery", "fps", "scene_knn", "knn_wide", "group", "interp",      # what the side stream launches
         "umbrella_mlp", "umbrella_mfma", "head", "adam", "mlp", "mlp_bf16", "mlp_sb", "mlp_split")                  # the network stream's units
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.mark.parametrize("unit", UNITS)
def test_no_unit_holds_packed_fp32_with_op_sel_on_the_second_source(unit, tmp_path):
    """The instruction the hazard was traced to (profiles/r06/eager_beside_graph.txt: v_pk_add_f32 / v_pk_fma_f32 whose src1 carries op_sel)
    must not appear in anything the side stream launches -- nor, since it costs nothing, anywhere else: checked in the device code of the
    built objects."""
    import shutil
    import subprocess
    obj = os.path.join(ROOT, "build", unit + ".hip.o")
    if not os.path.exists(obj) or not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no built object / no llvm tools here (make builds build/*.hip.o)")
    work = shutil.copy(obj, tmp_path / "unit.o")          # (llvm-objcopy rewrites its input)
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={tmp_path / 'unit.bundle'}", str(work)])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={tmp_path / 'unit.bundle'}", f"--output={tmp_path / 'unit.elf'}"])
    text = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", str(tmp_path / "unit.elf")], text=True)
    assert "s_endpgm" in text, "no device code disassembled"
    bad = [ln.strip() for ln in text.splitlines() if re.search(r"v_pk_(add|fma)_f32\b.*op_sel:\[[01],1", ln)]
    assert not bad, f"{unit}: {len(bad)} packed-fp32 instructions with op_sel on src1, e.g. {bad[0]}"


@pytest.mark.gpu
def test_fan_features_beside_the_replaying_network_graph_equal_the_kernel_alone():
    from repsurf_amd import _lib, ops
    from repsurf_amd.graph import RaggedSegStep
    from repsurf_amd.head import CrossEntropyLoss
    from tests.test_seg_gpu import _seg_model
    cuda = torch.device("cuda")
    r = np.random.RandomState(1)
    clouds, pts = 16, 4096                                   # the network graph of bench.py --workload seg --ragged: ~2.5 ms of GEMM-heavy replay
    sizes = r.randint(pts // 2, pts + 1, clouds)
    n = int(sizes.sum())
    batch = [torch.from_numpy((r.rand(n, 3) * 2 - 1).astype(np.float32)).to(cuda), torch.from_numpy(r.rand(n, 3).astype(np.float32)).to(cuda),
             ops.offsets_tensor(np.cumsum(sizes).tolist(), cuda)]
    label = torch.from_numpy(r.randint(0, 13, n).astype(np.int64)).to(cuda)
    with subproject("segmentation"):
        model = _seg_model()
        step = RaggedSegStep(model, CrossEntropyLoss(ignore_index=255), None, batch, label, capacity=clouds * pts, max_cloud_rows=pts)
        coord, off = batch[0][:4096].contiguous(), ops.offsets_tensor([4096], cuda)
        idx, _ = ops.knnquery_offset(9, coord, coord, off, off)
        alone = ops.umbrella_fan_offset(coord, coord, idx, off, None, True)
        torch.cuda.synchronize()
        ring = torch.empty((200,) + tuple(alone.shape), dtype=alone.dtype, device=cuda)
        deviating, launches = 0, 0
        for _ in range(40):                                      # 40 x (25 back-to-back replays, 8 fan launches under each) = 8 000 launches
            for rep in range(25):
                with torch.cuda.stream(step.main):
                    step.g_net[0].replay()
                with torch.cuda.stream(step.side):
                    for k in range(8):
                        _lib.call("rs_umbrella_fan_offset", 4096, 9, 1, 1, coord.data_ptr(), coord.data_ptr(), idx.data_ptr(), off.data_ptr(), None,
                                  ring[rep * 8 + k].data_ptr(), step.side.cuda_stream)
            torch.cuda.synchronize()
            launches += 200
            bad = (ring != alone).flatten(1).any(1)
            for k in torch.nonzero(bad).flatten().tolist():
                rows = torch.nonzero((ring[k] != alone).flatten(1).any(1)).flatten()
                deviating += 1
                print(f"deviating launch: rows {rows[:3].tolist()}..{rows[-1:].tolist()} (mod 64: {sorted({int(r) % 64 for r in rows})[:4]}..)")
        step.close()
    assert deviating == 0, f"{deviating} of {launches} fan-feature launches beside the network graph differ from the kernel alone"
