#!/usr/bin/env python3
"""Two-stream hazard, synthetic victims (tools/victim/victim.hip): which KIND of kernel, launched on a side stream beside the replaying network
graph of the ragged segmentation step, ever writes other values than it does alone?  usage: victim_probe.py [launches] [iters]  (GPU box;
build first: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -shared -fPIC
-o tools/victim/libvictim.so tools/victim/victim.hip)"""
import os, sys, time, ctypes, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "repsurf_amd", "segmentation"))
import numpy as np, torch
from repsurf_amd import ops, _lib
from repsurf_amd.graph import RaggedSegStep
from repsurf_amd.head import CrossEntropyLoss
from repsurf_amd.optim import Adam
from models.repsurf.repsurf_umb_ssg import Model
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "victim", "libvictim.so"))
lib.victim_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
FAN = None
if os.environ.get("FAN_LIB"):          # an experimental build of the fan-feature kernel (same entry point) instead of the product library's
    FAN = ctypes.CDLL(os.path.join(ROOT, "tools", "victim", os.environ["FAN_LIB"]))
    FAN.rs_umbrella_fan_offset.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 7
dev = torch.device("cuda")
torch.manual_seed(0)
model = Model(argparse.Namespace(return_polar=False, in_channel=6, group_size=8, num_class=13)).to(dev).train()
crit = CrossEntropyLoss(ignore_index=255)
opt = Adam(model.parameters(), lr=1e-3)
r = np.random.RandomState(1)
clouds, pts = 16, 4096
sizes = r.randint(pts // 2, pts + 1, clouds); nn = int(sizes.sum())
batch = [torch.from_numpy((r.rand(nn, 3) * 2 - 1).astype(np.float32)).to(dev), torch.from_numpy(r.rand(nn, 3).astype(np.float32)).to(dev),
         ops.offsets_tensor(np.cumsum(sizes).tolist(), dev)]
label = torch.from_numpy(r.randint(0, 13, nn).astype(np.int64)).to(dev)
rs = RaggedSegStep(model, crit, opt, batch, label, capacity=clouds * pts, max_cloud_rows=pts, overlap=True)
AGG = os.environ.get("AGG", "graph")      # what runs on the main stream: the replayed network graph | fwd | fwdbwd (an eager GEMM stack) | matmul
if AGG in ("fwd", "fwdbwd"):
    from repsurf_amd import mlp as _mlp
    a_convs = torch.nn.ModuleList([torch.nn.Conv2d(64, 128, 1), torch.nn.Conv2d(128, 256, 1), torch.nn.Conv2d(256, 512, 1)]).to(dev)
    a_bns = torch.nn.ModuleList([torch.nn.BatchNorm2d(128), torch.nn.BatchNorm2d(256), torch.nn.BatchNorm2d(512)]).to(dev).train()
    a_x = torch.randn(2048 * 32, 64, device=dev, requires_grad=True)
if AGG.startswith("custom:"):
    lib.aggressor_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    c_kind = int(AGG.split(":")[1]); c_sink = torch.zeros(1024, device=dev); c_src = torch.zeros(64 * 4096 + 65536, device=dev)
if AGG == "matmul":
    m_a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16); m_b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)


def aggressor():
    if AGG == "graph":
        rs.g_net[0].replay()
    elif AGG.startswith("custom:"):
        for _ in range(3):
            lib.aggressor_launch(c_kind, c_sink.data_ptr(), c_src.data_ptr(), 1024, {1: 400, 8: 400, 10: 400}.get(c_kind, 150), rs.main.cuda_stream)
    elif AGG == "matmul":
        for _ in range(4):
            torch.matmul(m_a, m_b)
    else:
        for _ in range(3):
            out_ = _mlp.sa_mlp_plain(a_x, a_convs, a_bns, 32)
            if AGG == "fwdbwd":
                out_.sum().backward()
                a_x.grad = None


L = int(sys.argv[1]) if len(sys.argv) > 1 else 400
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
per = int(sys.argv[4]) if len(sys.argv) > 4 else 8
c0, off0 = batch[0][:4096].contiguous(), ops.offsets_tensor([4096], dev)
i0, _ = ops.knnquery_offset(9, c0, c0, off0, off0)
fan_ref = ops.umbrella_fan_offset(c0, c0, i0, off0, None, True).clone()
table = torch.from_numpy(r.randint(0, n // 3 - 1, n).astype(np.int32)).to(dev)
src = torch.rand(n, device=dev)
names = {17: "asm v_pk_add op_sel:[0,1]", 18: "asm v_pk_add neg only", 19: "asm v_pk_mul op_sel:[0,1]", 20: "asm v_pk_fma op_sel:[0,1,0]", 21: "asm v_pk_add op_sel:[1,0]", 12: "asm v_pk_fma, SGPR op_sel_hi", 13: "asm v_pk_mul, SGPR", 14: "asm v_pk_add op_sel:[0,1] neg", 15: "asm v_pk_mul, constant", 16: "asm, the four in turn", 7: "packed fp32, VGPR operands", 10: "packed fp32, uniform operand", 11: "packed fp32, constants / neg", 8: "FPS (32 x 1024 -> 512)", 9: "the fan-feature kernel", 5: "sort, lane-mask selects", 6: "sort, VGPR-mask selects", 0: "fma chain", 1: "v_rcp / v_sqrt chain", 2: "IEEE division + sqrt", 3: "atan2f / acosf", 4: "gather loads"}
for kind in ([int(k) for k in sys.argv[5].split(',')] if len(sys.argv) > 5 else (9, 0, 1, 2, 3, 4, 5, 6)):
    ref = torch.empty(n, device=dev)
    if kind == 8:
        fps_xyz = torch.rand(32, 1024, 3, device=dev)
        ref = ops.furthestsampling(fps_xyz, 512).flatten().float()
    if kind == 8:
        pass
    elif kind == 9:
        ref = fan_ref.flatten()
        if FAN is not None:
            ref = torch.empty_like(ref)
            FAN.rs_umbrella_fan_offset(4096, 9, 1, 1, c0.data_ptr(), c0.data_ptr(), i0.data_ptr(), off0.data_ptr(), None, ref.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
    else:
        lib.victim_launch(kind, ref.data_ptr(), table.data_ptr(), src.data_ptr(), n, iters, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for beside in (False, True):
        out = torch.empty((L, ref.numel()), device=dev)
        torch.cuda.synchronize()
        done = 0
        t_start = time.perf_counter()
        while done < L:
            if beside:
                with torch.cuda.stream(rs.main):
                    aggressor()
            with torch.cuda.stream(rs.side):
                for _ in range(min(per, L - done)):             # `per` victim launches under one network replay
                    if kind == 8:
                        fidx = out[done].view(torch.int32).view(32, 512)          # (int32 picks written into the float row: compared as bits below)
                        _lib.call("rs_furthestsampling", 32, 1024, 512, fps_xyz.data_ptr(), None, None, fidx.data_ptr(), rs.side.cuda_stream)
                    elif kind == 9:
                        if FAN is not None:
                            FAN.rs_umbrella_fan_offset(4096, 9, 1, 1, c0.data_ptr(), c0.data_ptr(), i0.data_ptr(), off0.data_ptr(), None,
                                                       out[done].data_ptr(), rs.side.cuda_stream)
                        else:
                            _lib.call("rs_umbrella_fan_offset", 4096, 9, 1, 1, c0.data_ptr(), c0.data_ptr(), i0.data_ptr(), off0.data_ptr(), None,
                                      out[done].data_ptr(), rs.side.cuda_stream)
                    else:
                        lib.victim_launch(kind, out[done].data_ptr(), table.data_ptr(), src.data_ptr(), n, iters, rs.side.cuda_stream)
                    done += 1
            if done % (per * 25) == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        if kind == 8:
            bad = (out.view(torch.int32) != ref.to(torch.int32)).any(1)
        else:
            bad = (out != ref).any(1)
        nbad = int(bad.sum())
        wall_ms = (time.perf_counter() - t_start) * 1e3
        msg = f"[{wall_ms:7.0f} ms] {names[kind]:24s} {('beside ' + AGG) if beside else 'alone on the side stream'}: {nbad} of {L} launches wrote other values"
        if nbad:
            k = int(torch.nonzero(bad)[0])
            lanes = torch.nonzero((out[k].view(torch.int32) != ref.to(torch.int32)) if kind == 8 else (out[k] != ref)).flatten()
            msg += f"; first: {lanes.numel()} values, threads {lanes[:3].tolist()}..{lanes[-1:].tolist()}"
            if kind == 9:
                rows = sorted({int(v) // 90 for k_ in torch.nonzero(bad).flatten()[:40].tolist() for v in torch.nonzero(out[k_] != ref).flatten().tolist()})
                msg += f"; rows mod 64 of the first 40 deviating launches: {sorted({r_ % 64 for r_ in rows})}"
                cols = {}
                for k_ in torch.nonzero(bad).flatten()[:40].tolist():
                    for v in torch.nonzero(out[k_] != ref).flatten().tolist():
                        cols[(v % 90) // 9] = cols.get((v % 90) // 9, 0) + 1
                msg += f"; deviating values per 9-column group: {dict(sorted(cols.items()))}"
        print(msg, flush=True)
