#!/usr/bin/env python3
"""GPU diagnostic: per-tensor errors of the HIP shared-MLP executor vs torch fp32 / fp64, and timings.
Writes gpurun_out/mlp_diag.json (never asserts)."""
import copy, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.test_mlp_gpu import CASES, make_cd, run_cd, rel

out = {"cases": []}
for (groups, ns, pos, feat, widths) in CASES + [(16384, 32, 6, 10, [64, 64, 128]), (4096, 64, 6, 138, [128, 128, 256]), (32, 128, 6, 266, [256, 512, 1024])]:
    rec = {"case": [groups, ns, pos, feat, widths]}
    try:
        mod = make_cd(pos, feat, widths, 1)
        g = torch.Generator().manual_seed(2)
        x = torch.randn(groups * ns, pos + feat, generator=g).cuda()
        w = torch.randn(groups, widths[-1], generator=g).cuda()
        o_t, g_t = run_cd(copy.deepcopy(mod), x, ns, pos, "torch", w)
        o_h, g_h = run_cd(copy.deepcopy(mod), x, ns, pos, "hip", w)
        rec["out_rel"] = rel(o_h, o_t)
        rec["grad_rel"] = {k: rel(g_h[k], g_t[k]) for k in g_t}
        if groups * ns <= 200000:
            o_d, g_d = run_cd(copy.deepcopy(mod).double(), x.double(), ns, pos, "torch", w.double())
            rec["out_rel_vs64"] = {"hip": rel(o_h.double(), o_d), "torch": rel(o_t.double(), o_d)}
            rec["grad_rel_vs64"] = {k: [rel(g_h[k].double(), g_d[k]), rel(g_t[k].double(), g_d[k])] for k in g_d}
        for backend in ("torch", "hip"):
            m2 = copy.deepcopy(mod)
            for _ in range(2):
                run_cd(m2, x, ns, pos, backend, w)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                run_cd(m2, x, ns, pos, backend, w)
            torch.cuda.synchronize(); rec["ms_" + backend] = (time.perf_counter() - t0) / 5 * 1e3
    except Exception as e:  # noqa
        import traceback
        rec["error"] = traceback.format_exc()[-1500:]
    out["cases"].append(rec)
    print(json.dumps(rec)[:600], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "mlp_diag.json"), "w"), indent=1)
