#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 --timeout=600 -p no:cacheprovider > gpurun_out/pytest_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_all.log
timeout 600 python bench.py --steps 30 --warmup 3 --breakdown gpurun_out/breakdown_g.json > gpurun_out/bench_graph.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_graph.log
rm -f gpurun_out/fps_sweep.log
for w in 1 2 4 8; do RS_FPS_WAVES=$w python - <<'PY' >> gpurun_out/fps_sweep.log 2>&1
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from repsurf_amd import ops
x = (torch.rand(32, 1024, 3, device="cuda") * 2 - 1)
st = torch.zeros(32, dtype=torch.int32, device="cuda")
for n, m in ((1024, 512), (512, 128)):
    xx = x[:, :n].contiguous()
    for _ in range(3): ops.furthestsampling(xx, m, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.furthestsampling(xx, m, st)
    torch.cuda.synchronize(); print("waves", os.environ["RS_FPS_WAVES"], n, m, "us", round((time.perf_counter() - t0) / 20 * 1e6, 1))
PY
done
grep -E "passed|failed" gpurun_out/pytest_all.log | tail -2; tail -n 2 gpurun_out/bench_graph.log | cut -c1-200; grep waves gpurun_out/fps_sweep.log
