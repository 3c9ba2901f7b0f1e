#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 --timeout=600 -p no:cacheprovider > gpurun_out/pytest_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_all.log
timeout 600 python bench.py --steps 30 --warmup 3 --breakdown gpurun_out/breakdown_g.json > gpurun_out/bench_graph.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_graph.log
REPSURF_COMPACT=0 timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_dense.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_all.log | tail -n 12; tail -n 2 gpurun_out/bench_graph.log | cut -c1-200; tail -n 1 gpurun_out/bench_dense.log | cut -c1-200
