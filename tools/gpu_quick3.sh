#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 --timeout=600 -p no:cacheprovider > gpurun_out/pytest_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_all.log
timeout 600 python bench.py --steps 30 --warmup 3 --breakdown gpurun_out/breakdown_g.json > gpurun_out/bench_graph.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_graph.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_torchrun1.log 2>&1; echo "torchrun rc=$?" >> gpurun_out/bench_torchrun1.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
grep -E "passed|failed|FAILED" gpurun_out/pytest_all.log | tail -5; tail -n 2 gpurun_out/bench_graph.log | cut -c1-200; tail -n 2 gpurun_out/bench_torchrun1.log | cut -c1-200; tail -2 gpurun_out/smoke.log
