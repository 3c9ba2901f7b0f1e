#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ev; mkdir -p $O
timeout 900 python bench.py --workload seg > $O/bench_seg.json 2> $O/bench_seg.err; echo "seg rc=$?"; tail -1 $O/bench_seg.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo "== sharded step, 1-rank RCCL group, collective forced"; REPSURF_FORCE_ALLREDUCE=1 timeout 300 python tools/sharded_time.py pipe pipe_sharded pipe_sharded > $O/sharded_time.txt 2>&1; echo rc=$?; grep "ms/step\|rror" $O/sharded_time.txt | cut -c 1-300
