#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04aw; mkdir -p $O
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
timeout 900 python bench.py --workload seg > $O/bench_seg.json 2> $O/bench_seg.err; echo "seg rc=$?"; tail -1 $O/bench_seg.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
timeout 900 python bench.py --workload seg --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('again', d['value'], d['ms_per_step'])"
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
