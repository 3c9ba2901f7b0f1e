#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04av; mkdir -p $O
for i in 1 2 3; do timeout 900 python bench.py --workload seg --no-cpu-baseline > $O/seg_$i.json 2> $O/seg_$i.err; tail -1 $O/seg_$i.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('file', d['value'], d['ms_per_step'])"; done
for i in 1 2; do timeout 900 python bench.py --workload seg --no-cpu-baseline 2> $O/segp_$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipe', d['value'], d['ms_per_step'])"; done
