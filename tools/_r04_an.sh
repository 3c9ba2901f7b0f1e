#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04an; mkdir -p $O
for i in 1 2; do timeout 900 python -m pytest tests/test_graph_gpu.py -q -m gpu --timeout 600 > $O/graph_$i.log 2>&1; echo "graph tests run $i rc=$?"; grep -E "passed|failed|Aborted" $O/graph_$i.log | tail -2; done
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 --deselect tests/test_graph_gpu.py > $O/gpu_tests_rest.log 2>&1; echo "rest rc=$?"; grep -E "passed|failed" $O/gpu_tests_rest.log | tail -2; grep -E "^FAILED|^ERROR" $O/gpu_tests_rest.log | head
