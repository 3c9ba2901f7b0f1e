#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ao; mkdir -p $O
for i in 1 2; do timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $O/gpu_tests_$i.log 2>&1; echo "full run $i rc=$?"; grep -E "passed|failed|Aborted" $O/gpu_tests_$i.log | tail -2; grep -E "^FAILED|^ERROR" $O/gpu_tests_$i.log | head -5; done
