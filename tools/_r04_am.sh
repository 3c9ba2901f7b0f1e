#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04am; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/gpu_tests.log | tail -2; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
