#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ak; mkdir -p $O
timeout 900 python -m pytest tests/test_seg_gpu.py -q -m gpu -x --timeout 600 -k "large_and_ragged" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^FAILED|^ERROR|Error|assert " $O/tests.log | head
