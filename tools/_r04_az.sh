#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04az; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu --timeout 900 > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/gpu_tests.log | tail -2; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_mfma_launches']['frac'])"
