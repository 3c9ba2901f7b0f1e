#!/bin/bash
# Boxes of the pool differ by a few percent (power / clocks).  Re-run the bench + profile parts of the evidence only on a box whose quick
# classification step reaches the round's earlier boxes' time; print the quick number either way.
cd $GRAFT_REPO_ROOT
Q=$(python bench.py --no-extra-legs --no-alt-arithmetic --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "quick cls ms_per_step on this box: $Q"
if python -c "import sys; sys.exit(0 if float('$Q') < ${1:-1.24} else 1)"; then
  bash tools/r06_evidence.sh bench profile 2>&1 | grep -v "^[WEI]2026" | tail -30
else
  echo "slow box: evidence not re-run"
fi
