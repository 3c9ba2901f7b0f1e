#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
b() { python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
echo "cls default                $(b)"
echo "cls WGRAD_CHUNKS=256       $(REPSURF_WGRAD_CHUNKS=256 b)"
echo "cls WGRAD_CHUNKS=1024      $(REPSURF_WGRAD_CHUNKS=1024 b)"
echo "cls WGRAD_MIN_ROWS=128     $(REPSURF_WGRAD_MIN_ROWS=128 b)"
echo "cls WGRAD_MIN_ROWS=32      $(REPSURF_WGRAD_MIN_ROWS=32 b)"
echo "cls SLOTS64_FWD=768        $(RS_GEMM_SLOTS64_FWD=768 b)"
echo "cls SLOTS64=768            $(RS_GEMM_SLOTS64=768 b)"
echo "cls default                $(b)"
echo "seg default                $(b --workload seg)"
echo "seg WGRAD_MIN_ROWS=128     $(REPSURF_WGRAD_MIN_ROWS=128 b --workload seg)"
echo "seg SLOTS64_FWD=768        $(RS_GEMM_SLOTS64_FWD=768 b --workload seg)"
