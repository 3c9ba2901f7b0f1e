for cfg in "RS_GEMM_BM64=1" "RS_GEMM_SLOTS64=768 REPSURF_PARTIAL_BLOCKS=768" "RS_GEMM_BN64_BELOW64=256" "RS_GEMM_BN64_BELOW64=1024" "RS_GEMM_SLOTS=768 RS_GEMM_SLOTS64=768 REPSURF_PARTIAL_BLOCKS=768"; do
  env $cfg python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])"
done
