python -m pytest tests/test_mlp_gpu.py -x -q -m gpu 2>&1 | tail -2
for cfg in "REPSURF_UMB_BLOCKS_BWD=512" "REPSURF_UMB_BLOCKS_BWD=256" "REPSURF_UMB_BLOCKS_BWD=192" "REPSURF_UMB_BLOCKS_BWD=128" "REPSURF_UMB_BLOCKS_BWD=512" "REPSURF_UMB_BLOCKS_BWD=256"; do
  env $cfg python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])"
done
