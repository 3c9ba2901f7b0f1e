for cfg in "REPSURF_PIPE_AT=start" "REPSURF_PIPE_AT=backward" "REPSURF_PIPE_AT=start" "REPSURF_PIPE_AT=backward"; do
  env $cfg python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])"
done
