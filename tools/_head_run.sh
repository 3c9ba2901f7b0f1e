python -m pytest tests/test_mlp_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -4
for cfg in "A=1" "A=2"; do
  env $cfg python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])"
done
