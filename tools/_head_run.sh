python -m pytest tests/test_optim_gpu.py tests/test_head_gpu.py -x -q -m gpu 2>&1 | tail -8
python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/bench_a1.json 2>gpurun_out/bench_a1.err
python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/bench_a2.json 2>/dev/null
tail -3 gpurun_out/bench_a1.err
python - <<'P'
import json
for f in ["bench_a1","bench_a2"]:
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
P
