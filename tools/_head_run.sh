python -m pytest tests/test_head_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/bench_head_hip.json 2>gpurun_out/bench_head_hip.err
REPSURF_HEAD=torch python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/bench_head_torch.json 2>gpurun_out/bench_head_torch.err
python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/bench_head_hip2.json 2>/dev/null
python bench.py --no-cpu-baseline --no-graph --steps 10 --warmup 3 --breakdown gpurun_out/breakdown_h2.json > /dev/null 2>&1
python - <<'P'
import json
for f in ["bench_head_hip","bench_head_torch","bench_head_hip2"]:
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
b=json.load(open("gpurun_out/breakdown_h2.json"))
for r in b["kernels"]:
    if "head" in r["kernel"] or "smooth" in r["kernel"]: print(r["kernel"], r["dims"], r["launches"], round(r["avg_us"],1))
P
