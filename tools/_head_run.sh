python -m pytest tests/test_mlp_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/bench_a1.json 2>gpurun_out/bench_a1.err
python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/bench_a2.json 2>/dev/null
python bench.py --no-cpu-baseline --no-graph --steps 10 --warmup 3 --breakdown gpurun_out/breakdown_i.json > /dev/null 2>&1
python - <<'P'
import json
for f in ["bench_a1","bench_a2"]:
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
b=json.load(open("gpurun_out/breakdown_i.json"))
for r in b["kernels"]:
    if r["kernel"] in ("rs_mlp_gemm_rows",): print(r["kernel"], r["dims"], r["launches"], round(r["avg_us"],1))
P
