#!/bin/bash
cd /root/repo
O=gpurun_out/r02_i; mkdir -p $O
timeout 600 python bench.py > $O/bench_cls.json 2>$O/bench_cls.err; python -c "import json;d=json.load(open('$O/bench_cls.json'));print('cls',d['ms_per_step'],d['value'],d['roofline']['frac'],d['roofline']['dims'])"
timeout 600 python bench.py --workload seg --steps 20 > $O/bench_seg.json 2>$O/bench_seg.err; python -c "import json;d=json.load(open('$O/bench_seg.json'));print('seg',d['ms_per_step'],d['value'],d['roofline']['frac'],d['roofline']['dims'])"
timeout 300 python bench.py --no-cpu-baseline --no-pipeline --steps 50 > $O/bench_cls_nopipe.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_cls_nopipe.json'));print('nopipe',d['ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --dtype bf16 --steps 50 > $O/bench_cls_bf16.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_cls_bf16.json'));print('bf16',d['ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --dtype bf16 --batch 64 --points 2048 --steps 30 > $O/bench_cls_bf16_b64.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_cls_bf16_b64.json'));print('bf16 b64x2048',d['ms_per_step'],d['value'])"
timeout 300 python bench.py --no-cpu-baseline --batch 64 --points 2048 --steps 30 > $O/bench_cls_fp32_b64.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_cls_fp32_b64.json'));print('fp32 b64x2048',d['ms_per_step'],d['value'])"
timeout 300 python bench.py --no-cpu-baseline --workload seg --dtype bf16 --steps 20 > $O/bench_seg_bf16.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_seg_bf16.json'));print('seg bf16',d['ms_per_step'])"
bash tools/gpu_profile.sh r02f cls > $O/profile_cls.log 2>&1; tail -4 $O/profile_cls.log
bash tools/gpu_profile.sh r02f seg > $O/profile_seg.log 2>&1; tail -3 $O/profile_seg.log
