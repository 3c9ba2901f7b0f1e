#!/bin/bash
cd /root/repo
O=gpurun_out/r02_f; mkdir -p $O
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -6 $O/tests.log
timeout 600 python bench.py > $O/bench_cls.json 2>$O/bench_cls.err; python -c "import json;d=json.load(open('$O/bench_cls.json'));print(d['ms_per_step'],d['value'],d['roofline'])"
timeout 600 python bench.py --workload seg --steps 20 > $O/bench_seg.json 2>$O/bench_seg.err; python -c "import json;d=json.load(open('$O/bench_seg.json'));print(d['ms_per_step'],d['value'],d['roofline'])"
