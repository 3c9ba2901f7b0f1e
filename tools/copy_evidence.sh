#!/bin/bash
# copy the merged outputs of tools/r05_evidence.sh (gpurun_out/r0Nev, gpurun_out/prof_r0N_*) into profiles/<round>/ under the names DESIGN.md cites
R=${1:-r03}; P=profiles/$R; H=gpurun_out/${R}h; [ -d gpurun_out/${R}ev ] && H=gpurun_out/${R}ev
mkdir -p $P
for f in bench_cls bench_cls_real bench_cls_dense bench_cls_2x bench_cls_nopipe bench_cls_bf16_b64 bench_seg bench_seg_ragged bench_seg_ragged_s3dis; do [ -f $H/$f.json ] && tail -1 $H/$f.json > $P/$f.json; done
[ -f $H/parity_report.jsonl ] && cp $H/parity_report.jsonl $P/parity_report.jsonl
cp gpurun_out/prof_${R}_cls/graph_kernel_stats.csv $P/cls_graph_kernel_stats.csv
cp gpurun_out/prof_${R}_cls/graph_kernel_stats_by_grid.csv $P/cls_graph_kernel_stats_by_grid.csv
cp gpurun_out/prof_${R}_cls/eager_kernel_stats.csv $P/cls_eager_kernel_stats.csv
cp gpurun_out/prof_${R}_cls/pmc_summary.csv $P/cls_pmc_summary.csv
cp gpurun_out/prof_${R}_cls/traffic.json $P/traffic.json
cp gpurun_out/prof_${R}_seg/graph_kernel_stats.csv $P/seg_graph_kernel_stats.csv
cp gpurun_out/prof_${R}_seg/graph_kernel_stats_by_grid.csv $P/seg_graph_kernel_stats_by_grid.csv
cp gpurun_out/prof_${R}_seg/traffic.json $P/traffic_seg.json
[ -f $H/ballquery_phases.txt ] && cp $H/ballquery_phases.txt $P/ballquery_cells_phase_costs_final.txt
for f in umb_bench.txt grid_meet.txt sharded_time.txt bench_spawn_dry_run.json gpu_tests.log gpu_tests_x5.log gpu_tests_x2.log sharded_soak.txt smoke.log knn_grid_bench.txt umbrella_grid_bench.txt; do [ -f $H/$f ] && cp $H/$f $P/$f; done
python3 - <<PY
import csv,json
rows=list(csv.DictReader(open('$P/cls_graph_kernel_stats_by_grid.csv')))
steps=max(int(r['calls']) for r in rows if 'head_out_fwd' in r['kernel'])
t=sum(float(r['total_ms']) for r in rows if ('gemm_' in r['kernel'] or 'wgrad' in r['kernel']))
tot=sum(float(r['total_ms']) for r in rows)
print("in-graph: steps",steps,"GEMM+wgrad us/step %.1f = %.2f TF = %.4f of 157.3; all kernels %.1f us/step; launches/step %.1f"%(t/steps*1e3, 42.58e9/(t/steps*1e-3)/1e12, 42.58e9/(t/steps*1e-3)/1e12/157.3, tot/steps*1e3, sum(int(r['calls']) for r in rows)/steps))
for r in rows:
    if 'gemm_rows_kernel<64, 64, 4, 4' in r['kernel'] and r['workgroups_x']=='64': print("dominant class in graph:", r['avg_us'], "us = %.3f"%(4.295e9/(float(r['avg_us'])*1e-6)/1e12/157.3))
for f in ['bench_cls','bench_cls_real','bench_cls_dense','bench_cls_2x','bench_cls_nopipe','bench_cls_bf16_b64','bench_seg']:
    d=json.loads(open('$P/%s.json'%f).read()); ro=d.get('roofline') or {}
    print(f, d['value'], d['ms_per_step'], d['steps_timed'], ro.get('frac'), ro.get('avg_launch_us'), (ro.get('all_mfma_launches') or {}).get('achieved'), (ro.get('all_mfma_launches') or {}).get('frac'), d.get('points_per_s'), (d.get('cpu_baseline') or {}).get('value'), d.get('gpu_over_cpu'))
d=json.loads(open('$P/bench_cls.json').read()); print(d['roofline_ballquery']['clouds_per_launch']['2048'], d['fps_us_per_pick']['sa1_1024_to_512'], d['fps_us_per_pick']['sa1_1024_to_512_x3'])
PY
