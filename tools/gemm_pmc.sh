#!/bin/bash
# SQ counters of the row-GEMM kernel at one shape: where do the wave cycles go?
# usage (GPU box): tools/gemm_pmc.sh <rows> <k> <n> [frac]
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/gemm_pmc
rm -rf $D
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $D -o a -- python $R/tools/gemm_bench.py one "$@" > $D.a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $D -o b -- python $R/tools/gemm_bench.py one "$@" > $D.b.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM --output-format csv -d $D -o c -- python $R/tools/gemm_bench.py one "$@" > $D.c.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/gemm_pmc/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'gemm_rows_kernel' not in r['Kernel_Name']:
            continue
        k = r['Kernel_Name'].split('(')[0][-28:]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in agg.items():
    print(k)
    wc = sum(c.get('SQ_WAVE_CYCLES', [0])) / max(1, len(c.get('SQ_WAVE_CYCLES', [1])))
    for n, v in sorted(c.items()):
        m = sum(v) / len(v)
        print('   %-28s %14.0f  (%5.1f%% of WAVE_CYCLES)  n=%d' % (n, m, 100 * m / wc if wc else 0, len(v)))
PY
tail -3 $D.a.log | cut -c1-250
