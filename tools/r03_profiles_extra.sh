#!/bin/bash
# final round-3 evidence refresh: full GPU tests (parity report), rocprof kernel stats of the dense / real / 2x / configs[4] lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
export REPSURF_PARITY_REPORT=$GRAFT_REPO_ROOT/$O/parity_report.jsonl
rm -f $REPSURF_PARITY_REPORT
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c 1-200
unset REPSURF_PARITY_REPORT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
prof() { # tag env... -- args
  local tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$tag -o g -- python $R/bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 3 --min-seconds 0.05 "$@" > $R/$O/prof_$tag.log 2>&1
  echo "prof $tag rc=$?"
  python $R/tools/kernel_stats_by_grid.py $R/$O/prof_$tag/g_kernel_trace.csv > $R/$O/${tag}_kernel_stats_by_grid.csv 2>/dev/null
  cp $R/$O/prof_$tag/g_kernel_stats.csv $R/$O/${tag}_kernel_stats.csv 2>/dev/null
  rm -rf $R/$O/prof_$tag
}
REPSURF_COMPACT=0 prof cls_dense
prof cls_real --data real
prof cls_2x --model repsurf_ssg_umb_2x
prof cls_bf16_b64 --dtype bf16 --batch 64 --points 2048
cd $R; ls -la $O | head -20
