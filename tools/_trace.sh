R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/prof_h; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o graph -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 --no-kernel-timing > $D/graph.log 2>&1
ls $D
