#!/bin/bash
# GPU box, after tools/build_exp_split.sh SP_PIPE: the software-pipelined split-product loop against the product build --
# correctness first (tests/test_mlp_gpu.py on the experiment library), then the kernels alone and both steps, interleaved.
#   gpurun --timeout 400 -- 'bash tools/sp_pipe_ab.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sp_pipe; mkdir -p $O
cd $R
L=$R/build_exp/librepsurf_SP_PIPE.so
REPSURF_HIP_LIB=$L timeout 300 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu -k "not fp32_mfma_instances" 2>&1 | tail -2
REPSURF_HIP_LIB=$L timeout 100 python tools/gemm_split_ab.py 2>&1 | grep "rows=" > $O/pipe.txt
timeout 100 python tools/gemm_split_ab.py 2>&1 | grep "rows=" > $O/base.txt
paste -d'\n' $O/base.txt $O/pipe.txt | cut -c26-230
b() { python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for i in 1 2; do
echo "cls product   $(b)"
echo "cls pipelined $(REPSURF_HIP_LIB=$L b)"
done
echo "seg product   $(b --workload seg)"
echo "seg pipelined $(REPSURF_HIP_LIB=$L b --workload seg)"
