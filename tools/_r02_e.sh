#!/bin/bash
cd /root/repo
O=gpurun_out/r02_e; mkdir -p $O
timeout 300 python tools/fps_sweep.py > $O/fps_sweep.txt 2>&1; cat $O/fps_sweep.txt
for i in 1 2; do
for v in product NO_EARLY_PREFETCH EARLY_PREFETCH_ALL LDS_EPILOGUE; do
  if [ $v = product ]; then L=""; else L="build_exp/librepsurf_$v.so"; fi
    REPSURF_HIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 60 > $O/bench_${v}_$i.json 2>$O/bench_${v}.err; python -c "import json;d=json.load(open('$O/bench_${v}_$i.json'));print('$v',d['ms_per_step'])"
done
done 2>&1 | tee $O/ab_step.txt
bash tools/gpu_profile.sh r02 cls > $O/profile_cls.log 2>&1; tail -25 $O/profile_cls.log
bash tools/gpu_profile.sh r02 seg > $O/profile_seg.log 2>&1; tail -8 $O/profile_seg.log
