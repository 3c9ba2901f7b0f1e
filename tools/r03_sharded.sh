#!/bin/bash
# round 3: the sharded step with a 1-rank RCCL group and the collective forced (captured / between / 2 buckets) -> profiles/r03/sharded_time.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
timeout 600 python -m pytest tests/test_graph_gpu.py -q -m gpu --timeout 300 -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log | cut -c 1-300
echo "== captured collective (1-rank RCCL, forced)"; REPSURF_FORCE_ALLREDUCE=1 timeout 300 python tools/sharded_time.py pipe pipe_sharded pipe_sharded > $O/captured.log 2>&1; echo rc=$?
grep "ms/step\|rror" $O/captured.log | cut -c 1-300
echo "== eager collective between graphs (round-2 form)"; REPSURF_FORCE_ALLREDUCE=1 REPSURF_CAPTURE_ALLREDUCE=0 timeout 300 python tools/sharded_time.py pipe pipe_sharded two_graph > $O/eager.log 2>&1; echo rc=$?
grep "ms/step\|rror" $O/eager.log | cut -c 1-300
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
echo "== 1 bucket captured"; REPSURF_FORCE_ALLREDUCE=1 timeout 300 python tools/sharded_time.py pipe pipe_sharded > $O/b1.log 2>&1; echo rc=$?; grep "ms/step\|rror" $O/b1.log | cut -c 1-300
echo "== 2 buckets captured (bucket 0 issued from the backward hook)"; REPSURF_GRAD_BUCKETS=2 REPSURF_FORCE_ALLREDUCE=1 timeout 300 python tools/sharded_time.py pipe_sharded pipe_sharded > $O/b2.log 2>&1; echo rc=$?; grep "ms/step\|rror" $O/b2.log | cut -c 1-300; grep -v "^frame" $O/b2.log | tail -5 | cut -c 1-300
timeout 600 python -m pytest tests/test_graph_gpu.py -q -m gpu --timeout 300 -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log | cut -c 1-200
