#!/bin/bash
# first GPU session: parity tests, smoke, bench, rocprof kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt
echo "== pytest gpu (goldens + op-level)" 
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout 600 -k "not full_size" 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
echo "== bench"
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; tail -5 gpurun_out/bench.log
echo "== full size tests"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout 800 -k "full_size" 2>&1 | tail -30 > gpurun_out/pytest_gpu_full.log
tail -12 gpurun_out/pytest_gpu_full.log
echo "== rocprof"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; tail -3 gpurun_out/rocprof.log; find gpurun_out/prof_r01 -name "*stats*" | head
