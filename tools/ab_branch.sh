mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_round4.py -x -q -k "branch_stream or golden_size" 2>&1 | tail -5
for r in 1 2; do for m in 0 1; do
PNSFM_BRANCH_STREAM=$m timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra --gpu-baseline off > gpurun_out/r04c_branch_${m}_$r.log 2>&1
python - <<P
import json
l=[x for x in open('gpurun_out/r04c_branch_${m}_$r.log') if x.startswith('{')][-1]
d=json.loads(l); print('branch=$m run=$r', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('achieved'), d.get('wgrad_kernel',{}).get('frac'))
P
done; done
