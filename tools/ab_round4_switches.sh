#!/bin/bash
# Same-box A/B of the round-4 kernel changes that still have a switch, on the final build: bench.py with each one switched off.
mkdir -p gpurun_out
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra --gpu-baseline off > gpurun_out/r04g_$name.log 2>&1
  python - <<P
import json
l=[x for x in open('gpurun_out/r04g_$name.log') if x.startswith('{')][-1]
d=json.loads(l); print('%-34s %8.3f img/s  %7.3f ms/step  conv frac %.4f  wgrad frac %.4f' % ('$name', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['wgrad_kernel']['frac']))
P
}
for r in 1 2; do
run default_$r X=1
run conv3d_dgrad_col_off_$r PNSFM_CONV3D_DGRAD_COL=0
run conv3d_wgrad_ring_off_$r PNSFM_CONV3D_WGRAD_RING=0
run block_map_weights_off_$r PNSFM_BLOCK_MAP_WEIGHTS=0
run all_three_off_$r PNSFM_CONV3D_DGRAD_COL=0 PNSFM_CONV3D_WGRAD_RING=0 PNSFM_BLOCK_MAP_WEIGHTS=0
done
