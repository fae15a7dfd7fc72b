cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $O/prof_tl -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --gpu-baseline off --no-prof > $O/r05_tl.log 2>&1
t=$(find $O/prof_tl -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $t 5 | tee $O/r05_step_timeline.txt
python $R/tools/step_timeline.py $t 6 | head -3
tail -1 $O/r05_tl.log | cut -c1-200
rm -rf $O/prof_tl
