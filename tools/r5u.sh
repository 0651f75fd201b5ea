cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5u; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --output-format csv --kernel-trace -d $O/kt -o kt -- python3 bench.py --config 5 --steps 4 --warmup 2 --no-cpu-baseline --no-training-leg > $O/bench.log 2>&1
python3 tools/tail_trace.py $O/kt/kt_kernel_trace.csv > $O/config5_tail_timeline.txt 2>&1; cat $O/config5_tail_timeline.txt | cut -c1-150
find $O -name "*kernel_trace.csv" -delete
