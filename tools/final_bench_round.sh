cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof2
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 20 --warmup 5 --no-training-leg > $O/bench_profiled.log 2>&1 &&
python3 bench.py --steps 60 --warmup 10 > $O/bench.log 2>&1
find $O -type f ! -name "*.csv" ! -name "*.log" -delete
find $O -name "*kernel_trace.csv" -delete
