# Round 5, first call: the GPU suite on the round's first build, the default line (now with validated.oracle), SURVEY 8d's second
# dataset (uniform-v1) as a bench line + kernel statistics + the FPS phase profile on both datasets.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5a
rm -rf $O; mkdir -p $O
cd $R
rm -f gpurun_out/parity_achieved_error.jsonl
python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -3 $O/tests.log
cp gpurun_out/parity_achieved_error.jsonl $O/ 2>/dev/null
python3 bench.py --steps 60 --warmup 10 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 bench.py --dataset uniform-v1 --steps 60 --warmup 10 > $O/bench_uniform.json 2> $O/bench_uniform.err; echo "bench uniform rc=$?"
python3 tools/fps_profile.py 16384 4096 kitti-lidar-v1 > $O/fps_profile_kitti.txt 2>&1
python3 tools/fps_profile.py 16384 4096 uniform-v1 > $O/fps_profile_uniform.txt 2>&1
python3 tools/tail_events.py > $O/tail_events.txt 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt_uniform -o kt -- python3 bench.py --dataset uniform-v1 --steps 20 --warmup 5 --no-training-leg --no-cpu-baseline > $O/bench_uniform_profiled.json 2> $O/bench_uniform_profiled.err
find $O -type f ! -name "*.csv" ! -name "*.log" ! -name "*.json" ! -name "*.txt" ! -name "*.err" ! -name "*.jsonl" -delete
find $O -name "*kernel_trace.csv" -size +8M -delete
tail -c 600 $O/bench.json; echo; tail -c 300 $O/bench_uniform.json; echo; cat $O/fps_profile_uniform.txt | tail -3
