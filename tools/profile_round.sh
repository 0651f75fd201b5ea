cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd $R
# (counter passes serialise the kernels of different streams: a bounded wait dispatched between the producer's sorting pre-pass
#  and its FPS kernel would spin to its bound, so the pre-pass is off for them -- the counted kernels are unaffected)
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 20 --warmup 5 --no-training-leg > $O/bench_profiled.log 2>&1 &&
SPS_FPS_PRESORT=0 rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc16 -o p -- python3 bench.py --steps 4 --warmup 1 --no-fp32-leg --no-cpu-baseline --no-training-leg > $O/pmc16.log 2>&1 &&
SPS_FPS_PRESORT=0 rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc32 -o p -- python3 bench.py --steps 4 --warmup 1 --mlp-precision fp32 --no-cpu-baseline --no-training-leg > $O/pmc32.log 2>&1 &&
SPS_FPS_PRESORT=0 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p -- python3 bench.py --steps 3 --warmup 1 --no-fp32-leg --no-cpu-baseline --no-training-leg > $O/fetch.log 2>&1 &&
SPS_FPS_PRESORT=0 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/write -o p -- python3 bench.py --steps 3 --warmup 1 --no-fp32-leg --no-cpu-baseline --no-training-leg > $O/write.log 2>&1 &&
python3 bench.py --steps 60 --warmup 10 > $O/bench.log 2>&1
find $O -name "*.csv" | head -30; tail -2 $O/pmc16.log | cut -c1-300
# keep the summaries only (the raw rocprofv3 databases exceed what gpurun copies back)
find $O -type f ! -name "*.csv" ! -name "*.log" -delete
find $O -name "*kernel_trace.csv" -size +8M -delete
du -sh $O
