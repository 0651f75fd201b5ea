# Round-5 profile set (run on the GPU box through gpurun; summaries land in gpurun_out/prof5, copy what is judged to profiles/round5).
# Headline = strict fp32.  Counter passes serialise the kernels of different streams: a bounded wait dispatched between the producer's
# sorting pre-pass and its FPS kernel would spin to its bound, so the pre-pass is off for them -- the counted kernels are unaffected.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof5
rm -rf $O; mkdir -p $O
cd $R
PMC="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
QUIET="--no-cpu-baseline --no-training-leg"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 20 --warmup 5 --no-training-leg > $O/bench_profiled.log 2>&1 &&
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt_uniform -o kt -- python3 bench.py --dataset uniform-v1 --steps 20 --warmup 5 --no-training-leg --no-cpu-baseline > $O/bench_uniform_profiled.log 2>&1 &&
SPS_FPS_PRESORT=0 rocprofv3 --output-format csv --kernel-trace --pmc $PMC -d $O/pmc32 -o p -- python3 bench.py --steps 4 --warmup 1 --no-fp16x2-leg $QUIET > $O/pmc32.log 2>&1 &&
SPS_FPS_PRESORT=0 rocprofv3 --output-format csv --kernel-trace --pmc $PMC -d $O/pmc16 -o p -- python3 bench.py --steps 4 --warmup 1 --mlp-precision fp16x2 --no-fp32-leg $QUIET > $O/pmc16.log 2>&1 &&
rocprofv3 --output-format csv --kernel-trace --pmc $PMC -d $O/pmc5 -o p -- python3 bench.py --config 5 --steps 3 --warmup 1 --no-validate $QUIET > $O/pmc5.log 2>&1 &&
SPS_FPS_PRESORT=0 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p -- python3 bench.py --steps 3 --warmup 1 --no-fp16x2-leg $QUIET > $O/fetch.log 2>&1 &&
SPS_FPS_PRESORT=0 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/write -o p -- python3 bench.py --steps 3 --warmup 1 --no-fp16x2-leg $QUIET > $O/write.log 2>&1
echo "profiled passes rc=$?"
python3 bench.py --steps 60 --warmup 10 > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"
python3 bench.py --dataset uniform-v1 --steps 60 --warmup 10 > $O/bench_uniform.log 2>> $O/bench.err
python3 bench.py --config 4 --steps 40 --warmup 10 --no-cpu-baseline --no-training-leg > $O/bench_config4.log 2>> $O/bench.err
python3 bench.py --config 5 --steps 20 --warmup 5 --no-training-leg > $O/bench_config5.log 2>> $O/bench.err
python3 bench.py --steps 20 --warmup 5 --force-exchange --no-fp16x2-leg $QUIET > $O/bench_one_rank_rccl.log 2>> $O/bench.err
SPS_BENCH_REHEARSAL=one-gpu python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-fp16x2-leg > $O/bench_two_rank_rehearsal.log 2>> $O/bench.err
python3 tools/tail_events.py > $O/tail_events.txt 2>&1
python3 tools/backbone_time.py 8 16384 20 > $O/backbone_time_fp32.txt 2>&1
SPS_MLP_PRECISION=fp16x2 BACKBONE_GRAPH=0 python3 tools/backbone_time.py 8 16384 20 > $O/backbone_time_fp16x2.txt 2>&1
python3 tools/backbone_train_time.py 8 16384 5 IASSD > $O/backbone_train_time_fp32.txt 2>&1
SPS_TRAIN_PRECISION=fp16x2 python3 tools/backbone_train_time.py 8 16384 5 IASSD > $O/backbone_train_time_fp16x2.txt 2>&1
python3 tools/host_time.py > $O/host_time.txt 2>&1
python3 tools/pmc_mfma.py $O/pmc32/p_counter_collection.csv fp32 $O/pmc_mfma.json $O/pmc16/p_counter_collection.csv:fp16x2 $O/pmc5/p_counter_collection.csv:fp16 > $O/pmc_mfma.txt 2>&1
python3 tools/pmc_traffic.py $O/fetch/p_counter_collection.csv $O/write/p_counter_collection.csv $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
cat $O/pmc_mfma.txt $O/pmc_traffic.txt; tail -c 400 $O/bench.log
# keep the summaries only (the raw rocprofv3 databases exceed what gpurun copies back)
find $O -type f ! -name "*.csv" ! -name "*.log" ! -name "*.json" ! -name "*.txt" ! -name "*.err" -delete
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
du -sh $O
