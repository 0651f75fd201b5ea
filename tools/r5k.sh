# (the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round5.sh, run separately after its config-5 counter pass had stopped the chain)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof5b; rm -rf $O; mkdir -p $O; cd $R
QUIET="--no-cpu-baseline --no-training-leg"
SPS_FPS_PRESORT=0 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p -- python3 bench.py --steps 3 --warmup 1 --no-fp16x2-leg $QUIET > $O/fetch.log 2>&1 &&
SPS_FPS_PRESORT=0 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/write -o p -- python3 bench.py --steps 3 --warmup 1 --no-fp16x2-leg $QUIET > $O/write.log 2>&1
echo "rc=$?"
python3 tools/pmc_traffic.py $O/fetch/p_counter_collection.csv $O/write/p_counter_collection.csv $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1; cat $O/pmc_traffic.txt
find $O -type f ! -name "*.csv" ! -name "*.log" ! -name "*.json" ! -name "*.txt" -delete
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
