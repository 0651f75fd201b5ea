# bench.py --pipelined (two CU-fenced passes in flight) with and without the reserved producer queue of spsnet_amd/streams.py.
# (profiles/round5/r5s_producer_queue_modes.txt also holds two placements that were built for this A/B and removed again: the fence's
#  own stream adopted as the pass's exclusive helper, and both.)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5s; mkdir -p $O; rm -f $O/modes.txt
for rep in 1 2; do for m in 0 1; do SPS_RESERVE_PRODUCER=$m timeout -k 10 300 python3 bench.py --steps 60 --warmup 10 --pipelined --no-cpu-baseline --no-training-leg --no-fp16x2-leg > $O/m_$m.json 2>> $O/err.txt; python3 -c "
import json; d=json.loads(open('$O/m_$m.json').read().strip().splitlines()[-1]); print('SPS_RESERVE_PRODUCER=$m', round(d['ms_per_step'],3), round(d['pipelined']['ms_per_step'],3), {k:d['helper_streams'][k] for k in ('probes','rejected','unplaced')})" | tee -a $O/modes.txt; done; done
