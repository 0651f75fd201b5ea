cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5s; mkdir -p $O; rm -f $O/modes.txt
for rep in 1 2; do for m in none reserve adopt both; do SPS_PRODUCER_MODE=$m timeout -k 10 300 python3 bench.py --steps 60 --warmup 10 --pipelined --no-cpu-baseline --no-training-leg --no-fp16x2-leg > $O/m_$m.json 2>> $O/err.txt; python3 -c "
import json; d=json.loads(open('$O/m_$m.json').read().strip().splitlines()[-1]); print('$m', round(d['ms_per_step'],3), round(d['pipelined']['ms_per_step'],3), {k:d['helper_streams'][k] for k in ('probes','rejected','unplaced')})" | tee -a $O/modes.txt; done; done
