cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5t; mkdir -p $O
timeout -k 10 300 python3 bench.py --gpus 1 --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value']/1e6, d['ms_per_step'], d['validated']['oracle']['ok'], d['pipelined'], d['training_step']['ms'], d['backbone_forward']['backbone_forward_ms_fp32'])"
timeout -k 10 600 python3 -m pytest tests/test_bench_launch_gpu.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
