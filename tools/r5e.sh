cd $GRAFT_REPO_ROOT; O=gpurun_out/r5e; mkdir -p $O
timeout -k 10 600 python3 tools/pipelined_graph.py 1,2,3,4,6,8 80 > $O/pipelined_graph.txt 2>&1; grep -v amdgpu $O/pipelined_graph.txt
GPU_MAX_HW_QUEUES=8 timeout -k 10 600 python3 tools/pipelined_graph.py 2,4,8 80 > $O/pipelined_graph_q8.txt 2>&1; grep -v amdgpu $O/pipelined_graph_q8.txt
timeout -k 10 600 python3 bench.py --steps 40 --warmup 5 --pipelined --no-cpu-baseline --no-training-leg --no-fp16x2-leg > $O/bench_pipelined.json 2> $O/bench_pipelined.err; python3 -c "
import json; d=json.loads(open('$O/bench_pipelined.json').read().strip().splitlines()[-1]); print(d['value']/1e6, d['pipelined'])"
