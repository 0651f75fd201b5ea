cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5r; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 300 python3 bench.py --gpus 1 --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value']/1e6, d['ms_per_step'], d['validated']['oracle']['ok'], d['roofline']['frac'], d['cpu_baseline']['value'])"
