cd $GRAFT_REPO_ROOT; O=gpurun_out/r5d; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -x -q -k "graph or golden_backbone" > $O/graph_tests.log 2>&1; echo "rc=$?" >> $O/graph_tests.log; tail -12 $O/graph_tests.log
timeout -k 10 600 python3 tools/backbone_time.py 8 16384 20 > $O/backbone_time_fp32.txt 2>&1; grep -v amdgpu $O/backbone_time_fp32.txt
