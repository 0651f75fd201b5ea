cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5q; mkdir -p $O
for s in 31 32 33 34 35 36; do timeout -k 10 300 python3 tests/sweep_parity_gpu.py 800 $s 2>&1 | grep -v amdgpu | tail -1 >> $O/parity_sweep.txt; done
echo "# the same with the clustered FPS forced to its widest shapes (SPS_FPS_CLUSTER=16,4 / 8,8), seeds 41-44" >> $O/parity_sweep.txt
for s in 41 42; do SPS_FPS_CLUSTER=16,4 timeout -k 10 300 python3 tests/sweep_parity_gpu.py 800 $s 2>&1 | grep -v amdgpu | tail -1 >> $O/parity_sweep.txt; done
for s in 43 44; do SPS_FPS_CLUSTER=8,8 timeout -k 10 300 python3 tests/sweep_parity_gpu.py 800 $s 2>&1 | grep -v amdgpu | tail -1 >> $O/parity_sweep.txt; done
cat $O/parity_sweep.txt
