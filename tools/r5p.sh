cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5p; mkdir -p $O
timeout -k 10 300 python3 tools/fps_cluster_probe.py 180000 16384 1 "16,4;8,8" >> $O/early_fetch.txt 2>&1
timeout -k 10 300 python3 tools/fps_cluster_probe.py 65536 16384 2 "16,4" >> $O/early_fetch.txt 2>&1
timeout -k 10 300 python3 tools/fps_cluster_probe.py 32768 8192 8 "8,8" >> $O/early_fetch.txt 2>&1
grep -v amdgpu $O/early_fetch.txt
timeout -k 10 900 python3 -m pytest tests/test_parity_gpu.py -x -q -k "fps_large_scene_kernel or exchange or large_scene_degenerate or config5" > $O/fps_tests.log 2>&1; echo "rc=$?" >> $O/fps_tests.log; tail -3 $O/fps_tests.log
