cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5g; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_parity_gpu.py -x -q -k "fps_large_scene_kernel or exchange or large_scene_degenerate or config5 or streamed_first_layer_survives" > $O/fps_tests.log 2>&1; echo "rc=$?" >> $O/fps_tests.log; tail -4 $O/fps_tests.log
timeout -k 10 600 python3 tools/fps_cluster_probe.py 180000 16384 1 "8,4;16,4;16,3;8,8;12,5;16,2;8,6" > $O/cluster_probe_180k.txt 2>&1; grep -v amdgpu $O/cluster_probe_180k.txt
timeout -k 10 600 python3 tools/fps_cluster_probe.py 65536 16384 2 "8,4;16,4;8,8;16,3" > $O/cluster_probe_65k.txt 2>&1; grep -v amdgpu $O/cluster_probe_65k.txt
timeout -k 10 600 python3 tools/fps_cluster_probe.py 32768 8192 8 "8,4;8,8;8,6" > $O/cluster_probe_32k.txt 2>&1; grep -v amdgpu $O/cluster_probe_32k.txt
timeout -k 10 600 python3 tools/fps_cluster_probe.py 180000 16384 4 "8,4;16,4;8,8" > $O/cluster_probe_180k_b4.txt 2>&1; grep -v amdgpu $O/cluster_probe_180k_b4.txt
