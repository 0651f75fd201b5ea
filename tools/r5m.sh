cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5m; mkdir -p $O
for x in 1 0; do echo "## SPS_FPS_CLUSTER_XCD=$x" >> $O/xcd_ab.txt
SPS_FPS_CLUSTER_XCD=$x timeout -k 10 300 python3 tools/fps_cluster_probe.py 180000 16384 1 "16,4;8,8" >> $O/xcd_ab.txt 2>&1
SPS_FPS_CLUSTER_XCD=$x timeout -k 10 300 python3 tools/fps_cluster_probe.py 65536 16384 2 "16,4" >> $O/xcd_ab.txt 2>&1
SPS_FPS_CLUSTER_XCD=$x timeout -k 10 300 python3 tools/fps_cluster_small.py 8 16384 4096 "8,8;8,4;4,8" >> $O/xcd_ab.txt 2>&1
done; grep -v amdgpu $O/xcd_ab.txt
timeout -k 10 900 python3 -m pytest tests/test_parity_gpu.py -x -q -k "fps_large_scene_kernel or exchange or large_scene_degenerate or config5" > $O/fps_tests.log 2>&1; echo "rc=$?" >> $O/fps_tests.log; tail -3 $O/fps_tests.log
