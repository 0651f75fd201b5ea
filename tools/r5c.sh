cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_parity_gpu.py -x -q -k "fps_large_scene_kernel or presorted or exchange or large_scene_degenerate or fps_kitti_full_size or streamed_first_layer" > $O/fps_tests.log 2>&1; echo "rc=$?" >> $O/fps_tests.log; tail -4 $O/fps_tests.log
timeout -k 10 600 python3 tools/fps_cluster_probe.py 180000 16384 1 "8,4;16,2;16,1;12,2;8,3" > $O/cluster_probe_180k.txt 2>&1; cat $O/cluster_probe_180k.txt | grep -v amdgpu
timeout -k 10 600 python3 tools/fps_cluster_probe.py 65536 16384 2 "8,4;16,2" > $O/cluster_probe_65k.txt 2>&1; cat $O/cluster_probe_65k.txt | grep -v amdgpu
for m in streamed seq; do
  GRAPH_TRY_WARM=1 GRAPH_TRY_HOST=1 timeout -k 10 300 python3 tools/graph_try.py $m fp32 >> $O/graph_warm.txt 2>&1
  GRAPH_TRY_WARM=1 GRAPH_TRY_HOST=1 GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python3 tools/graph_try.py $m fp32 >> $O/graph_warm.txt 2>&1
done
GRAPH_TRY_HOST=1 timeout -k 10 300 python3 tools/graph_try.py streamed fp32 >> $O/graph_warm.txt 2>&1
grep -v amdgpu $O/graph_warm.txt
