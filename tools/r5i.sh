cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5i; mkdir -p $O
timeout -k 10 300 python3 tools/fps_cluster_small.py 8 16384 4096 "8,8;8,4;4,8;8,6;4,16;2,8" > $O/cluster_small.txt 2>&1; grep -v amdgpu $O/cluster_small.txt
timeout -k 10 300 python3 tools/fps_cluster_small.py 4 16384 4096 "16,4;8,8" >> $O/cluster_small.txt 2>&1; grep -v amdgpu $O/cluster_small.txt | tail -3
