cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5f; mkdir -p $O
./tools/microbench_valu_pk > $O/microbench_valu_pk_gfx950.txt 2>&1; cat $O/microbench_valu_pk_gfx950.txt
rm -rf /tmp/pcb && cp -r spsnet_amd/csrc/_build /tmp/pcb && cd spsnet_amd/csrc && \
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -std=c++17 -DSPS_PC_PROFILE -c fps_pruned_cluster.hip -o /tmp/pcb/fps_pruned_cluster.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libspsnet_sa_pcprof.so /tmp/pcb/*.o && cd $GRAFT_REPO_ROOT && \
for s in 8,4 8,3 16,2 4,8; do SPS_LIBSPSNET_SA=/tmp/libspsnet_sa_pcprof.so timeout -k 10 120 python3 tools/fps_cluster_profile.py 180000 16384 $s >> $O/cluster_profile_180k.txt 2>&1; done
grep -v amdgpu $O/cluster_profile_180k.txt
