cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5l; mkdir -p $O
rm -rf /tmp/pcb && cp -r spsnet_amd/csrc/_build /tmp/pcb && cd spsnet_amd/csrc && \
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -std=c++17 -DSPS_PC_PROFILE -c fps_pruned_cluster.hip -o /tmp/pcb/fps_pruned_cluster.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libspsnet_sa_pcprof.so /tmp/pcb/*.o && cd $GRAFT_REPO_ROOT && \
for s in 8,8 8,4 4,8; do SPS_LIBSPSNET_SA=/tmp/libspsnet_sa_pcprof.so timeout -k 10 120 python3 tools/fps_cluster_profile.py 16384 4096 $s 8 >> $O/cluster_profile_16k.txt 2>&1; done
grep -v amdgpu $O/cluster_profile_16k.txt
