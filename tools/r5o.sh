cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5o; mkdir -p $O
rm -rf /tmp/pcb && cp -r spsnet_amd/csrc/_build /tmp/pcb && cd spsnet_amd/csrc && \
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -std=c++17 -DSPS_PC_PROFILE -c fps_pruned_cluster.hip -o /tmp/pcb/fps_pruned_cluster.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libspsnet_sa_pcprof.so /tmp/pcb/*.o && cd $GRAFT_REPO_ROOT && \
for s in 16,4 8,8; do SPS_LIBSPSNET_SA=/tmp/libspsnet_sa_pcprof.so timeout -k 10 120 python3 tools/fps_cluster_profile.py 180000 16384 $s >> $O/cluster_profile_180k.txt 2>&1; done
SPS_FPS_CLUSTER_XCD=0 SPS_LIBSPSNET_SA=/tmp/libspsnet_sa_pcprof.so timeout -k 10 120 python3 tools/fps_cluster_profile.py 180000 16384 16,4 >> $O/cluster_profile_180k.txt 2>&1
grep -v amdgpu $O/cluster_profile_180k.txt
timeout -k 10 600 python3 bench.py --config 5 --steps 20 --warmup 5 --no-training-leg > $O/bench_config5.json 2> $O/bench_config5.err; python3 -c "
import json; d=json.loads(open('$O/bench_config5.json').read().strip().splitlines()[-1]); print('config5', d['value']/1e6, d['ms_per_step'], d['roofline']['launch_ms'], d['validated']['oracle']['ok'], d['roofline_mlp']['mfma_busy_frac'])"
