"""Time ONE grouped-MLP launch (gather + 3 layers + max-pool) at a given shape, HIP events over 30 launches.
Default: IA-SSD layer 5 (8 scenes, 512 points with 256 features -> 256 vote centres), both scales.
usage: python tools/mlp_time.py [B N M c_feat [l1|l2|l5 [fp32|fp16x2]]]   (l1: the widths / radii of IA-SSD layer 1, e.g. 8 4096 1024 64 l1;
SPS_PM_FP32=0 selects the channel-major fp32 kernel for an A/B)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spsnet_amd import fused, pointnet2_modules as M, pointnet2_utils as U

B, N, Mc, c_feat = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 512, 256, 256)
if len(sys.argv) > 6:
    fused.set_precision(sys.argv[6])
dev = torch.device("cuda:0")
torch.manual_seed(0)
rng = np.random.default_rng(0)
xyz = torch.from_numpy(rng.uniform(-20, 20, (B, N, 3)).astype(np.float32)).to(dev)
feats = fused.attach_point_major_twin(torch.randn(B, c_feat, N, device=dev))
new_xyz = xyz[:, :Mc].contiguous()
SCALES = (([256, 256, 512], 16, 4.8), ([256, 512, 1024], 32, 6.4))
if len(sys.argv) > 5 and sys.argv[5] == "l1":
    SCALES = (([64, 64, 128], 16, 0.8), ([64, 96, 128], 32, 1.6))
if len(sys.argv) > 5 and sys.argv[5] == "l2":       # e.g. 8 1024 512 128 l2
    SCALES = (([128, 128, 256], 16, 1.6), ([128, 256, 256], 32, 4.8))
for widths, ns, radius in SCALES:
    mod = M.PointnetSAModuleMSG_WithSampling(
        npoint_list=[Mc], sample_range_list=[-1], sample_type_list=['D-FPS'], radii=[radius], nsamples=[ns],
        mlps=[[c_feat] + widths], use_xyz=True, dilated_group=False, aggregation_mlp=None, confidence_mlp=None,
        num_class=3).to(dev).eval()
    with torch.no_grad():
        plan = mod._fused_plan(xyz, new_xyz, feats)
        assert plan, "no fused kernel for this scale"
        idx = U.ball_query(radius, ns, xyz, new_xyz)
        out = torch.zeros((B, widths[-1], Mc), device=dev)
        kw = {}
        if os.environ.get("MLP_TIME_HOIST", "0") == "1":   # layer 1's feature product once per point (timed separately)
            rows = fused.layer1_per_point(feats, plan[0])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                fused.layer1_per_point(feats, plan[0], out=rows)
            e1.record()
            torch.cuda.synchronize()
            print(f"   layer1_per_point: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us")
            kw = dict(hoisted=rows)
        for _ in range(3):
            fused.group_mlp_pool(xyz, new_xyz, feats, idx, plan[0], out, 0, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fused.group_mlp_pool(xyz, new_xyz, feats, idx, plan[0], out, 0, **kw)
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    flop = 2.0 * B * Mc * ns * ((c_feat + 3) * widths[0] + widths[0] * widths[1] + widths[1] * widths[2])
    print(f"[{fused.PRECISION} split={plan[0].split} pm={plan[0].point_major}] "
          f"{c_feat + 3}->{widths} ns {ns}: {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s algorithmic "
          f"(split-fp16: x3 issued = {3 * flop / us / 1e6 / 2500 * 100:.1f} % of the 2.5 PF peak)", flush=True)
