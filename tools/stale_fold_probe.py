"""Probe for the eval -> frozen-weight train forwards -> eval sequence: prints who differs from whom."""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spsnet_amd import pointnet2_modules as M, scenes
dev = torch.device("cuda:0")
for fused_train in (True, False):
    M.FUSED_MLP_TRAINING = fused_train
    torch.manual_seed(3)
    mod = M.PointnetSAModuleMSG_WithSampling(
        npoint_list=[256], sample_range_list=[-1], sample_type_list=['D-FPS'], radii=[0.6, 1.2], nsamples=[16, 32],
        mlps=[[4, 16, 16, 32], [4, 32, 32, 64]], use_xyz=True, dilated_group=False, aggregation_mlp=[64], confidence_mlp=[32],
        num_class=3)
    scenes.fill_parameters(mod, 5)
    mod = mod.to(dev)
    for p in mod.parameters():
        p.requires_grad_(False)
    rng = np.random.default_rng(8)
    xyz = torch.from_numpy(rng.uniform(-3, 3, (2, 2048, 3)).astype(np.float32)).to(dev)
    feats = torch.from_numpy((3.0 * rng.normal(size=(2, 4, 2048)) + 1.5).astype(np.float32)).to(dev)
    with torch.no_grad():
        mod.eval()
        first = mod(xyz, feats)[1].clone()
        vers0 = {k: v._version for k, v in mod.named_buffers()}
        mod.train()
        for _ in range(3):
            mod(xyz, feats)
        vers1 = {k: v._version for k, v in mod.named_buffers()}
        print("versions moved:", {k: (vers0[k], vers1[k]) for k in vers0 if vers0[k] != vers1[k]})
        print("versions NOT moved:", [k for k in vers0 if vers0[k] == vers1[k]])
        mod.eval()
        second = mod(xyz, feats)[1].clone()
        twin = copy.deepcopy(mod).eval()
        for m_ in twin.modules():
            for attr in [a for a in vars(m_) if a.startswith("_sps")]:
                delattr(m_, attr)
        want = twin(xyz, feats)[1]
        third = mod(xyz, feats)[1]
        # torch reference of the grouped-MLP-free part is not needed: compare the four
        d = lambda a, b: float((a - b).abs().max())
        print(f"fused_train={fused_train}: |second-first|={d(second, first):.4g} |second-want|={d(second, want):.4g} "
              f"|first-want|={d(first, want):.4g} |third-second|={d(third, second):.4g} max|want|={float(want.abs().max()):.4g}")
