import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import cpu_stack
from spsnet_amd import pointnet2_modules as M, sa_stack, scenes
cfg = sa_stack.scaled_config(npoints=[2048, 512, 128], nsamples=[(64, 64)] * 3)
layers = sa_stack.build_sa_layers(M, cfg, seed=5)
xyz, feats = scenes.make_batch("kitti-lidar-v1", 1, 20000, seed0=3)
t0 = time.time(); want = cpu_stack.sa_stack_cpu(cpu_stack.cpu_copy(layers), xyz, feats); print("cpu", time.time() - t0)
dev = torch.device("cuda:0"); layers = layers.to(dev)
with torch.no_grad():
    got = sa_stack.run_sa_layers(layers, torch.from_numpy(xyz).to(dev), torch.from_numpy(feats).to(dev))
torch.cuda.synchronize()
for k in range(3):
    same = np.array_equal(got[k][3].cpu().numpy(), want[k][3])
    err = np.abs(got[k][1].cpu().numpy() - want[k][1]).max() / max(1.0, np.abs(want[k][1]).max())
    print(k, "idx equal:", same, "feat rel err:", err)
