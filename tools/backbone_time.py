"""Time IASSD_Backbone / PAGNet_Backbone / PointNet2MSG.forward (spsnet_amd/backbones.py) at the KITTI configuration.
usage: python tools/backbone_time.py [B] [N] [reps] [name prefix]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spsnet_amd import backbones as BB, scenes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
only = sys.argv[4] if len(sys.argv) > 4 else ""
dev = torch.device("cuda:0")
xyz, feats = scenes.make_batch("kitti-lidar-v1", B, N, seed0=1)
bidx = np.repeat(np.arange(B, dtype=np.float32), N)[:, None]
points = torch.from_numpy(np.concatenate([bidx, xyz.reshape(-1, 3), feats.transpose(0, 2, 1).reshape(-1, 1)], 1).astype(np.float32)).to(dev)
stds = torch.from_numpy(np.random.default_rng(0).uniform(0, 40, (B, N)).astype(np.float32)).to(dev)
for tag, cls, cfg in (("IASSD_Backbone", BB.IASSD_Backbone, BB.IASSD_KITTI_CFG), ("PAGNet_Backbone", BB.PAGNet_Backbone, BB.SPSNET_KITTI_CFG),
                      ("PointNet2MSG", BB.PointNet2MSG, BB.POINTRCNN_KITTI_CFG)):
    if only and not tag.startswith(only):
        continue
    kw = {} if tag == "PointNet2MSG" else {"num_class": 3}
    net = scenes.fill_parameters(cls(cfg, input_channels=4, **kw), 5).to(dev).eval()
    def batch():
        d = dict(batch_size=B, points=points)
        if tag.startswith("PAG"):
            d["stds"] = stds
        return d
    with torch.no_grad():
        for _ in range(5):
            net(batch())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            net(batch())
        torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / reps
    # host side alone: how long the Python call takes to enqueue one forward on an idle GPU (includes the one blocking
    # read of the equal-size verdict), and the latency of a single forward
    host = lat = 0.0
    with torch.no_grad():
        for _ in range(reps):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            net(batch())
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            host += t2 - t1
            lat += time.perf_counter() - t1
    print(f"{tag:16s} {B}x{N}: {ms:7.3f} ms per forward  ({B * N / ms / 1e3:.1f} M points/s); host enqueue "
          f"{1e3 * host / reps:.3f} ms, single-forward latency {1e3 * lat / reps:.3f} ms", flush=True)
    if os.environ.get("BACKBONE_GRAPH", "1") != "0":
        # the same forward captured WARM into a HIP graph (spsnet_amd.graphs): one host call per forward
        from spsnet_amd import graphs
        try:
            g = graphs.graphed_backbone(net, B, points, stds if tag.startswith("PAG") else None)
            args = (points, stds) if tag.startswith("PAG") else (points,)
            with torch.no_grad():
                ref = net(batch())
            out = g(*args)
            torch.cuda.synchronize()
            same = all(torch.equal(out[k], ref[k]) for k in ("centers", "centers_features", "ctr_offsets") if k in ref) if "centers" in ref \
                else torch.equal(out["point_features"], ref["point_features"])
            for _ in range(3):
                g(*args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                g(*args)
            torch.cuda.synchronize()
            gms = 1e3 * (time.perf_counter() - t0) / reps
            ghost = 0.0
            for _ in range(reps):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                g(*args)
                ghost += time.perf_counter() - t1
            torch.cuda.synchronize()
            print(f"{'':16s} warm-graph replay: {gms:7.3f} ms per forward, host {1e3 * ghost / reps:.3f} ms per replay call; "
                  f"outputs identical to the eager forward: {same}", flush=True)
        except Exception as exc:  # noqa: BLE001
            print(f"{'':16s} warm-graph replay: FAILED {type(exc).__name__}: {str(exc)[:300]}", flush=True)
