#!/usr/bin/env python3
"""bench.py -- SA-stack throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

One "step" = one full pass of the IA-SSD set-abstraction stack L0-L2 (D-FPS 16 384->4 096,
D-FPS ->1 024, ctr-aware top-k ->512; per layer two ball-query radii, grouping, grouped MLP,
max-pool, aggregation and confidence heads; tools/cfgs/kitti_models/IA-SSD.yaml:35-55) over one
batch of 8 synthetic KITTI-shaped scenes per GPU (BASELINE.json configs[1]).  Inputs are resident
in HBM before the timed region.  Scenes shard over ranks with no data-path collective (weak
scaling, DESIGN.md "multi-GPU"); each rank all-gathers its sampled indices once after the timed
region only to prove the exchange path works.

Prints ONE JSON line (rank 0) with value = total points/s over all ranks, plus
  roofline     -- the dominant kernel (layer-0 FPS) on ALGORITHMIC touched bytes (SURVEY.md 8d:
                  20*N*(m-1) B per scene) over its HIP-event time measured inside the timed region;
  cpu_baseline -- the CPU oracle port of the same stack on the host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "points/sec through SA stack (FPS+ball-query+grouped-MLP), KITTI 16k→512"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="scenes per GPU")
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--dataset", default="kitti-lidar-v1", choices=["kitti-lidar-v1", "uniform-v1"])
    ap.add_argument("--sampler", default="ctr_aware", choices=["ctr_aware", "sss_aware"],
                    help="layer-2 sampler (BASELINE configs[1] / configs[3])")
    ap.add_argument("--mlp-precision", default="fp16x2", choices=["fp32", "fp16x2"],
                    help="grouped-MLP arithmetic: exact fp32 MFMA, or split-fp16 (hi+lo halves, 3 MFMAs, ~1e-6 rel.)")
    ap.add_argument("--no-stream-first-layer", dest="stream_first_layer", action="store_false",
                    help="do not let layer 0's ball query / MLP consume the D-FPS picks while FPS is still running")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipelined", action="store_true",
                    help="also time the same passes with two batches in flight on two streams (informational object)")
    ap.add_argument("--no-pipelined", action="store_true", help=argparse.SUPPRESS)  # accepted for older command lines
    ap.add_argument("--cpu-scenes", type=int, default=8, help="scenes in the bounded CPU-baseline sample")
    return ap.parse_args()


class FpsProbe:
    """HIP events around every layer-0 FPS launch, on the stream the kernel is launched on
    (pointnet2_batch_cuda launches on torch's current stream)."""

    def __init__(self, ext, n_points):
        self.ext, self.n = ext, n_points
        self.orig = ext.farthest_point_sampling_wrapper
        self.orig_publish = ext.fps_publish
        self.pairs = []
        self.on = False
        ext.farthest_point_sampling_wrapper = self
        ext.fps_publish = self.publish

    def __call__(self, b, n, m, points, temp, idx):
        if not (self.on and n == self.n):
            return self.orig(b, n, m, points, temp, idx)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = self.orig(b, n, m, points, temp, idx)
        e.record()
        self.pairs.append((s, e, b, n, m))
        return r

    def publish(self, xyz, temp, idx, progress):
        """Same probe around the publishing launch used by the streamed first layer (same kernel, same stream)."""
        if not (self.on and xyz.shape[1] == self.n):
            return self.orig_publish(xyz, temp, idx, progress)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        self.orig_publish(xyz, temp, idx, progress)
        e.record()
        self.pairs.append((s, e, xyz.shape[0], xyz.shape[1], idx.shape[1]))

    def summary(self):
        if not self.pairs:
            return None
        ms = [s.elapsed_time(e) for s, e, *_ in self.pairs]
        _, _, b, n, m = self.pairs[0]
        return float(np.mean(ms)), float(np.min(ms)), b, n, m


def pmc_traffic_bytes(kernel, b, n, m):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (None if not measured for this shape)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "round1", "pmc_traffic.json"))).get(kernel)
    except (OSError, ValueError):
        return None
    if not rec or (rec["batch"], rec["n"], rec["m"]) != (b, n, m):
        return None
    return (rec["fetch_kb"] + rec["write_kb"]) * 1024.0


def cpu_baseline(layers, args):
    """Oracle port of the same stack on the host cores, on a bounded sample (args.cpu_scenes scenes)."""
    from oracle import cpu_stack
    from spsnet_amd import scenes
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    torch.set_num_threads(min(cores, 32))  # the tiny 1x1 convolutions stop scaling long before 256 threads
    nsc = max(1, args.cpu_scenes)
    xyz, feats = scenes.make_batch(args.dataset, nsc, args.points, seed0=0)
    stds = None
    if args.sampler == "sss_aware":
        stds = np.random.default_rng(99).uniform(0, 40, (nsc, args.points)).astype(np.float32)
    cpu_layers = cpu_stack.cpu_copy(layers)
    t0 = time.perf_counter()
    reps = 0
    while True:
        cpu_stack.sa_stack_cpu(cpu_layers, xyz, feats, stds)
        reps += 1
        el = time.perf_counter() - t0
        if el > 10.0 or reps >= 5:
            break
    return {"value": nsc * args.points * reps / el, "unit": "points/s", "cores": cores, "kind": "port",
            "sample": f"{reps} pass(es) over {nsc} scenes x {args.points} pts, same SA stack: C oracle "
                      f"(OpenMP) for FPS/ball-query/group/top-k + torch CPU fp32 for the grouped MLP, {el:.1f} s"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import spsnet_amd.pointnet2_batch_cuda as ext
    from spsnet_amd import pointnet2_modules as M, sa_stack, scenes
    from spsnet_amd.dist import all_gather_sampled_idx

    from spsnet_amd import fused
    fused.set_precision(args.mlp_precision)
    cfg = sa_stack.scaled_config(sample_methods=['D-FPS', 'D-FPS', args.sampler])
    layers = sa_stack.build_sa_layers(M, cfg, seed=0).to(dev)

    # rank r owns scenes [r*B, (r+1)*B) of the global batch (weak scaling: B per GPU is fixed)
    xyz_np, feat_np = scenes.make_batch(args.dataset, args.batch, args.points, seed0=rank * args.batch)
    xyz = torch.from_numpy(xyz_np).to(dev)
    feats = torch.from_numpy(feat_np).to(dev)
    stds = None
    if args.sampler == "sss_aware":
        stds = torch.from_numpy(np.random.default_rng(99 + rank).uniform(0, 40, (args.batch, args.points))
                                .astype(np.float32)).to(dev)

    probe = FpsProbe(ext, args.points)

    def step():
        with torch.no_grad():
            return sa_stack.run_sa_layers(layers, xyz, feats, stds, stream_first_layer=args.stream_first_layer)

    for _ in range(args.warmup):
        outs = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    probe.on = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    probe.on = False

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # the one exchange the sharded path has: global view of every layer's sampled indices
        gathered = all_gather_sampled_idx([o[3] for o in outs])
        assert gathered[0].shape[0] == world * args.batch

    # Extra, reported separately (never `value`): the same K complete passes with TWO batches in flight on two
    # streams -- layer-0 FPS keeps one CU per scene busy for most of a pass, so a second pass fits beside it.
    pipelined = None
    if world == 1 and args.pipelined and not args.no_pipelined:
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        for s_ in streams:
            s_.wait_stream(torch.cuda.current_stream(dev))
        keep = []
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            with torch.cuda.stream(streams[i % 2]):
                keep.append(step())
                if len(keep) > 4:
                    keep.pop(0)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        pipelined = {"batches_in_flight": 2, "value": args.batch * args.points * args.steps / el2, "unit": "points/s",
                     "ms_per_step": 1e3 * el2 / args.steps,
                     "note": "same complete, independent passes issued round-robin on two HIP streams; informational, "
                             "`value` above is the strictly sequential figure"}

    fps = probe.summary()
    if rank == 0:
        total_points = world * args.batch * args.points * args.steps
        line = {
            "metric": METRIC, "value": total_points / elapsed, "unit": "points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.mlp_precision == "fp32" else "f32 (wide grouped-MLP scales as split-fp16 hi+lo pairs on MFMA, fp32 accumulate, <=2e-5 rel. vs fp32)",
            "data": "synthetic",
            "config": {"workload": f"batch={args.batch}/GPU x {args.points} pts ({args.dataset}), IA-SSD SA L0-L2 "
                                   f"(4096/1024/512 centroids, nsample 16&32, layer-2 sampler {args.sampler}), fp32 tensors, "
                                   f"grouped MLP {args.mlp_precision}",
                       "global_batch": world * args.batch, "points_per_scene": args.points,
                       "parallelism": f"scene-sharded x{world}, no data-path collective"},
        }
        if fps is not None:
            mean_ms, min_ms, b, n, m = fps
            touched = 20.0 * n * (m - 1) * b  # 12 B xyz + 4 B read + 4 B write per point per iteration
            ach = touched / (mean_ms * 1e-3) / 1e9
            line["roofline"] = {"bound": "hbm", "kernel": "fps_pruned_kernel<32> (layer-0 D-FPS, 16384->4096)",
                                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                "traffic": pmc_traffic_bytes("fps_pruned_kernel<32>", b, n, m),
                                "launch_ms": mean_ms, "launch_ms_min": min_ms,
                                "note": "effective bandwidth on ALGORITHMIC touched bytes 20*N*(m-1)*B (what the reference "
                                        "kernel moves through L2); this kernel keeps points and running distances in VGPRs "
                                        "and prunes untouched buckets, so its real HBM traffic (`traffic`, bytes per launch "
                                        "from rocprofv3 FETCH_SIZE+WRITE_SIZE, profiles/round1/pmc_traffic.json) is the "
                                        "compulsory 16*N+4*m B/scene and the kernel is latency-, not bandwidth-bound"}
        if pipelined is not None:
            line["pipelined"] = pipelined
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(layers, args)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
