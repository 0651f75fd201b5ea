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

`--config 4` selects BASELINE configs[3] (stability top-k at layer 2), `--config 5` the per-GPU share of
configs[4] (1 scene x 180 000 points -> 16 384 / 4 096 / 1 024, nsample 64, fp16 features on MFMA).

Prints ONE JSON line (rank 0) with value = total points/s over all ranks, plus
  validated    -- what was checked about the timed work after the timed region (no progress-wait timeout, no
                  split-fp16 overflow, last step's outputs bit-identical to one plain sequential pass);
  value_fp32   -- the same K steps with the grouped MLP in strict fp32 (`--mlp-precision fp32`), timed the same way;
  ms_per_step_median -- median of the K per-step HIP-event times (ms_per_step is elapsed / K);
  roofline     -- the dominant kernel (layer-0 FPS) on ALGORITHMIC touched bytes (SURVEY.md 8d:
                  20*N*(m-1) B per scene) over its HIP-event time measured inside the timed region;
  roofline_mlp -- the largest grouped-MLP launch: algorithmic FLOP over its HIP-event time (measured after the timed
                  region on the launch's own arguments) and the MFMA-busy fraction from the committed PMC pass;
  cpu_baseline -- the CPU oracle port of the same stack on the host cores, all cores and one thread (rank 0, N=1 only).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "points/sec through SA stack (FPS+ball-query+grouped-MLP), KITTI 16k→512"
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TF = {"fp32": 157.3, "fp16x2": 2500.0, "fp16": 2500.0}  # dense peaks: fp32 MFMA, fp16 MFMA (no sparsity)
PROFILE_DIRS = ("round2", "round1")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # median of >= 50 runs after 10 warm-ups
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=[2, 4, 5],
                    help="BASELINE.json configs[] entry (1-based): 2 = headline, 4 = stability top-k sampler, "
                         "5 = Waymo-shaped 180k-point scenes, nsample 64, fp16 features")
    ap.add_argument("--batch", type=int, default=None, help="scenes per GPU (default 8; config 5: 1)")
    ap.add_argument("--points", type=int, default=None, help="points per scene (default 16384; config 5: 180000)")
    ap.add_argument("--dataset", default="kitti-lidar-v1", choices=["kitti-lidar-v1", "uniform-v1"])
    ap.add_argument("--sampler", default=None, choices=["ctr_aware", "sss_aware"],
                    help="layer-2 sampler (BASELINE configs[1] / configs[3])")
    ap.add_argument("--mlp-precision", default=None, choices=["fp32", "fp16x2", "fp16"],
                    help="grouped-MLP arithmetic: exact fp32 MFMA, split-fp16 (fp32 operands as hi+lo halves, 3 MFMAs, "
                         "<=2e-5 rel.), or fp16 (features stored as fp16, config 5)")
    ap.add_argument("--no-stream-first-layer", dest="stream_first_layer", action="store_false",
                    help="do not let layer 0's ball query / MLP consume the D-FPS picks while FPS is still running")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the strict-fp32 repetition (value_fp32)")
    ap.add_argument("--no-validate", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pipelined", action="store_true",
                    help="also time the same passes with two batches in flight (informational object)")
    ap.add_argument("--no-pipelined", action="store_true", help=argparse.SUPPRESS)  # accepted for older command lines
    ap.add_argument("--no-cu-fence", action="store_true", help="pipelined leg without the CU partition")
    ap.add_argument("--in-flight", type=int, default=2, help="batches in flight in the pipelined leg")
    ap.add_argument("--cu-fence", action="store_true",
                    help="also in the sequential steps: layer-0 FPS on compute units of its own (CuFence)")
    ap.add_argument("--cpu-scenes", type=int, default=8, help="scenes in the bounded CPU-baseline sample")
    ap.add_argument("--no-training-leg", action="store_true",
                    help="skip the informational `training_step` object (forward + backward of the same SA layers in train() mode)")
    args = ap.parse_args()
    if args.config == 5:
        args.batch = 1 if args.batch is None else args.batch
        args.points = 180000 if args.points is None else args.points
        args.mlp_precision = args.mlp_precision or "fp16"
    args.batch = 8 if args.batch is None else args.batch
    args.points = 16384 if args.points is None else args.points
    args.mlp_precision = args.mlp_precision or "fp16x2"
    if args.sampler is None:
        args.sampler = "sss_aware" if args.config == 4 else "ctr_aware"
    return args


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

    def publish(self, xyz, temp, idx, progress, **kw):
        """Same probe around the publishing launch used by the streamed first layer (same stream; the bracket holds the
        producer's sorting pre-pass too, where one is used: the figure is the whole producer, not the kernel alone)."""
        if not (self.on and xyz.shape[1] == self.n):
            return self.orig_publish(xyz, temp, idx, progress, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        work = self.orig_publish(xyz, temp, idx, progress, **kw)
        e.record()
        self.pairs.append((s, e, xyz.shape[0], xyz.shape[1], idx.shape[1]))
        return work

    def summary(self):
        if not self.pairs:
            return None
        ms = [s.elapsed_time(e) for s, e, *_ in self.pairs]
        _, _, b, n, m = self.pairs[0]
        self.pairs = []
        return float(np.mean(ms)), float(np.min(ms)), b, n, m


class MlpProbe:
    """Remembers the arguments of the costliest grouped-MLP launch shape of a pass (by algorithmic FLOP) so that the
    launch can be repeated and timed with HIP events after the timed region, on the stream it is launched on."""

    def __init__(self, fused):
        self.fused = fused
        self.orig = fused.group_mlp_pool
        self.best = None
        fused.group_mlp_pool = self

    def __call__(self, xyz, new_xyz, features, idx, packed, out, channel_offset, j0=0, jcount=None, **kw):
        B, M, ns = idx.shape
        cols = B * (M if jcount is None else jcount) * ns
        flop = 2.0 * cols * (packed.cin * packed.c1 + packed.c1 * packed.c2 + packed.c2 * packed.c3_real)
        if self.best is None or flop > self.best[0]:
            self.best = (flop, (xyz, new_xyz, features, idx, packed, out, channel_offset, j0, jcount), kw)
        return self.orig(xyz, new_xyz, features, idx, packed, out, channel_offset, j0, jcount, **kw)

    def measure(self, reps=20):
        if self.best is None:
            return None
        flop, call, kw = self.best
        for _ in range(3):
            self.orig(*call, **kw)
        pairs = []
        for _ in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            self.orig(*call, **kw)
            e.record()
            pairs.append((s, e))
        torch.cuda.synchronize()
        ms = [s.elapsed_time(e) for s, e in pairs]
        packed, idx = call[4], call[3]
        return {"flop": flop, "ms": float(np.median(ms)), "widths": (packed.cin, packed.c1, packed.c2, packed.c3_real),
                "packed_columns": kw.get("columns") is not None,
                "nsample": int(idx.shape[2]), "columns": int(flop / (2.0 * (packed.cin * packed.c1 + packed.c1 * packed.c2
                                                                           + packed.c2 * packed.c3_real)))}


def _profile_json(name):
    for d in PROFILE_DIRS:
        try:
            return json.load(open(os.path.join(ROOT, "profiles", d, name))), d
        except (OSError, ValueError):
            continue
    return None, None


def pmc_traffic_bytes(kernel, b, n, m):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (None if not measured for this shape)."""
    data, _ = _profile_json("pmc_traffic.json")
    rec = (data or {}).get(kernel)
    if not rec or (rec["batch"], rec["n"], rec["m"]) != (b, n, m):
        return None
    return (rec["fetch_kb"] + rec["write_kb"]) * 1024.0


def pmc_mfma_busy(widths, nsample, precision):
    """MFMA-busy fraction of the grouped-MLP kernel for these widths from the committed PMC pass
    (profiles/roundN/pmc_mfma.json: SQ_VALU_MFMA_BUSY_CYCLES / (duration x SIMDs x clock)), or None."""
    data, where = _profile_json("pmc_mfma.json")
    if not data:
        return None, None
    key = f"{precision}:{widths[1]},{widths[2]},ns{nsample}"
    rec = data.get(key)
    if not rec:
        return None, None
    return rec, f"profiles/{where}/pmc_mfma.json[{key}]"


def cpu_baseline(layers, args):
    """Oracle port of the same stack on the host cores, on a bounded sample: once with all cores, once single-threaded
    (SURVEY.md 8d asks for both)."""
    from oracle import cpu_stack, oracle as O
    from spsnet_amd import scenes
    cores = os.cpu_count() or 1
    cpu_layers = cpu_stack.cpu_copy(layers)

    def sample(nsc, threads, budget_s, max_reps):
        O.set_threads(threads)
        torch.set_num_threads(min(threads, 32))  # the tiny 1x1 convolutions stop scaling long before 256 threads
        xyz, feats = scenes.make_batch(args.dataset, nsc, args.points, seed0=0)
        stds = None
        if args.sampler == "sss_aware":
            stds = np.random.default_rng(99).uniform(0, 40, (nsc, args.points)).astype(np.float32)
        t0 = time.perf_counter()
        reps = 0
        while True:
            cpu_stack.sa_stack_cpu(cpu_layers, xyz, feats, stds)
            reps += 1
            el = time.perf_counter() - t0
            if el > budget_s or reps >= max_reps:
                break
        return nsc * args.points * reps / el, reps, el

    big = args.points > 65536          # one Waymo-sized scene is already ~1 min of CPU work: a single all-core pass
    nsc = max(1, min(args.cpu_scenes, args.batch) if big else args.cpu_scenes)
    v_all, reps_all, el_all = sample(nsc, cores, 10.0, 1 if big else 5)
    what = ("same SA stack: C oracle (OpenMP) for FPS/ball-query/group/top-k + torch CPU fp32 for the grouped MLP")
    out = {"value": v_all, "unit": "points/s", "cores": cores, "kind": "port",
           "sample": f"{reps_all} pass(es) over {nsc} scenes x {args.points} pts, {what}, {el_all:.1f} s"}
    if not big:
        v_one, reps_one, el_one = sample(1, 1, 8.0, 3)
        out["single_thread"] = {"value": v_one, "unit": "points/s", "cores": 1,
                                "sample": f"{reps_one} pass(es) over 1 scene x {args.points} pts, 1 thread, {el_one:.1f} s"}
    O.set_threads(cores)
    return out


def training_step_leg(modules_pkg, sa_stack, cfg, args, xyz, feats, dev, reps=10):
    """Informational, never `value`: one TRAINING step (forward + backward, BatchNorm on batch statistics) of the same SA
    layers on the same batch -- SURVEY 8 f-1.  Two timings: the step as a plain loop runs it, and with the next batch's
    layer-0 sampling started beside the backward (sa_stack.prefetch_first_layer; here the next batch is the same tensor)."""
    layers = sa_stack.build_sa_layers(modules_pkg, cfg, seed=0).to(dev).train()
    f = feats.float()

    def step(prefetch):
        for p in layers.parameters():
            p.grad = None
        outs = sa_stack.run_sa_layers(layers, xyz, f)
        loss = sum(o[1].square().mean() for o in outs) + sum(o[2].square().mean() for o in outs if o[2] is not None)
        if prefetch:
            sa_stack.prefetch_first_layer(layers, xyz)
        loss.backward()

    res = {}
    for key, pre in (("ms", False), ("ms_next_batch_sampling_prefetched", True)):
        layers[0]._presampled = layers[0]._preball = None
        for _ in range(3):
            step(pre)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step(pre)
        torch.cuda.synchronize()
        res[key] = 1e3 * (time.perf_counter() - t0) / reps
    layers[0]._presampled = layers[0]._preball = None
    res["points_per_s"] = xyz.shape[0] * xyz.shape[1] / (res["ms"] * 1e-3)
    res["grouped_mlp"] = ("fused train-mode kernels (csrc/mlp_train.hip), split-fp16 MFMA with exact power-of-two operand scaling"
                          if (modules_pkg.FUSED_MLP_TRAINING and args.mlp_precision != "fp32") else "op-by-op fp32 kernels")
    res["note"] = "forward + backward of SA layers 0-2 in train() mode on the bench batch; informational, not the headline metric"
    return res


def same_outputs(got, want):
    """Bit-for-bit equality of two run_sa_layers results -> (ok, first difference)."""
    for k, (g, w) in enumerate(zip(got, want)):
        for name, a, b in zip(("new_xyz", "new_features", "cls", "sampled_idx"), g, w):
            if a is None and b is None:
                continue
            if a is None or b is None or a.shape != b.shape or not torch.equal(a, b):
                return False, f"layer {k} {name}"
    return True, ""


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 launch on a one-GPU box (SPS_BENCH_REHEARSAL=one-gpu): every rank on cuda:0, gloo collectives
    # (RCCL refuses two ranks on one device).  Same code path otherwise; the line it prints says so and is no measurement.
    rehearsal = world > 1 and os.environ.get("SPS_BENCH_REHEARSAL", "") == "one-gpu"
    if rehearsal:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = 0
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
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
    if args.mlp_precision != "fp16":   # "fp16" is not a mode switch: fp16 FEATURE TENSORS select the pure-fp16 kernels
        fused.set_precision(args.mlp_precision)
    if args.config == 5:
        cfg = sa_stack.scaled_config(npoints=[16384, 4096, 1024], nsamples=[[64, 64]] * 3,
                                     sample_methods=['D-FPS', 'D-FPS', args.sampler])
        shape_txt = "16384/4096/1024 centroids, nsample 64&64"
    else:
        cfg = sa_stack.scaled_config(sample_methods=['D-FPS', 'D-FPS', args.sampler])
        shape_txt = "4096/1024/512 centroids, nsample 16&32"
    layers = sa_stack.build_sa_layers(M, cfg, seed=0).to(dev)

    # rank r owns scenes [r*B, (r+1)*B) of the global batch (weak scaling: B per GPU is fixed)
    xyz_np, feat_np = scenes.make_batch(args.dataset, args.batch, args.points, seed0=rank * args.batch)
    xyz = torch.from_numpy(xyz_np).to(dev)
    feats = torch.from_numpy(feat_np).to(dev)
    if args.mlp_precision == "fp16":
        feats = feats.half()   # features live in HBM as fp16 (BASELINE configs[4]); coordinates stay fp32
    stds = None
    if args.sampler == "sss_aware":
        stds = torch.from_numpy(np.random.default_rng(99 + rank).uniform(0, 40, (args.batch, args.points))
                                .astype(np.float32)).to(dev)

    if args.cu_fence:
        # CU-masked streams are blocking streams: they synchronise with the legacy default stream, so the steps must be
        # issued on a stream of their own for the fenced FPS to run beside anything
        own = torch.cuda.Stream(device=dev)
        own.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(own)
        sa_stack.enable_cu_fence(dev, scenes=args.batch)
    probe = FpsProbe(ext, args.points)
    mlp_probe = MlpProbe(fused)

    def step(**kw):
        with torch.no_grad():
            return sa_stack.run_sa_layers(layers, xyz, feats, stds, stream_first_layer=args.stream_first_layer, **kw)

    def timed(steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks.
        -> (elapsed s, per-step ms from HIP events on the stream the steps are issued on, last outputs)"""
        outs = None
        for _ in range(warmup):
            outs = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        probe.on = True
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            outs = step()
            marks[i + 1].record()
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
        per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
        return elapsed, per_step, outs

    def validate(outs):
        """Prove the timed work: no bounded wait gave up, no split-fp16 operand left the exact range, and the last
        step's outputs equal one plain sequential pass (no streaming, no overlap) bit for bit."""
        if args.no_validate:
            return None
        timeouts = bool(sa_stack.check_timeouts())
        overflow = bool(fused.check_overflow())
        with torch.no_grad():
            ref = sa_stack.run_sa_layers(layers, xyz, feats, stds, overlap=False, stream_first_layer=False)
        torch.cuda.synchronize()
        ok, where = same_outputs(outs, ref)
        rec = {"progress_wait_timeouts": timeouts, "split_fp16_overflow": overflow,
               "last_step_bit_identical_to_sequential_pass": ok}
        if timeouts or overflow or not ok:
            raise SystemExit(f"bench.py: the timed work failed validation: {rec} {where}")
        return rec

    elapsed, per_step, outs = timed(args.steps, args.warmup)
    fps = probe.summary()
    checked = validate(outs)
    if world > 1:
        # the one exchange the sharded path has: global view of every layer's sampled indices
        gathered = all_gather_sampled_idx([o[3] for o in outs])
        assert gathered[0].shape[0] == world * args.batch
    mlp = mlp_probe.measure() if world == 1 or rank == 0 else None

    # the same K steps in strict fp32 (every grouped-MLP scale on the exact fp32 MFMA kernel), timed the same way
    fp32_leg = None
    if not args.no_fp32_leg and args.mlp_precision != "fp32" and args.config != 5:
        fused.set_precision("fp32")
        mlp_probe.best = None            # (the costliest launch of THIS leg: same shape, weights packed for the fp32 kernel)
        el32, per32, outs32 = timed(args.steps, max(args.warmup, 2))
        probe.summary()
        chk32 = validate(outs32)
        mlp32 = mlp_probe.measure() if world == 1 or rank == 0 else None   # the same launch on the exact fp32 MFMA kernel
        fp32_leg = (el32, per32, chk32, mlp32)
        fused.set_precision(args.mlp_precision)

    # Extra, reported separately (never `value`): the same K complete passes with TWO batches in flight on two
    # streams -- layer-0 FPS keeps one CU per scene busy for most of a pass, so a second pass fits beside it.
    pipelined = None
    if world == 1 and args.pipelined and not args.no_pipelined:
        pipelined = sa_stack.pipelined_bench(step, args.steps, dev, in_flight=args.in_flight, scenes=args.batch,
                                             fenced=not args.no_cu_fence)
        pipelined["value"] = args.batch * args.points * args.steps / pipelined.pop("elapsed_s")
        same, where = same_outputs(pipelined.pop("last_outputs"), outs)
        if not same or sa_stack.check_timeouts():
            raise SystemExit(f"bench.py: the pipelined leg's outputs differ from the sequential steps' ({where})")
        pipelined["validated"] = "last pass bit-identical to the sequential steps' outputs, no progress-wait timeout"

    if rank == 0:
        total_points = world * args.batch * args.points * args.steps
        dtype_txt = {"fp32": "f32",
                     "fp16x2": "f32 (wide grouped-MLP scales as split-fp16 hi+lo pairs on MFMA, fp32 accumulate, "
                               "<=2e-5 rel. vs fp32)",
                     "fp16": "f16 features and grouped-MLP operands on MFMA, fp32 accumulate; coordinates, distances "
                             "and indices fp32/int32"}[args.mlp_precision]
        line = {
            "metric": METRIC, "value": total_points / elapsed, "unit": "points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_median": float(statistics.median(per_step)),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_txt,
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{args.config - 1}]: batch={args.batch}/GPU x {args.points} pts "
                                   f"({args.dataset}), IA-SSD SA L0-L2 ({shape_txt}, layer-2 sampler {args.sampler}), "
                                   f"grouped MLP {args.mlp_precision}",
                       "global_batch": world * args.batch, "points_per_scene": args.points,
                       "parallelism": f"scene-sharded x{world}, no data-path collective"},
        }
        if rehearsal:
            line["data"] = "synthetic; REHEARSAL: all ranks share one GPU over gloo -- not a measurement"
        if checked is not None:
            line["validated"] = checked
        if fp32_leg is not None:
            el32, per32, chk32, mlp32 = fp32_leg
            line["value_fp32"] = total_points / el32
            line["ms_per_step_fp32"] = 1e3 * el32 / args.steps
            line["ms_per_step_fp32_median"] = float(statistics.median(per32))
            if chk32 is not None:
                line["validated_fp32"] = chk32
        elif args.mlp_precision == "fp32":
            line["value_fp32"] = line["value"]
        if fps is not None:
            mean_ms, min_ms, b, n, m = fps
            touched = 20.0 * n * (m - 1) * b  # 12 B xyz + 4 B read + 4 B write per point per iteration
            ach = touched / (mean_ms * 1e-3) / 1e9
            # (scenes beyond 16 384 points: spread over up to 8 workgroups each, fps_pruned_cluster.hip)
            kname = "fps_pruned_kernel<32>" if n <= 16384 else ("fps_pruned_cluster_kernel" if b <= 32 else "fps_pruned_big_kernel")
            line["roofline"] = {"bound": "hbm", "kernel": f"{kname} (layer-0 D-FPS, {n}->{m})",
                                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                "traffic": pmc_traffic_bytes(kname, b, n, m),
                                "launch_ms": mean_ms, "launch_ms_min": min_ms,
                                "note": "effective bandwidth on ALGORITHMIC touched bytes 20*N*(m-1)*B (what the reference "
                                        "kernel moves through L2); this kernel keeps points and running distances on chip "
                                        "and prunes untouched buckets, so its real HBM traffic (`traffic`, bytes per launch "
                                        "from rocprofv3 FETCH_SIZE+WRITE_SIZE, profiles/*/pmc_traffic.json) is the "
                                        "compulsory 16*N+4*m B/scene and the kernel is latency-, not bandwidth-bound"}
        def mlp_roofline(mlp, precision):
            peak = MFMA_PEAK_TF[precision]
            ach_tf = mlp["flop"] / (mlp["ms"] * 1e-3) / 1e12
            busy, src = pmc_mfma_busy(mlp["widths"], mlp["nsample"], precision)
            return {
                "bound": "mfma", "kernel": f"grouped MLP {mlp['widths'][0]}->{mlp['widths'][1]}->{mlp['widths'][2]}->"
                                           f"{mlp['widths'][3]}, nsample {mlp['nsample']}, {mlp['columns']} columns",
                "achieved": ach_tf, "peak": peak, "unit": "TFLOP/s", "frac": ach_tf / peak, "launch_ms": mlp["ms"],
                "flop": mlp["flop"], "packed_columns": mlp["packed_columns"],
                "mfma_busy_frac": None if busy is None else busy.get("mfma_busy_frac"),
                "mfma_busy_source": src,
                "note": "achieved = ALGORITHMIC flop 2*columns*sum(Cin*Cout) of the launch (padded duplicate columns "
                        "included, as the reference computes them) / HIP-event time of that launch repeated on its own "
                        "arguments after the timed region; peak = dense MFMA peak of the operand type (fp16x2 issues 3 "
                        "fp16 MFMAs per product block, so its useful ceiling is a third of it); mfma_busy_frac = "
                        "SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 1024 SIMDs x 2.4 GHz) from the committed PMC pass"}
        if mlp is not None:
            line["roofline_mlp"] = mlp_roofline(mlp, args.mlp_precision)
        if fp32_leg is not None and fp32_leg[3] is not None:   # the strict-fp32 leg's dominant launch (fp32 MFMA pipe, 157 TF)
            line["roofline_mlp_fp32"] = mlp_roofline(fp32_leg[3], "fp32")
        if pipelined is not None:
            line["pipelined"] = pipelined
        if world == 1 and not args.no_training_leg and args.config != 5 and args.mlp_precision != "fp16":
            try:
                line["training_step"] = training_step_leg(M, sa_stack, cfg, args, xyz, feats, dev)
            except Exception as exc:   # informational only: never lose the headline line over it
                line["training_step"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(layers, args)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
