#!/usr/bin/env python3
"""bench.py -- SA-stack throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: ONE command per node, like the reference's tools/scripts/dist_train.sh:1-20 -- when no launcher has set
    WORLD_SIZE, bench.py starts its N ranks itself (`python -m torch.distributed.run --nproc-per-node N bench.py ...`) as
    child processes BEFORE anything touches the GPU and relays rank 0's line and the exit code; launched by
    torch.distributed.run it is one of those ranks.  A WORLD_SIZE that differs from --gpus is an error, never a silent
    one-GPU measurement.)

One "step" = one full pass of the IA-SSD set-abstraction stack L0-L2 (D-FPS 16 384->4 096,
D-FPS ->1 024, ctr-aware top-k ->512; per layer two ball-query radii, grouping, grouped MLP,
max-pool, aggregation and confidence heads; tools/cfgs/kitti_models/IA-SSD.yaml:35-55) over one
batch of 8 synthetic KITTI-shaped scenes per GPU (BASELINE.json configs[1]).  Inputs are resident
in HBM before the timed region.  Scenes shard over ranks (weak scaling, DESIGN.md "multi-GPU").
At N > 1 the step is BASELINE configs[2]: it ends with the path's one exchange, the packed RCCL
all-gather of every layer's sampled indices (int32 (8, 4096 + 1024 + 512) per rank),
INSIDE the timed region; `ms_per_step_no_exchange` is the same K steps without it.

The headline `value` is the reference's arithmetic: every grouped MLP in strict fp32 (fp32 MFMA).  The
split-fp16 form of the wide scales (hi + lo halves, ~22-bit products, <= 2e-5 relative) is timed the
same way and reported beside it as `value_fp16x2`.

`--config 4` selects BASELINE configs[3] (stability top-k at layer 2), `--config 5` the per-GPU share of
configs[4] (1 scene x 180 000 points -> 16 384 / 4 096 / 1 024, nsample 64, fp16 features on MFMA).

Prints ONE JSON line (rank 0) with value = total points/s over all ranks, plus
  validated    -- what was checked about the timed work after the timed region (no progress-wait timeout, no
                  split-fp16 overflow, last step's outputs bit-identical to one plain sequential pass);
  value_fp16x2 -- the same K steps with the wide grouped-MLP scales as split-fp16 pairs, timed the same way;
  ms_per_step_median -- median of the K per-step HIP-event times (ms_per_step is elapsed / K);
  roofline     -- the dominant kernel (layer-0 FPS) on ALGORITHMIC touched bytes (SURVEY.md 8d:
                  20*N*(m-1) B per scene) over its HIP-event time measured inside the timed region;
  roofline_mlp -- the largest grouped-MLP launch: algorithmic FLOP over its HIP-event time (measured after the timed
                  region on the launch's own arguments) and the MFMA-busy fraction from the committed PMC pass;
  roofline_ball_query -- layer 0's two-radius ball query (whole layer, one scan): pair tests / s and algorithmic bytes
                  12 (N + M) + 4 M (ns_a + ns_b) per scene over its HIP-event time, plus the chunked launches of the
                  streamed pass as the step runs them;
  roofline_gather -- the gather / group side of the fused grouped-MLP launches: bytes gathered + written per launch
                  over its HIP-event time, per scale;
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
# issue-rate ceiling of the ball query's pair test (see ball_query_roofline): 1024 SIMDs x 2.4 GHz x 64 lanes / 23.1 cycles
VALU_PAIR_TEST_PEAK = 1024 * 2.4e9 * 64 / (4 * 2.31 + 2 * 2.72 + 2 * 4.20)
MFMA_PEAK_TF = {"fp32": 157.3, "fp16x2": 2500.0, "fp16": 2500.0}  # dense peaks: fp32 MFMA, fp16 MFMA (no sparsity)
PROFILE_DIRS = ("round5", "round4", "round3", "round2", "round1")


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
    ap.add_argument("--no-fp16x2-leg", "--no-fp32-leg", dest="no_second_leg", action="store_true",
                    help="skip the repetition in the other grouped-MLP arithmetic (value_fp16x2; value_fp32 when "
                         "--mlp-precision fp16x2 is the headline)")
    ap.add_argument("--no-exchange", action="store_true",
                    help="N > 1: leave the all-gather of the sampled indices out of the step (config 2 x N instead of configs[2])")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N = 1 only: initialise RCCL with a group of ONE rank and end every step with the same packed all-gather "
                         "(what the RCCL call itself costs a step; no xGMI is involved and the line says so)")
    ap.add_argument("--no-validate", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pipelined", action="store_true",
                    help="also time the same passes with two batches in flight (informational object `pipelined`; on by default "
                         "at N = 1 except for --config 5, where this flag adds it)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the informational `pipelined` leg")
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
    args.mlp_precision = args.mlp_precision or "fp32"     # the reference's arithmetic is the headline
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
        # what the launch really ISSUES to the matrix cores: the packed tile stream's columns (16 per tile, padding of the
        # power-of-two slots included) instead of the padded nsample columns per centroid, layer 1 reduced to its coordinate
        # k-step where its feature product is formed once per point, widths as padded for the MFMA tiles
        cols_alg = int(idx.shape[0] * (idx.shape[1] if call[8] is None else call[8]) * idx.shape[2])
        cols_issued = int(kw["columns"].ntiles.item()) * 16 if kw.get("columns") is not None else cols_alg
        k1 = 4 if kw.get("hoisted") is not None else (packed.cin + 3) // 4 * 4
        c3p = (packed.c3_real + 15) // 16 * 16
        flop_issued = 2.0 * cols_issued * (k1 * packed.c1 + packed.c1 * packed.c2 + packed.c2 * c3p)
        return {"flop": flop, "flop_issued": flop_issued, "columns_issued": cols_issued, "layer1_per_point": kw.get("hoisted") is not None,
                "ms": float(np.median(ms)), "widths": (packed.cin, packed.c1, packed.c2, packed.c3_real),
                "packed_columns": kw.get("columns") is not None,
                "nsample": int(idx.shape[2]), "columns": int(flop / (2.0 * (packed.cin * packed.c1 + packed.c1 * packed.c2
                                                                           + packed.c2 * packed.c3_real)))}


def _event_times(fn, reps, warm=3):
    """Median / mean HIP-event time (ms) of fn() on the stream it launches on (torch's current stream)."""
    for _ in range(warm):
        fn()
    pairs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        pairs.append((s, e))
    torch.cuda.synchronize()
    ms = [s.elapsed_time(e) for s, e in pairs]
    return float(np.median(ms)), float(np.mean(ms))


class ChunkProbe:
    """HIP events around every chunked two-radius ball-query launch of the streamed first layer (the consumer stream, behind
    the chunk's progress wait) -- switched on for a few probe steps AFTER the timed region, never inside it."""

    def __init__(self, ext):
        self.ext, self.orig, self.on, self.pairs = ext, ext.ball_query_full2_range, False, []
        ext.ball_query_full2_range = self

    def __call__(self, ra, rb, xyz, new_xyz, idx_a, idx_b, j0, jcount, **kw):
        if not self.on:
            return self.orig(ra, rb, xyz, new_xyz, idx_a, idx_b, j0, jcount, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = self.orig(ra, rb, xyz, new_xyz, idx_a, idx_b, j0, jcount, **kw)
        e.record()
        self.pairs.append((s, e, xyz.shape[0] * jcount * xyz.shape[1]))
        return r

    def summary(self, steps):
        if not self.pairs or steps <= 0:
            return None
        ms = sum(s.elapsed_time(e) for s, e, _ in self.pairs) / steps
        tests = sum(t for _, _, t in self.pairs) / steps
        n = len(self.pairs) // steps
        self.pairs = []
        return {"launches_per_step": n, "ms_per_step": ms, "pair_tests_per_step": tests, "pair_tests_per_s": tests / (ms * 1e-3)}


def ball_query_roofline(ext, layers, xyz, outs, chunk_stats):
    """Layer 0's two-radius ball query (`12 (N + M) + 4 M ns` compulsory bytes per scene and radius -- one scan
    serves both radii here, so the coordinates are counted once; `M N` pair tests per scene)."""
    ga, gb = layers[0].groupers
    new_xyz = outs[0][0]
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    keep = ext.BQ_GRID_MIN
    try:
        ext.BQ_GRID_MIN = None                     # the lane-per-centroid SCAN kernel: every one of the M N pairs is tested
        scan_ms, _ = _event_times(lambda: ext.ball_query_full2(ga.radius, ga.nsample, gb.radius, gb.nsample, xyz, new_xyz), 20)
        ext.BQ_GRID_MIN = (0, 0)                   # the cell-grid kernel chain: same rows from far fewer tests
        grid_ms, _ = _event_times(lambda: ext.ball_query_full2(ga.radius, ga.nsample, gb.radius, gb.nsample, xyz, new_xyz), 20)
    finally:
        ext.BQ_GRID_MIN = keep
    algo = float(B) * (12.0 * (N + M) + 4.0 * M * (ga.nsample + gb.nsample))
    tests = float(B) * M * N
    ach = algo / (scan_ms * 1e-3) / 1e9
    traffic = None
    data, where = _profile_json("pmc_traffic.json")
    rec = (data or {}).get("ball_query_dual_kernel")
    if rec and (rec["batch"], rec["n"], rec["m"]) == (B, N, M):
        traffic = (rec["fetch_kb"] + rec["write_kb"]) * 1024.0
    rate = tests / (scan_ms * 1e-3)
    out = {"bound": "valu", "kernel": f"ball_query_dual_kernel (layer 0, both radii in one scan: {M} centroids x {N} points, "
                                      f"r {ga.radius}/{gb.radius}, nsample {ga.nsample}/{gb.nsample})",
           "pair_tests_per_s": rate, "pair_tests": tests, "valu_issue_frac": rate / VALU_PAIR_TEST_PEAK,
           "valu_issue_peak_pair_tests_per_s": VALU_PAIR_TEST_PEAK, "launch_ms": scan_ms,
           "hbm": {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                   "algorithmic_bytes": algo},
           "grid_variant_ms": grid_ms,
           "note": "the kernel is VALU-bound on its M N pair tests, not on HBM (the cloud is re-read from the scalar cache / "
                   "L2; the `hbm` object prices the compulsory 12 (N + M) + 4 M (ns_a + ns_b) bytes per scene and is small by "
                   "construction).  valu_issue_frac = pair tests per second / the rate at which the chip can ISSUE the "
                   "eight VALU instructions of one test for 64 pairs (3 v_sub + v_mul at 2.31, 2 v_fma at 2.72, 2 v_cmp at "
                   "4.20 cycles per instruction and SIMD with >= 2 waves per SIMD: tools/microbench_valu*.hip, "
                   "profiles/round1/microbench_valu*_gfx950.txt; 23.1 cycles per 64 tests x 1024 SIMDs at 2.4 GHz = 6.8 T "
                   "tests/s) -- hit-list bookkeeping and the prefix sums come on top; grid_variant_ms = the same rows "
                   "through the cell-grid kernels (ball_query_grid.hip)"}
    if chunk_stats is not None:
        out["streamed_chunks"] = dict(chunk_stats, kernel="ball_query_wave_multi_kernel / ball_query_wave_seg_kernel "
                                      "(the step's chunked launches, each gathering its own centroids)")
    return out


def gather_roofline(fused, layers, xyz, feats, outs):
    """The gather / group side of the fused grouped-MLP launches (group_points + cat + centring in the reference,
    group_points_gpu.cu:14-50): per scale, bytes a launch gathers (4 (3 + C) per grouped column + its 4-byte index) and
    writes (4 C_out per centroid), over the HIP-event time of the whole-layer launch on the pass's own tensors."""
    from spsnet_amd import pointnet2_batch_cuda as ext
    rows = []
    src_xyz, src_f = xyz, feats
    for k, layer in enumerate(layers):
        new_xyz, new_f = outs[k][0], outs[k][1]
        plan = layer._fused_plan(src_xyz, new_xyz, src_f)
        if not plan or len(layer.groupers) != 2:
            src_xyz, src_f = new_xyz, new_f
            continue
        ga, gb = layer.groupers
        ia, ib = ext.ball_query_full2(ga.radius, ga.nsample, gb.radius, gb.nsample, src_xyz, new_xyz)
        B, M = new_xyz.shape[0], new_xyz.shape[1]
        out = torch.zeros((B, sum(p.c3_real for p in plan), M), dtype=torch.float32, device=xyz.device)
        off = 0
        for ix, packed in zip((ia, ib), plan):
            ms, _ = _event_times(lambda ix=ix, packed=packed, off=off:
                                 fused.group_mlp_pool(src_xyz, new_xyz, src_f, ix, packed, out, off), 10)
            cols = float(ix.numel())
            cin = 3 + (0 if src_f is None else src_f.shape[1])
            esz = 2.0 if (src_f is not None and src_f.dtype == torch.float16) else 4.0
            gathered = cols * (12.0 + esz * (cin - 3) + 4.0)
            written = 4.0 * B * M * packed.c3_real
            rows.append({"layer": k, "widths": [packed.cin, packed.c1, packed.c2, packed.c3_real], "nsample": int(ix.shape[2]),
                         "columns": int(cols), "gathered_bytes": gathered, "written_bytes": written, "launch_ms": ms,
                         "GB_per_s": (gathered + written) / (ms * 1e-3) / 1e9})
            off += packed.c3_real
        src_xyz, src_f = new_xyz, new_f
    if not rows:
        return None
    top = max(rows, key=lambda r: r["gathered_bytes"] + r["written_bytes"])
    ach = top["GB_per_s"]
    return {"bound": "hbm", "kernel": f"gather inside the fused grouped MLP, layer {top['layer']} scale "
                                      f"{'->'.join(str(w) for w in top['widths'])}, nsample {top['nsample']}",
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
            "launch_ms": top["launch_ms"], "per_scale": rows,
            "note": "bytes = what the reference's group_points + cat would move for the same columns (gathered inputs + "
                    "pooled output; its (B, C, M, ns) intermediates never exist here); the launch time is the WHOLE fused "
                    "kernel (gather + three layers on MFMA + pool), so this is a lower bound on the gather's own rate -- "
                    "wide scales are MFMA-bound, layer 0's are gather/VALU-bound"}


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


def oracle_verdict(got, want, samplers, tol):
    """The timed batch against the CPU oracle's pass over the SAME scenes and weights (run by cpu_baseline, outside every timed
    region): `got` / `want` = per layer (new_xyz, new_features, cls, sampled_idx) as numpy.  D-FPS layers: indices and
    centroids bit-exact, features / class scores within `tol` of the layer's largest reference magnitude (>= 1).  A
    score-sampled layer sits behind features that carry that tolerance, so its picks may swap near-ties: rows are matched by
    sampled index and compared on the shared picks.  -> (record, list of failures)"""
    rec = {"scenes": int(want[0][3].shape[0]), "tolerance": tol, "idx_exact": True, "new_xyz_exact": True,
           "max_abs_err_features": 0.0, "max_rel_err": 0.0, "layers": []}
    bad = []
    for k, ((gx, gf, gc, gi), (wx, wf, wc, wi)) in enumerate(zip(got, want)):
        row = {"layer": k, "sampler": samplers[k], "M": int(wi.shape[1])}
        scale = max(1.0, float(np.abs(wf).max()))
        if samplers[k] == "D-FPS" or np.array_equal(gi, wi):
            row["idx_exact"] = bool(np.array_equal(gi, wi))
            row["new_xyz_exact"] = bool(np.array_equal(gx, wx))
            err = float(np.abs(gf - wf).max()) if row["idx_exact"] else float("nan")
            cerr = float(np.abs(gc - wc).max()) if (row["idx_exact"] and wc is not None) else None
            if not row["idx_exact"]:
                rec["idx_exact"] = False
                bad.append(f"layer {k}: sampled indices differ from the oracle's")
            if not row["new_xyz_exact"]:
                rec["new_xyz_exact"] = False
                bad.append(f"layer {k}: centroids differ from the oracle's")
        else:
            err, cerr, shared = 0.0, (0.0 if wc is not None else None), []
            for b in range(wi.shape[0]):
                common, gpos, wpos = np.intersect1d(gi[b], wi[b], return_indices=True)
                shared.append(len(common) / wi.shape[1])
                if not np.array_equal(gx[b][gpos], wx[b][wpos]):
                    bad.append(f"layer {k} scene {b}: centroids of shared picks differ")
                err = max(err, float(np.abs(gf[b][:, gpos] - wf[b][:, wpos]).max()))
                if wc is not None:
                    cerr = max(cerr, float(np.abs(gc[b][gpos] - wc[b][wpos]).max()))
            row["picks_shared_with_oracle"] = float(np.mean(shared))
            row["compared"] = "rows matched by sampled index (score sampler behind 1e-4 features: near-ties may swap)"
            if row["picks_shared_with_oracle"] < 0.98:
                bad.append(f"layer {k}: only {row['picks_shared_with_oracle']:.3f} of the picks shared with the oracle")
        row.update(max_abs_err_features=err, feature_scale=scale, max_rel_err=err / scale)
        if cerr is not None:
            row["max_abs_err_cls"] = cerr
            if not cerr <= tol * max(1.0, float(np.abs(wc).max())):
                bad.append(f"layer {k}: class scores off by {cerr}")
        if not err <= tol * scale:
            bad.append(f"layer {k}: features off by {err} (scale {scale})")
        if err == err:
            rec["max_abs_err_features"] = max(rec["max_abs_err_features"], err)
            rec["max_rel_err"] = max(rec["max_rel_err"], err / scale)
        rec["layers"].append(row)
    rec["ok"] = not bad
    return rec, bad


def cpu_baseline(layers, args, gpu_outs=None, samplers=None):
    """Oracle port of the same stack on the host cores, on a bounded sample: once with all cores, once single-threaded
    (SURVEY.md 8d asks for both).  The all-core sample IS rank 0's timed batch (same scenes: seeds 0..B-1, same weights), so
    its first pass doubles as the checker of the GPU outputs the timed region produced (`gpu_outs`, snapshotted to the
    host right behind the timed steps) -> (baseline object, oracle verdict or None)."""
    from oracle import cpu_stack, oracle as O
    from spsnet_amd import scenes
    cores = os.cpu_count() or 1
    cpu_layers = cpu_stack.cpu_copy(layers)
    first = {}

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
            res = cpu_stack.sa_stack_cpu(cpu_layers, xyz, feats, stds)
            if reps == 0:
                first[nsc] = res
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
    verdict = None
    if gpu_outs is not None and nsc == args.batch:
        verdict, bad = oracle_verdict(gpu_outs, first[nsc], samplers, 2e-3 if args.mlp_precision == "fp16" else 1e-4)
        verdict["checker"] = ("oracle/cpu_stack.sa_stack_cpu over oracle/sa_oracle.c on the timed batch's own scenes and weights, "
                              "run after the timed region (its all-core pass is also cpu_baseline's sample)")
        if bad:
            raise SystemExit("bench.py: the timed batch does not match the CPU oracle: " + "; ".join(bad) + f" {verdict}")
    return out, verdict


def training_step_leg(modules_pkg, sa_stack, cfg, args, xyz, feats, dev, reps=10):
    """Informational, never `value`: one TRAINING step (forward + backward, BatchNorm on batch statistics) of the same SA
    layers on the same batch -- SURVEY 8 f-1.  Two timings: the step as a plain loop runs it, and with the next batch's
    layer-0 sampling started beside the backward (sa_stack.prefetch_first_layer; here the next batch is the same tensor)."""
    layers = sa_stack.build_sa_layers(modules_pkg, cfg, seed=0).to(dev).train()
    f = feats.float()
    from spsnet_amd import fused

    def step(prefetch):
        for p in layers.parameters():
            p.grad = None
        outs = sa_stack.run_sa_layers(layers, xyz, f)
        loss = sum(o[1].square().mean() for o in outs) + sum(o[2].square().mean() for o in outs if o[2] is not None)
        if prefetch:
            sa_stack.prefetch_first_layer(layers, xyz)
        loss.backward()

    res = {}
    keep = fused.TRAIN_PRECISION
    # the library default first (exact fp32 = the reference's arithmetic: fused train-mode kernels on v_mfma_f32_16x16x4_f32), then
    # the opt-in split-fp16 form (exact power-of-two operand scaling: gradients within 1-3e-6 of float64)
    for prec, suffix in (("fp32", ""), ("fp16x2", "_fp16x2")):
        fused.set_train_precision(prec)
        for key, pre in (("ms", False), ("ms_next_batch_sampling_prefetched", True)):
            layers[0]._presampled = layers[0]._preball = None
            for _ in range(3):
                step(pre)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                step(pre)
            torch.cuda.synchronize()
            res[key + suffix] = 1e3 * (time.perf_counter() - t0) / reps
    layers[0]._presampled = layers[0]._preball = None
    fused.set_train_precision(keep)
    res["points_per_s"] = xyz.shape[0] * xyz.shape[1] / (res["ms"] * 1e-3)
    res["grouped_mlp"] = ("fused train-mode kernels (csrc/mlp_train.hip): `ms` in exact fp32 MFMA (the default, the reference's "
                          "arithmetic), `ms_fp16x2` with split-fp16 operands (opt-in)"
                          if modules_pkg.FUSED_MLP_TRAINING else "op-by-op fp32 kernels")
    res["note"] = "forward + backward of SA layers 0-2 in train() mode on the bench batch; informational, not the headline metric"
    return res


def backbone_forward_leg(args, dev, reps=20):
    """Informational, never `value`: the whole IASSD_Backbone forward (SA layers 0-3, the vote layer and layer 5 with its
    256 / 512 / 1024-wide scales: IASSD_backbone.py:93-168, IA-SSD.yaml:35-55) on a batch of the bench shape, in both grouped-MLP
    arithmetics.  Strict fp32 runs layer 5 on the point-major fp32 MFMA kernel -- no library GEMM on the inference path."""
    from spsnet_amd import backbones as BB, fused, scenes
    xyz, feats = scenes.make_batch(args.dataset, args.batch, args.points, seed0=1)
    bidx = np.repeat(np.arange(args.batch, dtype=np.float32), args.points)[:, None]
    points = torch.from_numpy(np.concatenate([bidx, xyz.reshape(-1, 3), feats.transpose(0, 2, 1).reshape(-1, 1)], 1)
                              .astype(np.float32)).to(dev)
    net = scenes.fill_parameters(BB.IASSD_Backbone(BB.IASSD_KITTI_CFG, input_channels=4, num_class=3), 5).to(dev).eval()
    res = {}
    keep = fused.PRECISION
    try:
        for prec in ("fp32", "fp16x2"):
            fused.set_precision(prec)
            with torch.no_grad():
                for _ in range(4):
                    net(dict(batch_size=args.batch, points=points))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    net(dict(batch_size=args.batch, points=points))
                torch.cuda.synchronize()
            res[f"backbone_forward_ms_{prec}"] = 1e3 * (time.perf_counter() - t0) / reps
    finally:
        fused.set_precision(keep)
    res["note"] = (f"IASSD_Backbone.forward, {args.batch} x {args.points} points, wall clock over {reps} back-to-back forwards; "
                   "informational, not the headline metric")
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


def launch_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as children (torch.distributed.run, one per GPU,
    rendezvous on 127.0.0.1) -- the parent never initialises the GPU -- relay their stdout (rank 0's ONE line) and stderr,
    and exit with the launcher's code.  Mirrors the reference's one command per node (tools/scripts/dist_train.sh:1-20 ->
    tools/train.py:64-72 `init_dist_pytorch`)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: the only form this host's driver supports
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: starting", args.gpus, "ranks:", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    if int(env_world or "1") != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: the line would report {env_world} GPU(s); "
                         "launch with --nproc-per-node equal to --gpus (or run `python bench.py --gpus N`, which starts its "
                         "own ranks)")
    # stdout carries the ONE JSON line and nothing else: whatever a library prints there (RCCL's version banner at
    # communicator set-up, for one) goes to stderr for the rest of the run
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
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
    if args.force_exchange:
        if world != 1:
            raise SystemExit("--force-exchange is the one-rank exercise of the exchange; at N > 1 the exchange is in the step anyway")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
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
    chunk_probe = ChunkProbe(ext)
    # BASELINE configs[2]: at N > 1 every step ends with the ONE exchange the sharded path has -- the global-batch view of each
    # layer's sampled indices, all three layers packed into one all-gather (RCCL over xGMI; latency-bound: 22 KiB per rank)
    exchange = (world > 1 or args.force_exchange) and not args.no_exchange
    state = {"exchange": exchange, "gathered": None}

    def step(**kw):
        with torch.no_grad():
            outs = sa_stack.run_sa_layers(layers, xyz, feats, stds, stream_first_layer=args.stream_first_layer, **kw)
            if state["exchange"]:
                state["gathered"] = all_gather_sampled_idx([o[3] for o in outs], always_collective=args.force_exchange)
            return outs

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
    # the pass's overlap (FPS producer beside its consumers) is only as good as the placement of its helper streams on
    # hardware queues: every rank reports whether all of them were SHOWN to run beside the pass (streams.py), AND-reduced
    from spsnet_amd import streams as _streams
    overlap_ok = bool(_streams.overlap_verified(dev, torch.cuda.current_stream(dev))) if args.stream_first_layer else None
    if overlap_ok is not None and world > 1:
        flag = torch.tensor([1 if overlap_ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        overlap_ok = bool(int(flag.item()))
    checked = validate(outs)
    # the timed batch's outputs, kept on the host for the oracle check that rides on the CPU-baseline leg (N = 1, rank 0)
    outs_host = None
    if world == 1 and not args.no_cpu_baseline and not args.no_validate:
        outs_host = [tuple(None if t is None else t.detach().float().cpu().numpy() if t.is_floating_point()
                           else t.detach().cpu().numpy() for t in o[:4]) for o in outs]
    no_exchange = None
    if exchange:
        g = state["gathered"]
        mine = [o[3] for o in outs]
        assert all(t.shape[0] == world * args.batch for t in g)
        assert all(torch.equal(t[rank * args.batch:(rank + 1) * args.batch], m.to(torch.int32)) for t, m in zip(g, mine)), \
            "the gathered sampled indices do not hold this rank's rows at its offset"
        state["exchange"] = False        # the same K steps without the exchange: what the all-gather costs a step
        el_ne, per_ne, _ = timed(args.steps, max(2, args.warmup // 2))
        probe.summary()
        no_exchange = (el_ne, per_ne)
        state["exchange"] = True
    mlp = mlp_probe.measure() if world == 1 or rank == 0 else None
    bq_line = gather_line = None
    if rank == 0 and args.config != 5:
        # per-kernel lines for the gather side (after the timed region; the chunked ball queries over a few probed steps)
        chunk_probe.on, keep_x = True, state["exchange"]
        state["exchange"] = False
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        chunk_probe.on, state["exchange"] = False, keep_x
        with torch.no_grad():
            bq_line = ball_query_roofline(ext, layers, xyz, outs, chunk_probe.summary(5))
            gather_line = gather_roofline(fused, layers, xyz, feats, outs)

    # the same K steps in the OTHER grouped-MLP arithmetic, timed the same way: split-fp16 pairs for the wide scales when the
    # headline is strict fp32 (the default), strict fp32 when --mlp-precision fp16x2 made the split form the headline
    other = {"fp32": "fp16x2", "fp16x2": "fp32"}.get(args.mlp_precision)
    second_leg = None
    if not args.no_second_leg and other is not None and args.config != 5:
        fused.set_precision(other)
        mlp_probe.best = None            # (the costliest launch of THIS leg: same shape, weights packed for its kernel)
        el2, per2, outs2 = timed(args.steps, max(args.warmup, 2))
        probe.summary()
        chk2 = validate(outs2)
        mlp2 = mlp_probe.measure() if world == 1 or rank == 0 else None
        second_leg = (el2, per2, chk2, mlp2)
        fused.set_precision(args.mlp_precision)

    # Extra, reported separately (never `value`): the same K complete passes with TWO batches in flight on two
    # streams -- layer-0 FPS keeps one CU per scene busy for most of a pass, so a second pass fits beside it.
    pipelined = None
    # (on by default at N = 1 since round 5 for the KITTI-sized configs; informational: an error in it never costs the line)
    if world == 1 and not args.no_pipelined and not exchange and (args.pipelined or args.config != 5):
        try:
            pipelined = sa_stack.pipelined_bench(step, args.steps, dev, in_flight=args.in_flight, scenes=args.batch,
                                                 fenced=not args.no_cu_fence)
            pipelined["value"] = args.batch * args.points * args.steps / pipelined.pop("elapsed_s")
            same, where = same_outputs(pipelined.pop("last_outputs"), outs)
            if not same or sa_stack.check_timeouts():
                pipelined = {"error": f"the pipelined leg's outputs differ from the sequential steps' ({where}) or a bounded wait gave up"}
            else:
                pipelined["validated"] = "last pass bit-identical to the sequential steps' outputs, no progress-wait timeout"
        except Exception as exc:   # noqa: BLE001
            pipelined = {"error": f"{type(exc).__name__}: {exc}"}

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
            "config": {"workload": (f"BASELINE configs[2]: batch={world * args.batch} sharded {world}-way ({args.batch}/GPU) x "
                                    f"{args.points} pts" if exchange and world > 1 else
                                    f"BASELINE configs[{args.config - 1}]: batch={args.batch}/GPU x {args.points} pts") +
                                   f" ({args.dataset}), IA-SSD SA L0-L2 ({shape_txt}, layer-2 sampler {args.sampler}), "
                                   f"grouped MLP {args.mlp_precision}" +
                                   (", one packed RCCL all-gather of the sampled indices per step" if exchange else "") +
                                   (" (--force-exchange: a group of ONE rank)" if args.force_exchange else ""),
                       "global_batch": world * args.batch, "points_per_scene": args.points,
                       "parallelism": (f"scene-sharded x{world}; per step one all-gather of the layers' sampled indices "
                                       f"(int32 ({args.batch}, sum M) per rank) inside the timed region" if exchange else
                                       f"scene-sharded x{world}, no data-path collective")},
        }
        if no_exchange is not None:
            line["ms_per_step_no_exchange"] = 1e3 * no_exchange[0] / args.steps
            line["ms_per_step_no_exchange_median"] = float(statistics.median(no_exchange[1]))
            line["exchange"] = {"collective": "all_gather_into_tensor (RCCL)" + (", group of ONE rank: the cost of the call, no xGMI"
                                                                               if args.force_exchange else ""),
                                "ranks": world, "per_step": 1,
                                "bytes_per_rank": int(4 * args.batch * sum(o[3].shape[1] for o in outs)),
                                "checked": "every rank's rows found at its offset of the gathered tensors"}
        if rehearsal:
            line["data"] = "synthetic; REHEARSAL: all ranks share one GPU over gloo -- not a measurement"
        if checked is not None:
            line["validated"] = checked
        # helper streams are probed for real concurrency with the pass's streams
        line["helper_streams"] = dict(_streams.stats, note="device-side probes at stream set-up; rejected = candidates that shared "
                                      "a hardware queue with the pass (HIP multiplexes streams onto GPU_MAX_HW_QUEUES queues); "
                                      "unplaced = roles that got NO queue of their own: the pass then serialises")
        if overlap_ok is not None:
            line["overlap_verified"] = overlap_ok       # all ranks: every helper stream shown to run beside the pass
            base_ms = (1e3 * no_exchange[0] / args.steps) if no_exchange is not None else 1e3 * elapsed / args.steps
            if fps is not None and (not overlap_ok or base_ms > fps[0] + 0.8):
                line["overlap_warning"] = (f"a step without the exchange takes {base_ms:.3f} ms against {fps[0]:.3f} ms for its FPS "
                                           "launch alone (+0.8 ms allowed for the tail behind the last pick)" +
                                           ("" if overlap_ok else "; helper streams unplaced: " + ", ".join(_streams.unplaced())) +
                                           " -- the FPS producer and its consumers are probably sharing a hardware queue "
                                           "(GPU_MAX_HW_QUEUES)")
        if second_leg is not None:
            el2, per2, chk2, mlp2 = second_leg
            line[f"value_{other}"] = total_points / el2
            line[f"ms_per_step_{other}"] = 1e3 * el2 / args.steps
            line[f"ms_per_step_{other}_median"] = float(statistics.median(per2))
            line[f"dtype_{other}"] = {"fp32": "f32", "fp16x2": "f32 results from split-fp16 operands (hi + lo halves, 3 MFMAs per "
                                      "product block, fp32 accumulate; <= 2e-5 relative to fp32) on the wide grouped-MLP scales"}[other]
            if chk2 is not None:
                line[f"validated_{other}"] = chk2
        if args.mlp_precision == "fp32":
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
            ach_alg = mlp["flop"] / (mlp["ms"] * 1e-3) / 1e12
            issue = 3.0 if precision == "fp16x2" else 1.0      # split-fp16: three fp16 MFMAs per product block
            ach_tf = issue * mlp["flop_issued"] / (mlp["ms"] * 1e-3) / 1e12
            busy, src = pmc_mfma_busy(mlp["widths"], mlp["nsample"], precision)
            return {
                "bound": "mfma", "kernel": f"grouped MLP {mlp['widths'][0]}->{mlp['widths'][1]}->{mlp['widths'][2]}->"
                                           f"{mlp['widths'][3]}, nsample {mlp['nsample']}, {mlp['columns']} columns",
                "achieved": ach_tf, "peak": peak, "unit": "TFLOP/s", "frac": ach_tf / peak, "launch_ms": mlp["ms"],
                "flop_issued": issue * mlp["flop_issued"], "columns_issued": mlp["columns_issued"],
                "layer1_per_point": mlp["layer1_per_point"],
                "achieved_algorithmic": ach_alg, "frac_algorithmic": ach_alg / peak, "flop": mlp["flop"],
                "packed_columns": mlp["packed_columns"],
                "mfma_busy_frac": None if busy is None else busy.get("mfma_busy_frac"),
                "mfma_busy_source": src,
                "note": "frac = UTILISATION: the MFMA flop the launch issues (packed tile stream x 16 columns, layer 1 without "
                        "its hoisted per-point feature product, MFMA-tile widths; x3 for split-fp16) / HIP-event time of "
                        "that launch repeated on its own arguments after the timed region / dense MFMA peak of the operand "
                        "type; frac_algorithmic = the reference's flop 2*columns*sum(Cin*Cout) (padded duplicate columns "
                        "included, as the reference computes them) over the same time -- a speed-up figure, not a "
                        "utilisation; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 1024 SIMDs x 2.4 GHz) "
                        "from the committed PMC pass (the counter's view of the same thing)"}
        if mlp is not None:
            line["roofline_mlp"] = mlp_roofline(mlp, args.mlp_precision)
        if second_leg is not None and second_leg[3] is not None:   # the other leg's dominant launch on its own pipe
            line[f"roofline_mlp_{other}"] = mlp_roofline(second_leg[3], other)
        if bq_line is not None:
            line["roofline_ball_query"] = bq_line
        if gather_line is not None:
            line["roofline_gather"] = gather_line
        if pipelined is not None:
            line["pipelined"] = pipelined
        if world == 1 and not args.no_training_leg and args.config != 5 and args.mlp_precision != "fp16":
            try:
                line["training_step"] = training_step_leg(M, sa_stack, cfg, args, xyz, feats, dev)
            except Exception as exc:   # informational only: never lose the headline line over it
                line["training_step"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_training_leg and args.config not in (4, 5) and args.mlp_precision != "fp16":
            try:
                line["backbone_forward"] = backbone_forward_leg(args, dev)
            except Exception as exc:   # informational only
                line["backbone_forward"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], verdict = cpu_baseline(layers, args, outs_host, [m[0] for m in cfg["sample_method_list"]])
            if verdict is not None and "validated" in line:
                line["validated"]["oracle"] = verdict
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
