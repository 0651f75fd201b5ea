/*
 * spsnet_sa.h -- C ABI of libspsnet_sa.so, the MI355X (gfx950) implementation of the
 * point-sampling / set-abstraction hot path of SPSNet / IA-SSD.
 *
 * This is the drop-in boundary.  Every entry point replaces one `*_kernel_launcher*`
 * of the reference's `pointnet2_batch_cuda` extension (all paths relative to
 * pcdet/ops/pointnet2/pointnet2_batch/src/ of the reference) and keeps its argument
 * order and meaning, with two deliberate changes:
 *   - a trailing `sps_stream_t` (a hipStream_t; NULL = the legacy default stream the
 *     reference launches on, sampling_gpu.cu:223);
 *   - an `int` status (SPS_OK / SPS_ERR_*) instead of fprintf + exit(-1)
 *     (sampling_gpu.cu:39-43); `sps_last_error()` returns the message.
 *
 * Ownership is the reference's: the caller allocates every buffer, including scratch
 * (`temp` pre-filled with 1e10, ball-query `idx` zeroed, gradient buffers zeroed --
 * pointnet2_utils.py:25-26,95,218,246).  Nothing is allocated, freed or synchronised
 * inside the launchers, so they are legal inside hipGraph stream capture; the two set-up
 * calls that do allocate or synchronise say so: sps_init() (once per device, OPTIONAL:
 * it creates the flag pool of the FPS sorting pre-pass; without it the FPS launchers use
 * the kernel that sorts for itself) and sps_streams_run_concurrently() (a probe).  All
 * pointers are DEVICE pointers to contiguous fp32 / int32 arrays.
 */
#ifndef SPSNET_SA_H
#define SPSNET_SA_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void *sps_stream_t; /* hipStream_t */

enum {
    SPS_OK = 0,
    SPS_ERR_INVALID = 1, /* bad shape / null pointer / unsupported size */
    SPS_ERR_LAUNCH = 2   /* hipGetLastError() != hipSuccess after the launch */
};

/* library identity: SPS_ABI_VERSION this header was written for.  2 (round 5): sps_init / sps_is_initialized, the
 * sps_mlp_train_* descriptor calls, sps_set_train_precision and sps_struct_size were added since 1, and the FPS launchers stopped
 * creating the pre-pass's flag pool themselves (a caller that passes a workspace for 6144 <= n <= 16 384 and never called
 * sps_init() gets the kernel that sorts for itself: same picks, no pre-pass).
 * sps_struct_size: sizeof() of the descriptor structs below as THIS library was compiled (which: 0 = sps_mlp_train_desc;
 * -1 for anything else) -- a binding that mirrors a struct by hand checks its own size against it. */
#define SPS_ABI_VERSION 2
int sps_abi_version(void);
long long sps_struct_size(int which);
/* message of the most recent failing call on this thread ("" if none) */
const char *sps_last_error(void);
/* One-time set-up of the library's per-device state (today: the 256 KiB flag pool of the FPS sorting pre-pass); `device`
 * is made current for the call, `stream` is synchronised.  Idempotent, thread-safe, optional (see above).  Call it before a
 * stream capture that is meant to record the fast FPS path.  No reference counterpart: the reference's launchers
 * (sampling_gpu.cu:211-253) own no device state.  sps_is_initialized: 1 once it has succeeded for `device`. */
int sps_init(int device, sps_stream_t stream);
int sps_is_initialized(int device);
/* block size the reference would use for an n-point FPS (cuda_utils.h:10-14) */
int sps_opt_n_threads(int work_size);
/* FPS kernel selection: 0 = automatic (spatially pruned kernel where it applies), 1 = brute-force
 * register-resident / streaming kernels only.  Both give bit-identical results; returns the old mode. */
int sps_set_fps_mode(int mode);

/* Stream plumbing for the CU-fenced schedule (no counterpart in the reference, which launches everything on the legacy
 * default stream): a HIP stream restricted to the compute units set in mask[0 .. words) (hipExtStreamCreateWithCUMask);
 * the caller owns it.  A pass's FPS chain occupies one CU per scene for most of the pass; on a stream of its own CUs a
 * second pass's whole-chip kernels no longer land on those CUs. */
int sps_stream_create_cu_mask(int words, const unsigned *mask, sps_stream_t *stream);
int sps_stream_destroy(sps_stream_t stream);
/* Do kernels of streams a and b really run side by side?  HIP multiplexes streams onto a few hardware queues
 * (GPU_MAX_HW_QUEUES, 4 by default) and two streams that share one are serialised; which streams share depends on everything
 * in the process that created streams (initialising RCCL is enough).  A one-lane kernel on `a` waits, bounded by limit_us of
 * wall clock, for a word that a kernel launched on `b` right behind it sets: *concurrent = 1 if it saw the word.  scratch =
 * two device ints.  Synchronises both streams: a set-up call (spsnet_amd/streams.py picks its helper streams with it). */
int sps_streams_run_concurrently(sps_stream_t a, sps_stream_t b, int *scratch, int limit_us, int *concurrent);

/* ---- the 11 functions of pointnet2_batch_cuda (src/pointnet2_api.cpp:10-26) ---------- */

/* farthest_point_sampling_kernel_launcher, sampling_gpu.cu:211-253 / sampling_gpu.h:24-25.
 * dataset (B,N,3) f32, temp (B,N) f32 in/out (running min distance), idxs (B,M) i32 out. */
int sps_farthest_point_sampling_kernel_launcher(int b, int n, int m, const float *dataset,
                                                float *temp, int *idxs, sps_stream_t stream);

/* furthest_point_sampling_with_dist_kernel_launcher, sampling_gpu.cu:374-416 / sampling_gpu.h:30-31.
 * dataset (B,N,N) f32 pairwise distances. */
int sps_furthest_point_sampling_with_dist_kernel_launcher(int b, int n, int m, const float *dataset,
                                                          float *temp, int *idxs, sps_stream_t stream);

/* gather_points_kernel_launcher_fast, sampling_gpu.cu:26-43 / sampling_gpu.h:12-13.
 * points (B,C,N), idx (B,npoints) -> out (B,C,npoints). */
int sps_gather_points_kernel_launcher_fast(int b, int c, int n, int npoints, const float *points,
                                           const int *idx, float *out, sps_stream_t stream);

/* gather_points_grad_kernel_launcher_fast, sampling_gpu.cu:65-83 / sampling_gpu.h:19-20.
 * grad_points (B,C,N) must be zeroed by the caller; contributions are accumulated. */
int sps_gather_points_grad_kernel_launcher_fast(int b, int c, int n, int npoints, const float *grad_out,
                                                const int *idx, float *grad_points, sps_stream_t stream);

/* ball_query_kernel_launcher_fast, ball_query_gpu.cu:48-67 / ball_query_gpu.h:12-13.
 * new_xyz (B,M,3), xyz (B,N,3) -> idx (B,M,nsample) i32; rows of empty balls are not written. */
int sps_ball_query_kernel_launcher_fast(int b, int n, int m, float radius, int nsample,
                                        const float *new_xyz, const float *xyz, int *idx,
                                        sps_stream_t stream);

/* ball_query_dilated_kernel_launcher_fast, ball_query_gpu.cu:120-139 / ball_query_gpu.h:18-19. */
int sps_ball_query_dilated_kernel_launcher_fast(int b, int n, int m, float max_radius, float min_radius,
                                                int nsample, const float *new_xyz, const float *xyz,
                                                int *idx, sps_stream_t stream);

/* group_points_kernel_launcher_fast, group_points_gpu.cu:74-92 / group_points_gpu.h:13-14.
 * points (B,C,N), idx (B,npoints,nsample) -> out (B,C,npoints,nsample). */
int sps_group_points_kernel_launcher_fast(int b, int c, int n, int npoints, int nsample,
                                          const float *points, const int *idx, float *out,
                                          sps_stream_t stream);

/* group_points_grad_kernel_launcher_fast, group_points_gpu.cu:33-50 / group_points_gpu.h:19-20. */
int sps_group_points_grad_kernel_launcher_fast(int b, int c, int n, int npoints, int nsample,
                                               const float *grad_out, const int *idx, float *grad_points,
                                               sps_stream_t stream);

/* three_nn_kernel_launcher_fast, interpolate_gpu.cu:61-80.
 * unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3) f32 (squared), idx (B,n,3) i32. */
int sps_three_nn_kernel_launcher_fast(int b, int n, int m, const float *unknown, const float *known,
                                      float *dist2, int *idx, sps_stream_t stream);

/* three_interpolate_kernel_launcher_fast, interpolate_gpu.cu:107-123.
 * points (B,C,m), idx/weight (B,n,3) -> out (B,C,n). */
int sps_three_interpolate_kernel_launcher_fast(int b, int c, int m, int n, const float *points,
                                               const int *idx, const float *weight, float *out,
                                               sps_stream_t stream);

/* three_interpolate_grad_kernel_launcher_fast, interpolate_gpu.cu:152-168. */
int sps_three_interpolate_grad_kernel_launcher_fast(int b, int c, int n, int m, const float *grad_out,
                                                    const int *idx, const float *weight, float *grad_points,
                                                    sps_stream_t stream);

/* D-FPS of a cloud that is itself the output of a D-FPS, in pick order (layer k+1 of IA-SSD sampling layer k's
 * centroids, IASSD_backbone.py:128-134): the answer is 0..m-1 unless exact distance ties interfere, so it is
 * CHECKED in two parallel passes (csrc/fps_verify.hip) and only scenes that fail the check are recomputed by
 * the ordinary kernel.  Always bit-identical to sps_farthest_point_sampling_kernel_launcher on the same
 * input, for ANY input (an unordered cloud just fails the check and is recomputed).
 * Device workspace from the caller: work_T (B*m f32), work_temp (B*n f32), flags (B i32). */
int sps_fps_ordered_prefix(int b, int n, int m, const float *xyz, float *temp, int *idxs, float *work_T,
                           float *work_temp, int *flags, sps_stream_t stream);
/* The two passes of sps_fps_ordered_prefix as separate calls (m <= 2048).  _begin (flags <- 0, pass 1) reads only the
 * first m points of every scene, so a caller that receives the cloud piecewise can issue it as soon as those exist;
 * _finish (pass 2 + recomputation of flagged scenes) needs the whole cloud. */
int sps_fps_ordered_prefix_begin(int b, int n, int m, const float *xyz, const float *temp, float *work_T, int *flags,
                                 sps_stream_t stream);
int sps_fps_ordered_prefix_finish(int b, int n, int m, const float *xyz, float *temp, int *idxs, const float *work_T,
                                  float *work_temp, int *flags, const int *force_redo, sps_stream_t stream);
/* Pass 2 in pieces, for a caller that receives the cloud piecewise (sa_stack's streamed first layer): its points are
 * independent, so _check_range covers the points [k0, k0 + kcount) of every scene (k0 a multiple of 64) as soon as THEY exist
 * (and the first m: pass 1 must have run), and _finish_from checks what is left, [k_from, n) (k_from a multiple of 64 below
 * n), reads force_redo and resolves -- sps_fps_ordered_prefix_finish is _finish_from with k_from = 0. */
int sps_fps_ordered_prefix_check_range(int b, int n, int m, int k0, int kcount, const float *xyz, const float *temp, int *idxs,
                                       const float *work_T, float *work_temp, int *flags, sps_stream_t stream);
int sps_fps_ordered_prefix_finish_from(int b, int n, int m, int k_from, const float *xyz, float *temp, int *idxs,
                                       const float *work_T, float *work_temp, int *flags, const int *force_redo,
                                       sps_stream_t stream);

/* ---- fused entry points for the SA module layer (pointnet2_modules.py) ----------------- */

/* Score + top-k sampler: replaces the max/sigmoid/(mul)/topk/int chain of
 * pointnet2_modules.py:287-303.  cls (B,N,C) f32 logits; stds (B,N) f32 or NULL.
 *   stds == NULL : score = sigmoid(max_c cls)                         ('ctr_aware' / 'cls')
 *   stds != NULL : score = sigmoid(max_c cls) * (1 - sigmoid(stds/8-3))   (SPSNet 'ss'/'sss')
 * idx (B,K) i32 receives the K best points, score descending, index ascending on ties;
 * score_out (B,N) f32 (may be NULL) receives the scores.  Requires K <= N <= 16384. */
int sps_score_topk(int b, int n, int c, int k, const float *cls, const float *stds, int *idx,
                   float *score_out, sps_stream_t stream);
/* The same with the sampler's centroid gather fused (pointnet2_modules.py:423-424): new_xyz (b, k, 3) = xyz (b, n, 3)[idx];
 * xyz / new_xyz may be NULL.  Scenes of up to 2048 points are ranked by 16 x more workgroups instead of sorted by one. */
int sps_score_topk_gather(int b, int n, int c, int k, const float *cls, const float *stds, const float *xyz, int *idx,
                          float *new_xyz, float *score_out, sps_stream_t stream);

/* Fused QueryAndGroup (pointnet2_utils.py:299-322): ball query, grouping of xyz (centred
 * on new_xyz) and of `features`, concatenated along channels, in one pass.
 * xyz (B,N,3), new_xyz (B,M,3), features (B,C,N) or NULL (C = 0)
 *   -> idx (B,M,nsample) i32 (required; fully written, zeros for empty balls)
 *   -> out (B,3+C,M,nsample) f32 when use_xyz != 0, else (B,C,M,nsample). */
int sps_query_and_group(int b, int n, int m, int c, float radius, int nsample, int use_xyz,
                        const float *xyz, const float *new_xyz, const float *features, int *idx,
                        float *out, sps_stream_t stream);

/* The grouping half of sps_query_and_group for neighbour indices the caller already holds (pointnet2_utils.py:312-320:
 * grouping_operation x2, subtract, cat): idx (B,M,nsample) i32 -> out as above.  Indices must lie in [0, N). */
int sps_group_concat(int b, int n, int m, int c, int nsample, int use_xyz, const float *xyz, const float *new_xyz,
                     const float *features, const int *idx, float *out, sps_stream_t stream);

/* sps_group_points_grad_kernel_launcher_fast for a grad_out that is a channel slice of a wider tensor (the feature rows of
 * the (B, 3+C, M, nsample) gradient sps_group_concat's output receives: group_points_gpu.cu:53-71 behind the slice that
 * torch.cat's backward makes in pointnet2_utils.py:316-320): scene s's c rows start at grad_out + s * grad_batch_stride
 * floats (>= c * npoints * nsample).  grad_points (b, c, n) is accumulated into (zero it first), summation order unspecified. */
int sps_group_points_grad_strided(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                  long long grad_batch_stride, const int *idx, float *grad_points, sps_stream_t stream);

/* new_xyz (B,M,3) = rows idx (B,M) of xyz (B,N,3): the transpose + gather_points + transpose of
 * pointnet2_modules.py:261,423-424 without the two layout copies (same values, it is a pure copy). */
int sps_gather_xyz(int b, int n, int m, const float *xyz, const int *idx, float *out, sps_stream_t stream);

/* ball_query_kernel_launcher_fast in "write every row" mode: rows of empty balls are written as zeros, so
 * the caller needs no zero-fill (what pointnet2_utils.py:246 + ball_query_gpu.cu:9-45 produce together). */
int sps_ball_query_full(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                        const float *xyz, int *idx, sps_stream_t stream);

/* The two ball queries of a two-scale SA layer (same centroids, same points, two radii / nsample) in ONE scan:
 * identical to two sps_ball_query_full calls, the pair distance is evaluated once.  perm_work: optional device
 * scratch of B*M int32; when given, centroids are processed in spatially sorted groups of 64 (same result).
 * sps_ball_query_full2_wave is the one-wave-per-centroid variant (per-centroid early exit; same result). */
int sps_ball_query_full2(int b, int n, int m, float radius_a, int nsample_a, float radius_b, int nsample_b,
                         const float *new_xyz, const float *xyz, int *idx_a, int *idx_b, int *perm_work,
                         sps_stream_t stream);
int sps_ball_query_full2_wave(int b, int n, int m, float radius_a, int nsample_a, float radius_b, int nsample_b,
                              const float *new_xyz, const float *xyz, int *idx_a, int *idx_b, sps_stream_t stream);

/* Fused group -> shared MLP -> max-pool of ONE grouping scale, inference mode (BatchNorm folded):
 * replaces grouping_operation x2 + cat + [Conv2d 1x1, BatchNorm2d, ReLU] x3 + max_pool2d of
 * pointnet2_modules.py:429-447 by one MFMA kernel (csrc/sa_mlp.hip).
 *   xyz (B,N,3), new_xyz (B,M,3), features (B,c_feat,N) or NULL, idx (B,M,nsample) from the ball query
 *   c1, c2, c3: layer widths PADDED to multiples of 16; c3_real <= c3 channels are written
 *   w1/b1, w2/b2, w3/b3: folded weights in MFMA fragment order (spsnet_amd/fused.py) and padded biases
 *   out (B, out_c_total, M): channels [out_c_off, out_c_off + c3_real) receive the pooled features.
 * B*M*nsample must be a multiple of 32.  nsample 64: `out` must be zero-filled by the caller (a centroid's samples
 * span two kernel units whose maxima are combined with an atomic max; pooled values are >= 0 after the ReLU).  Returns SPS_ERR_INVALID when no kernel variant exists for
 * (c1, c2, nsample): ask sps_sa_group_mlp_supported first. */
int sps_sa_group_mlp(int b, int n, int m, int c_feat, int nsample, const float *xyz, const float *new_xyz,
                     const float *features, const int *idx, int c1, int c2, int c3, int c3_real,
                     const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                     const float *b3, float *out, int out_c_total, int out_c_off, sps_stream_t stream);
int sps_sa_group_mlp_supported(int c1, int c2, int nsample);
/* 1 if only the shared-stream split-fp16 kernel (sps_sa_group_mlp_ex mode 2) serves these padded widths
 * (IA-SSD layer 5: 256-256-512 / 256-512-1024). */
int sps_sa_group_mlp_supported_stream(int c1, int c2, int c3, int nsample);
/* 1 if the exact-fp32 kernel for point-major features (sps_sa_group_mlp_ex mode 4) serves c_feat channels and these
 * padded widths. */
int sps_sa_group_mlp_pm_supported(int c_feat, int c1, int c2, int c3, int nsample);
/* Layer 1 of that kernel over the FEATURE channels once per point instead of once per grouped point (a point's features
 * meet the same weights in every ball the point falls into: nsample m / n times, 16 at IA-SSD layer 2; the reference's
 * SharedMLP, pointnet2_modules.py:114-122 via pcdet's pt_utils, multiplies them every time): out (npts, c1) = b1 + W1f .
 * features_pm[point] (npts = b n points of the (b, n, c_feat) twin; w1 / b1 as packed for mode 4).  The grouped launch then
 * takes `out` as its feature tensor with c_feat = c1 and mode bit 32: per grouped point only the coordinate k-step is left
 * of layer 1.  Sums in a different order than without it (within 1e-4 of torch either way). */
int sps_sa_layer1_per_point(int npts, int c_feat, int c1, const float *features_pm, const float *w1, const float *b1,
                            float *out, sps_stream_t stream);
int sps_sa_layer1_per_point_supported(int c_feat, int c1, int nsample);
/* Arithmetic of sps_sa_group_mlp: 0 = exact fp32 MFMA (default), 1 = split-fp16: every operand as hi+lo halves,
 * three v_mfma_f32_16x16x16_f16 per product block, fp32 accumulate (~1e-6 relative, csrc/sa_mlp_f16.hip).  The
 * weight buffers passed afterwards must be packed for the selected mode (spsnet_amd/fused.py).  Returns the old mode. */
int sps_set_mlp_precision(int mode);

/* ---- chunked layer 0: consume FPS output while FPS is still running (spsnet_amd/sa_stack.py) ------------
 * sps_fps_publish: the pruned FPS kernel (6144 <= n <= 16384) storing its samples write-through and publishing
 * progress[scene] = samples written so far every 64 samples (device i32 per scene; the CALLER zeroes it before
 * releasing any consumer stream).
 * sps_wait_progress: enqueue a bounded spin on `stream` until every scene has published `need` samples.
 * The *_range variants restrict the centroid loop of every scene to [j0, j0+jcount) (jcount a multiple of 64
 * for the ball query / MLP); buffers keep their full (B, M, ...) shapes.
 * Correct-or-redo: a bounded wait that gives up sets *timed_out and its consumers run on samples that were never written
 * (harmless: indices are clamped into the cloud).  The range kernels, sps_sa_group_mlp_packed and sps_pointwise_mlp_ex take
 * `run_if` (device i32, may be NULL): a launch with *run_if == 0 returns at once; and `full_range_if` (device i32, may be
 * NULL): a launch with *full_range_if != 0 covers all centroids [0, m) of every scene instead of its range.  The LAST
 * chunk of a streamed layer -- whose wait is patient, so its inputs are final -- passes full_range_if = timed_out: it
 * redoes every chunk when a wait did give up and costs nothing otherwise (one predicated centroid gather is the only extra
 * launch).  force_redo = timed_out goes to sps_fps_ordered_prefix_finish, which then recomputes every scene.
 * (The spin bounds of the waits can be forced from tests: include/spsnet_sa_debug.h.)  A cross-workgroup poll inside the FPS
 * kernels that runs out never traps: it raises the scene's give-up word, the scene's workgroups leave, and the launcher's
 * follow-up launch (always enqueued, normally empty) samples the scene with the one-workgroup kernel. */
/* sps_wait_progress sets timed_out[0 .. b) (one flag per scene, all of them) when it gives up.
 * sps_fps_redo_where: the ordinary FPS for the scenes with redo[scene] != 0 only (temp must be pre-filled with 1e10 for
 * them); the other scenes keep idxs / temp untouched. */
int sps_fps_redo_where(int b, int n, int m, const float *dataset, float *temp, int *idxs, const int *redo,
                       sps_stream_t stream);
/* temp may be NULL in sps_fps_publish: the running distances start at 1e10 and are not written back. */
int sps_fps_publish(int b, int n, int m, const float *dataset, float *temp, int *idxs, int *progress,
                    sps_stream_t stream);
/* The same for scenes of 16 385 .. 262 144 points (sampling_gpu.cu:93-253 again): `work` = b * sps_fps_workspace_floats(n)
 * floats; served by the clustered large-scene kernel for batches of at most 64 / K workgroups, SPS_ERR_INVALID otherwise
 * (the caller then runs the layer unstreamed).  6144 <= n <= 16 384: the register-resident kernel of sps_fps_publish; with
 * a workspace its scenes are sorted by a pre-pass of up to 8 workgroups per scene (fps_presort.hip; same picks), without
 * one (NULL) inside the kernel.  The pre-pass keeps the flags of its one cross-workgroup exchange in a 512 KiB pool the
 * library allocates once per device on first use (the only allocation inside the library; everything else is the caller's). */
int sps_fps_publish_ws(int b, int n, int m, const float *dataset, float *temp, int *idxs, int *progress, float *work,
                       sps_stream_t stream);
int sps_wait_progress(const int *progress, int b, int need, int *timed_out, sps_stream_t stream);
/* patient != 0: the wait for the producer's last sample -- ~64 x the bound, not affected by the diagnostic spin bound (spsnet_sa_debug.h). */
int sps_wait_progress_ex(const int *progress, int b, int need, int *timed_out, int patient, sps_stream_t stream);
int sps_gather_xyz_range(int b, int n, int m, int j0, int jcount, const float *xyz, const int *idx, float *out,
                         const int *run_if, sps_stream_t stream);
/* gather_idx (device i32 (b, m), may be NULL; per-wave launches only): the centroids of the range are
 * xyz[gather_idx[scene][j]] and the launch also WRITES them to new_xyz (gather_operation fused in, pointnet2_modules.py:423-424;
 * indices are clamped into the cloud). */
int sps_ball_query_full2_range(int b, int n, int m, int j0, int jcount, float radius_a, int nsample_a,
                               float radius_b, int nsample_b, const float *new_xyz, const float *xyz, int *idx_a,
                               int *idx_b, int *perm_work, const int *run_if, const int *full_range_if,
                               const int *gather_idx, sps_stream_t stream);
int sps_sa_group_mlp_range(int b, int n, int m, int j0, int jcount, int c_feat, int nsample, const float *xyz,
                           const float *new_xyz, const float *features, const int *idx, int c1, int c2, int c3,
                           int c3_real, const float *w1, const float *b1, const float *w2, const float *b2,
                           const float *w3, const float *b3, float *out, int out_c_total, int out_c_off,
                           sps_stream_t stream);

/* sps_sa_group_mlp_range with the arithmetic chosen per call (split_fp16 = 0: fp32 MFMA; 1: split-fp16, per-layer
 * fragment arrays; 2: split-fp16 with ONE concatenated fragment stream in w1 that a workgroup's waves share through
 * LDS -- first hidden width a multiple of 64; w2 / w3 are ignored; 3: pure fp16 -- `features` points at HALVES
 * (fp16 features in HBM, BASELINE configs[4]), weights are fp16 fragments of 1 KiB, one MFMA per product block, fp32
 * accumulate, hidden widths multiples of 32; weights packed accordingly by the caller; + 4:
 * `features` is point-major (b, n, c_feat), c_feat % 4 == 0, and layer 1's input channels are ordered
 * [features, xyz] instead of [xyz, features] -- a neighbour's channels are then contiguous 16-byte loads; with
 * arithmetic 0 (value 4: exact fp32 on point-major features, csrc/sa_mlp_pm.hip) c_feat % 16 == 0, w1 holds the
 * coordinate fragments [c1/16][64] followed by layer 1 over the features packed like w2 / w3, and the widths must be
 * listed by sps_sa_group_mlp_pm_supported) and, for the split-fp16 kernel, a device flag that is set to 1 if an operand left the exactly splittable range
 * (|x| > 65 504: the value was clamped).  overflow_flag may be NULL. */
int sps_sa_group_mlp_ex(int b, int n, int m, int j0, int jcount, int c_feat, int nsample, const float *xyz,
                        const float *new_xyz, const float *features, const int *idx, int c1, int c2, int c3,
                        int c3_real, const float *w1, const float *b1, const float *w2, const float *b2,
                        const float *w3, const float *b3, float *out, int out_c_total, int out_c_off,
                        int split_fp16, int *overflow_flag, sps_stream_t stream);

/* ---- packed columns: only the distinct neighbours of a ball go through the grouped MLP -------------------------------
 * A ball-query row repeats its first hit in the slots it could not fill (ball_query_gpu.cu:35-42) and max-pooling is
 * idempotent, so the repeated columns need not be computed: same pooled features bit for bit.  sps_pack_columns turns
 * idx (b, m, nsample), centroids [j0, j0 + jcount) of every scene, into a stream of 16-column tiles (cols = point index,
 * meta = centroid / scene / slot size per column, *ntiles = tiles written; the caller zeroes *ntiles and provides
 * tile_cap >= sps_pack_columns_capacity(b, jcount, nsample) tiles of 16 ints each).  nsample <= 64, m < 2^20, b <= 256. */
long long sps_pack_columns_capacity(int b, int jcount, int nsample);
int sps_pack_columns(int b, int m, int j0, int jcount, int nsample, const int *idx, int *cols, unsigned *meta, int *ntiles,
                     long long tile_cap, sps_stream_t stream);
/* both grouping scales of a layer (same b, m, range) in one launch */
int sps_pack_columns2(int b, int m, int j0, int jcount, int nsample_a, const int *idx_a, int *cols_a, unsigned *meta_a,
                      int *ntiles_a, long long tile_cap_a, int nsample_b, const int *idx_b, int *cols_b, unsigned *meta_b,
                      int *ntiles_b, long long tile_cap_b, sps_stream_t stream);
/* ---- a layer that starts on a partly written cloud (spsnet_amd/sa_stack.py: layer k + 1 samples the first picks of layer
 * k's D-FPS -- the verified identity prefix -- so its centroids exist long before layer k's last pick does) ---------------
 * Max-pooling is order-independent and a ball-query row is "the first nsample hits in index order": the row over the points
 * [0, k) is a prefix of the complete row.  So the columns of the early points go through the grouped MLP while the producer
 * still runs -- in stages, as the cloud grows -- and behind its last pick only the LAST stage's columns are left:
 * sps_ball_query_full2_points (rows over a point range), sps_pack_columns2_late (of a stage's rows, the columns the complete
 * rows would hold, given how many every centroid took before) and sps_sa_group_mlp_packed_merge with mode + 16
 * (pooled rows merged into `out` by an atomic max on the non-negative bit patterns).  Correct or redo: each of the three takes
 * the repair predicate -- one device flag OR any of `count` per-scene flags -- and then covers the whole cloud / packs whole
 * rows / stores plainly, i.e. recomputes the layer from scratch inside the same launches. */
int sps_ball_query_full2_points(int b, int n, int m, int k0, int kcount, float radius_a, int nsample_a, float radius_b,
                                int nsample_b, const float *new_xyz, const float *xyz, int *idx_a, int *idx_b,
                                const int *all_points_if, const int *all_points_if_any, int any_count, sps_stream_t stream);
/* ... with gather_idx (device i32 (b, m), may be NULL): the centroids are xyz[gather_idx[scene][j]] and the launch also writes
 * them to new_xyz (as sps_ball_query_full2_range does): the last stage takes the sampler's verified picks here instead of a
 * gather launch in front of it (pointnet2_modules.py:423-424 fused in). */
int sps_ball_query_full2_points_gather(int b, int n, int m, int k0, int kcount, float radius_a, int nsample_a, float radius_b,
                                       int nsample_b, const float *new_xyz, const float *xyz, int *idx_a, int *idx_b,
                                       const int *all_points_if, const int *all_points_if_any, int any_count,
                                       const int *gather_idx, sps_stream_t stream);
int sps_pack_columns2_late(int b, int m, int last_stage, int nsample_a, const int *prev_a, const int *idx_a, int *taken_a, int *cols_a,
                           unsigned *meta_a, int *ntiles_a, long long tile_cap_a, int nsample_b, const int *prev_b,
                           const int *idx_b, int *taken_b, int *cols_b, unsigned *meta_b, int *ntiles_b, long long tile_cap_b,
                           const int *full_if, const int *full_if_any, int any_count, sps_stream_t stream);
int sps_sa_group_mlp_packed_merge(int b, int n, int m, int j0, int jcount, int c_feat, int nsample, const float *xyz,
                                  const float *new_xyz, const float *features, const int *idx, const int *cols,
                                  const unsigned *meta, const int *ntiles, long long tile_cap, int c1, int c2, int c3,
                                  int c3_real, const float *w1, const float *b1, const float *w2, const float *b2,
                                  const float *w3, const float *b3, float *out, int out_c_total, int out_c_off,
                                  int split_fp16, int *overflow_flag, const int *run_if, const int *full_range_if,
                                  const int *unless_any, int unless_count, sps_stream_t stream);
/* sps_sa_group_mlp_ex over either idx + range or (cols != NULL) a packed column stream; split_fp16 + 8: `out` is
 * point-major (b, m, out_c_total).  With packed columns and nsample 64 `out` must be zero-filled as before. */
int sps_sa_group_mlp_packed(int b, int n, int m, int j0, int jcount, int c_feat, int nsample, const float *xyz,
                            const float *new_xyz, const float *features, const int *idx, const int *cols,
                            const unsigned *meta, const int *ntiles, long long tile_cap, int c1, int c2, int c3,
                            int c3_real, const float *w1, const float *b1, const float *w2, const float *b2,
                            const float *w3, const float *b3, float *out, int out_c_total, int out_c_off,
                            int split_fp16, int *overflow_flag, const int *run_if, const int *full_range_if,
                            sps_stream_t stream);

/* Aggregation stack (+ confidence head) of an SA layer as one kernel -- replaces, for inference with BatchNorm folded,
 * Conv1d+BN+ReLU (pointnet2_modules.py:213-228, 449-450) and Conv1d+BN+ReLU, Conv1d(bias) (:230-245, 454-455).
 * x (b, cin, m) -> y1 (b, c1, m) = relu(W1 x + b1); if w2 != NULL also y3 (b, m, classes) = W3 relu(W2 y1 + b2) + b3.
 * m, cin, c1, c2 multiples of 16; classes <= 16 (w3 / b3 padded to 16 rows); weights in the fragment order of
 * spsnet_amd/fused.py:_pack_pw.  Exact fp32 (MFMA f32). */
int sps_pointwise_mlp(int b, int m, int cin, int c1, int c2, int classes, const float *x, const float *w1,
                      const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, float *y1,
                      float *y1_point_major /* optional (b, m, c1) copy of y1, or NULL */, float *y3, sps_stream_t stream);
/* The same for the points [j0, j0 + jcount) of every scene only (multiples of 16): lets a caller that produces x chunk by
 * chunk (the streamed first layer) run the tail chunk by chunk too. */
int sps_pointwise_mlp_range(int b, int m, int j0, int jcount, int cin, int c1, int c2, int classes, const float *x,
                            const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                            const float *b3, float *y1, float *y1_point_major, float *y3, sps_stream_t stream);
/* The same with flags: 1 = y1 / y1_point_major are fp16 buffers (features stored as halves in HBM: BASELINE configs[4];
 * the arithmetic stays fp32 and the class scores are computed from the rounded features), 2 = x is point-major (B, M, cin). */
int sps_pointwise_mlp_ex(int b, int m, int j0, int jcount, int cin, int c1, int c2, int c3_real, const float *x,
                         const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                         const float *b3, void *y1, void *y1_point_major, float *y3, int flags, const int *run_if,
                         const int *full_range_if, sps_stream_t stream);

/* farthest_point_sampling_kernel_launcher (sampling_gpu.cu:93-253) with an optional device workspace of
 * b * sps_fps_workspace_floats(n) floats (0 for sizes that need none).  With it, scenes of 16 385 .. 262 144 points
 * take the spatially pruned large-scene kernel (same indices and final `temp`, bit for bit; ~100x faster than the
 * streaming brute-force sweep at Waymo sizes); without it this is sps_farthest_point_sampling_kernel_launcher.
 * Batches of at most 64 / K scenes are spread over K <= 16 workgroups per scene (8 through round 4) that exchange records every round
 * (fps_pruned_cluster.hip; same results): those b * K workgroups spin on each other and must be resident together, so do
 * not issue such a launch on a stream confined to a few compute units by a CU mask (environment SPS_FPS_CLUSTER=1 selects
 * the one-workgroup kernel, "K,T" a shape).  The workspace includes the exchange area, which the launch zeroes itself. */
long long sps_fps_workspace_floats(int n);
int sps_fps_with_workspace(int b, int n, int m, const float *dataset, float *temp, int *idxs, float *work,
                           sps_stream_t stream);

/* Deterministic form of group_points_grad / gather_points_grad (group_points_gpu.cu:53-71, sampling_gpu.cu:46-63, which
 * scatter with atomicAdd in an unspecified order): grad_points (b, c, n) += grad_out (b, c, cols) scattered by
 * idx (b, cols), every target summed in ascending column order -- the order of a sequential loop, so the result is
 * bit-identical from run to run.  cols = npoints * nsample (group) or npoints (gather); work = device ints,
 * sps_index_add_workspace_ints(b, n, cols) of them. */
long long sps_index_add_workspace_ints(int b, int n, int cols);
int sps_index_add_deterministic(int b, int c, int n, int cols, const float *grad_out, const int *idx, float *grad_points,
                                int *work, sps_stream_t stream);

/* DenseEdgeConv.forward (surface_feature.py:98-116) given the neighbour table of its radius query, as one kernel:
 * x (b, n, d) point-major features, idx (b, n, k) from ball_query(radius, k, pos, pos) (:55, 84-89) ->
 * out (b, n, d + 3*growth) = [max_k y3 | max_k y2 | max_k y1 | x], y1 = relu(W1 e + b1) over the edge features
 * e = [x_i, x_j, x_j - x_i] (relative_only = 0; = 1: x_j - x_i alone, :71-80; = 2: as 0 with the centre / neighbour /
 * difference weight blocks merged into two on the host, w1 = [(W1a - W1c) | (W1b + W1c)], ~1e-6 relative off), y2 = relu(W2 [y1, x_i] + b2),
 * y3 = W3 [y2, y1, x_i] + b3.  Built for the configuration the reference instantiates (FeatureExtraction defaults,
 * :120-131: d = 24, k = 16, growth = 12, three FC layers, ReLU, max); weights in the fragment order of
 * spsnet_amd/fused.py:pack_dense_edge_conv.  Exact fp32 (MFMA f32). */
int sps_dense_edge_conv(int b, int n, int d, int k, int growth, int relative_only, const float *x, const int *idx,
                        const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                        const float *b3, float *out, sps_stream_t stream);

/* FCLayer (surface_feature.py:8-27) on point-major rows: out (rows, cout) = act(x (rows, cin) W^T + bias), W (cout, cin)
 * as nn.Linear stores it, act = ReLU if relu != 0.  Built for cout = 24 (FeatureExtraction's conv_channels), cin <= 64. */
int sps_linear_rows(long long rows, int cin, int cout, const float *x, const float *w, const float *bias, int relu,
                    float *out, sps_stream_t stream);

/* ball_query_kernel_fast / ball_query_dilated_kernel_fast (ball_query_gpu.cu:9-45, 70-117) through a cell grid: the same
 * rows as sps_ball_query_kernel_launcher_fast / ..._dilated_..., bit for bit, but only the points of the cells a group
 * of 64 neighbouring centroids can reach are tested (visited in ascending index through an LDS bitmap), instead of
 * all n.  work = device ints, sps_ball_query_grid_workspace_ints(b, n, m) of them (0 = size not supported: the call then
 * runs the scan).  dilated != 0: hit if d2 == 0 or min_radius^2 <= d2 < max_radius^2.  fill_empty != 0: rows of empty
 * balls are written as zeros instead of being left to the caller (pointnet2_utils.py:246). */
long long sps_ball_query_grid_workspace_ints(int b, int n, int m);
int sps_ball_query_grid(int b, int n, int m, float max_radius, float min_radius, int dilated, int nsample, int fill_empty,
                        const float *new_xyz, const float *xyz, int *idx, int *work, sps_stream_t stream);
/* The two grouping radii of an SA layer (pointnet2_modules.py:429-447 calls ball_query once per scale) from one walk
 * of the grid: same rows as sps_ball_query_full2 (every row written, zeros for empty balls), same workspace. */
int sps_ball_query_grid2(int b, int n, int m, float radius_a, int nsample_a, float radius_b, int nsample_b,
                         const float *new_xyz, const float *xyz, int *idx_a, int *idx_b, int *work, sps_stream_t stream);

/* ---- pcdet/ops/pointnet2/pointnet2_stack: ragged batches (scenes concatenated along the point axis, per-scene counts).
 * Argument order = the reference launchers' (+ stream; device counters where the reference returns a count). ---- */
/* ball_query_kernel_launcher_stack (pointnet2_stack/src/ball_query_gpu.cu:15-90): new_xyz (M,3), xyz (N,3), idx (M,nsample)
 * pre-zeroed by the caller; first nsample hits in index order (indices local to the scene), idx[row][0] = -1 if empty. */
int sps_ball_query_kernel_launcher_stack(int b, int m, float radius, int nsample, const float *new_xyz,
                                         const int *new_xyz_batch_cnt, const float *xyz, const int *xyz_batch_cnt, int *idx,
                                         sps_stream_t stream);
/* voxel_query_kernel_launcher_stack (voxel_query_gpu.cu:12-113): neighbours through point_indices (B,R1,R2,R3) within
 * +-range voxels of new_coords (M,4) [b,z,y,x]; hit if d2 <= radius^2; global point indices; idx[row][0] = -1 if empty. */
int sps_voxel_query_kernel_launcher_stack(int m, int r1, int r2, int r3, int nsample, float radius, int z_range, int y_range,
                                          int x_range, const float *new_xyz, const float *xyz, const int *new_coords,
                                          const int *point_indices, int *idx, sps_stream_t stream);
/* stack_farthest_point_sampling_kernel_launcher (sampling_gpu.cu:187-348): per-scene FPS, global indices, temp (N)
 * pre-filled with 1e10, tie rule of the fixed 1024-thread block. */
int sps_stack_farthest_point_sampling_kernel_launcher(int n_total, int batch_size, const float *dataset, float *temp,
                                                      const int *xyz_batch_cnt, int *idxs, const int *num_sampled_points,
                                                      sps_stream_t stream);
/* group_points(_grad)_kernel_launcher_stack (group_points_gpu.cu:14-125): features (N,C), idx (M,nsample) local to the
 * scene -> out (M,C,nsample); grad: atomicAdd scatter into grad_features (N,C). */
int sps_group_points_kernel_launcher_stack(int b, int m, int c, int nsample, const float *features,
                                           const int *features_batch_cnt, const int *idx, const int *idx_batch_cnt, float *out,
                                           sps_stream_t stream);
int sps_group_points_grad_kernel_launcher_stack(int b, int m, int c, int n, int nsample, const float *grad_out, const int *idx,
                                                const int *idx_batch_cnt, const int *features_batch_cnt, float *grad_features,
                                                sps_stream_t stream);
/* three_nn / three_interpolate(_grad)_kernel_launcher_stack (interpolate_gpu.cu:14-194): global indices into `known`. */
int sps_three_nn_kernel_launcher_stack(int batch_size, int n, int m, const float *unknown, const int *unknown_batch_cnt,
                                       const float *known, const int *known_batch_cnt, float *dist2, int *idx,
                                       sps_stream_t stream);
int sps_three_interpolate_kernel_launcher_stack(int n, int channels, const float *features, const int *idx,
                                                const float *weight, float *out, sps_stream_t stream);
int sps_three_interpolate_grad_kernel_launcher_stack(int n, int channels, const float *grad_out, const int *idx,
                                                     const float *weight, float *grad_features, sps_stream_t stream);

/* F.max_pool2d(x, kernel_size=[1, nsample]) of the SA modules (pointnet2_modules.py:441-444) and its gradient on a
 * contiguous (rows, nsample) view, rows = B*C*M, nsample <= 255: out (rows) = row maxima, arg (rows) u8 = position of the
 * FIRST maximum (a NaN wins and propagates), grad_in (rows, nsample) = grad_out at arg, zero elsewhere. */
int sps_pool_max_fwd(long long rows, int nsample, const float *x, float *out, unsigned char *arg, sps_stream_t stream);
int sps_pool_max_bwd(long long rows, int nsample, const float *grad_out, const unsigned char *arg, float *grad_in,
                     sps_stream_t stream);

/* BatchNorm2d on batch statistics + ReLU (the [Conv2d, BatchNorm2d, ReLU] stacks of pointnet2_modules.py:203-209 in
 * train() mode), forward and backward, on a contiguous (b, c, l) tensor, l = M * nsample.  torch semantics: biased variance
 * to normalise, unbiased for running_var, running = (1 - momentum) running + momentum batch; weight / bias /
 * running_* may be NULL.  mean, invstd (c) are outputs of the forward and inputs of the backward; work = doubles,
 * sps_bn_train_workspace_doubles(b, c, l) of them; scratch2c = 2 c floats.  Fixed-order reductions (reproducible). */
long long sps_bn_train_workspace_doubles(int b, int c, long long l);
int sps_bn_relu_train_fwd(int b, int c, long long l, const float *x, const float *weight, const float *bias, float eps,
                          float momentum, float *running_mean, float *running_var, float *mean, float *invstd, float *y,
                          double *work, sps_stream_t stream);
int sps_bn_relu_train_bwd(int b, int c, long long l, const float *x, const float *dy, const float *mean,
                          const float *invstd, const float *weight, const float *bias, float *dx, float *dweight,
                          float *dbias, float *scratch2c, double *work, sps_stream_t stream);

/* Vector-pool family (pointnet2_stack/src/vector_pool_gpu.cu): one thread per new_xyz walks its scene in index order.
 * query_stacked_local_neighbor_idxs (:117-187): per centre, the support points inside a ball (neighbor_type 1) or cube of
 * max_neighbour_distance (first nsample if > 0, at most 1000), packed into stack_neighbor_idxs at a position taken from
 * *cumsum (device counter, caller-zeroed); start_len (M,2) = [start, length].  Lists are truncated at avg_length * M. */
int sps_query_stacked_local_neighbor_idxs_kernel_launcher_stack(
    const float *support_xyz, const int *xyz_batch_cnt, const float *new_xyz, const int *new_xyz_batch_cnt,
    int *stack_neighbor_idxs, int *start_len, int *cumsum, int avg_length_of_neighbor_idxs, float max_neighbour_distance,
    int batch_size, int m, int nsample, int neighbor_type, sps_stream_t stream);
/* query_three_nn_by_stacked_local_idxs (:14-72): three nearest of each centre's local list for every grid centre
 * (M, num_total_grids, 3); missing neighbours repeat the first, an empty list gives -1. */
int sps_query_three_nn_by_stacked_local_idxs_kernel_launcher_stack(
    const float *support_xyz, const float *new_xyz, const float *new_xyz_grid_centers, int *new_xyz_grid_idxs,
    float *new_xyz_grid_dist2, const int *stack_neighbor_idxs, const int *start_len, int m, int num_total_grids,
    sps_stream_t stream);
/* vector_pool_kernel_launcher_stack (:239-381): sums (pooling_type 0) or first-point picks (1) of the support features per
 * local grid cell; outputs pre-zeroed by the caller; grouped_idxs (num_max_sum_points, 3) rows in arbitrary order;
 * *cum_sum (device counter, caller-zeroed) ends as the value the reference returns. */
int sps_vector_pool_kernel_launcher_stack(
    const float *support_xyz, const float *support_features, const int *xyz_batch_cnt, const float *new_xyz,
    float *new_features, float *new_local_xyz, const int *new_xyz_batch_cnt, int *point_cnt_of_grid, int *grouped_idxs,
    int num_grid_x, int num_grid_y, int num_grid_z, float max_neighbour_distance, int batch_size, int n, int m, int num_c_in,
    int num_c_out, int num_total_grids, int use_xyz, int num_max_sum_points, int nsample, int neighbor_type, int pooling_type,
    int *cum_sum, sps_stream_t stream);
/* vector_pool_grad_kernel_launcher_stack (:383-431) */
int sps_vector_pool_grad_kernel_launcher_stack(const float *grad_new_features, const int *point_cnt_of_grid,
                                               const int *grouped_idxs, float *grad_support_features, int n, int m,
                                               int num_c_out, int num_c_in, int num_total_grids, int num_max_sum_points,
                                               sps_stream_t stream);

/* Gradient of sps_dense_edge_conv (DenseEdgeConv.forward, surface_feature.py:98-116) for training: recomputes the
 * convolution per centre and back-propagates through it.  relative_only as in the forward (0 or 1).  w_fwd = the forward's
 * fragments [w1 | w2 | w3] concatenated, w_transposed = 44 x 64 fragments of the transposed blocks (fused.py:
 * pack_dense_edge_conv_bwd).  Outputs: dx_centre (b, n, 24) = gradient through a point's role as centre (+ the
 * pass-through channels of the output), dx_neighbour (b, 24, n*16) = gradient per (centre, neighbour column), to be
 * scattered by idx (sps_group_points_grad_kernel_launcher_fast), grad_tiles (13 | 9 tiles of 16 x 16) = weight / bias
 * gradients in tile form (fused.py unpacks them); partial = scratch of sps_dense_edge_conv_bwd_blocks() * tiles * 256
 * floats.  Fixed-order reductions: reproducible. */
int sps_dense_edge_conv_bwd_blocks(void);
int sps_dense_edge_conv_bwd(int b, int n, int d, int k, int growth, int relative_only, const float *x, const int *idx,
                            const float *grad_out, const float *w_fwd, const float *w_transposed, const float *b1,
                            const float *b2, const float *b3, float *dx_centre, float *dx_neighbour, float *partial,
                            float *grad_tiles, sps_stream_t stream);

/* Gradient of sps_linear_rows: dx (rows, cin) = dy' w, grad_w_b (cout * cin + cout) = [dy'^T x | column sums of dy'], where
 * dy' = dy masked by y > 0 when relu != 0.  partial = scratch of sps_linear_rows_bwd_blocks() * (cout * cin + cout) floats.
 * Fixed-order reductions. */
int sps_linear_rows_bwd_blocks(void);
int sps_linear_rows_bwd(long long rows, int cin, int cout, const float *x, const float *y, const float *dy, const float *w,
                        int relu, float *dx, float *partial, float *grad_w_b, sps_stream_t stream);

/* The 1x1 convolutions of the grouped MLPs in training (Conv2d(k=1, bias=False), pointnet2_modules.py:203-209) on
 * channel-major activations (b, c, l), l = M * nsample, exact fp32 on the matrix cores.
 * sps_conv1x1_apply: out (b, co, l) = A (co x ci) in (b, ci, l) -- the forward with A = W, the data gradient with A = W^T;
 * afrag = A in fragment order [ceil(co/16)][ceil(ci/4)][64 lanes] = A[16 t + (lane & 15)][4 ks + (lane >> 4)], zero padded
 * (pointnet2_modules.py packs it); l % 4 == 0.
 * sps_conv1x1_wgrad: dw (co, ci) = sum over scenes and columns of dy (b, co, l) x (b, ci, l)^T; l % 16 == 0; work =
 * sps_conv1x1_wgrad_workspace_floats(b, ci, co, l) floats.  Fixed-order reductions. */
int sps_conv1x1_apply(int b, int ci, int co, long long l, const float *in, const float *afrag, float *out, sps_stream_t stream);
long long sps_conv1x1_wgrad_workspace_floats(int b, int ci, int co, long long l);
int sps_conv1x1_wgrad(int b, int ci, int co, long long l, const float *x, const float *dy, float *dw, float *work,
                      sps_stream_t stream);


/* PointnetFPModule.forward (pointnet2_modules.py:539-587) in inference as one kernel: three_interpolate of known_feats
 * (b, c_known, m) with idx / weight (b, n, 3), concatenation with the skip features (b, c_skip, n) (NULL when c_skip = 0),
 * then [Conv2d 1x1 + BatchNorm2d + ReLU] x (1 | 2) with BatchNorm folded; exact fp32.  y (b, c2 ? c2 : c1, n); c1, c2
 * multiples of 16, c2 = 0 and w2 = b2 = NULL for a one-layer stack; w1 has 16 ceil((c_known + c_skip) / 16) input columns
 * (zero beyond the real ones); weights packed as for sps_pointwise_mlp. */
int sps_fp_module_mlp(int b, int n, int m, int c_known, int c_skip, int c1, int c2, const float *known_feats, const float *skip,
                      const int *idx, const float *weight, const float *w1, const float *b1, const float *w2, const float *b2,
                      float *y, sps_stream_t stream);
/* ... with the interpolation weights formed in the kernel from three_nn's distances (weights_from_dist != 0: `weight` holds
 * dist (b, n, 3); 1 / (dist + 1e-8) normalised over the three neighbours, pointnet2_modules.py:572-574) and / or the output
 * written point-major, y (b, n, c2 ? c2 : c1) (y_point_major != 0: the per-point rows a backbone hands out,
 * pcdet/models/backbones_3d/pointnet2_backbone.py:91). */
int sps_fp_module_mlp_ex(int b, int n, int m, int c_known, int c_skip, int c1, int c2, const float *known_feats, const float *skip,
                         const int *idx, const float *weight, int weights_from_dist, const float *w1, const float *b1,
                         const float *w2, const float *b2, float *y, int y_point_major, sps_stream_t stream);

/* The grouped MLP of an SA layer in TRAINING mode ([Conv2d 1x1, BatchNorm2d on batch statistics, ReLU] x n + max-pool,
 * pointnet2_modules.py:203-211, 432-444), fused along its memory passes (csrc/mlp_train.hip): only the pre-BatchNorm
 * convolution outputs Y_l and the gradients dA_l w.r.t. the post-ReLU activations exist in memory; BatchNorm, ReLU, the pool's
 * gradient routing and the BatchNorm backward are applied where their operands are loaded.  All tensors (b, c, l) fp32
 * channel-major, l = M * nsample a multiple of 64; split-fp16 MFMA arithmetic with fp32 accumulation (operands beyond +-65504
 * or NaN / Inf poison the affected outputs with NaN and raise *overflow; never clamped silently).
 *
 * params: one block of 8 floats per channel {mean, invstd, scale = gamma invstd, shift = beta - mean scale, gamma, beta,
 * c1 = mean(dZ), c2 = mean(dZ xhat)}; sps_tbn_finalize writes the first six from the statistics, sps_tbn_bwd_finalize the last
 * two (plus d gamma, d beta).
 *
 * sps_tconv: out (b, co, l) = A . T(in), A[o][i] = trans ? w[i co + o] : w[o ci + i], ci <= 288.
 *   in_mode 0: T = identity                     1: T = relu(fma(in, scale, shift))   (pin = params of the input rows)
 *           2: T = scale (dZ - c1 - xhat c2), dZ = in [fma(in2, scale, shift) > 0], xhat = (in2 - mean) invstd
 *           3: as 2 with `in` = the pooled gradient gout (b, ci, m) routed by arg (b, ci, m) u8 (nsample % 4 == 0)
 *   epi_mode 0: none   1: per-channel sum / sum of squares of out -> partial   2: sums of dZ' and dZ' xhat' of the OUTPUT rows,
 *           dZ' = out [fma(epi_y, scale', shift') > 0] (pout = params of the output rows) -> partial
 *   partial: [sps_tconv_parts(b, l, co)][co][2] doubles, summed in a fixed order by the finalize calls.
 *   wamax: device scalar, the largest |w| (the weights are scaled by a power of two before they are split as well).
 * Operand scaling: every operand tensor is multiplied by an exact power of two that brings its largest magnitude near 2^10
 * before it is split into fp16 halves (gradients are routinely 1e-5 and smaller; unscaled, their low halves would be fp16
 * denormals), and the accumulators by the inverse.  amax_in (device scalar, in_mode >= 2): the largest magnitude of the
 * incoming gradient tensor; amax_out (device scalar the caller zeroed, epi_mode 2): receives the atomic maximum of |out|.
 * sps_tpool_bwd_stats produces it for the pooled gradient, sps_twgrad consumes it. */
int sps_tconv_parts(int b, long long l, int co);
int sps_tconv(int b, int ci, int co, long long l, int in_mode, int epi_mode, int trans, const float *w, const float *in,
              const float *in2, const float *gout, const unsigned char *arg, int nsample, int m, const float *pin, float *out,
              const float *epi_y, const float *pout, double *partial, const float *amax_in, float *amax_out,
              const float *wamax, int *overflow, sps_stream_t stream);
/* torch semantics: biased variance to normalise, unbiased for running_var, running = (1 - momentum) running + momentum batch;
 * gamma / beta / running_* may be NULL; count = b * l. */
int sps_tbn_finalize(int c, int nparts, double count, const double *partial, const float *gamma, const float *beta, float eps,
                     float momentum, float *running_mean, float *running_var, float *params, long long *num_batches_tracked,
                     sps_stream_t stream);   /* num_batches_tracked (int64 device scalar, may be NULL) += 1 */
/* the same with the element count on the device (count_dev, may be NULL = use `count`): nn.SyncBatchNorm's GLOBAL count,
 * all-reduced beside the sums so that ranks with different local batch sizes agree (torch gathers per-rank counts too) */
int sps_tbn_finalize_dc(int c, int nparts, double count, const double *count_dev, const double *partial, const float *gamma,
                        const float *beta, float eps, float momentum, float *running_mean, float *running_var, float *params,
                        long long *num_batches_tracked, sps_stream_t stream);
int sps_tbn_bwd_finalize_dc(int c, int nparts, double count, const double *count_dev, const double *partial, float *params,
                            float *dgamma, float *dbeta, sps_stream_t stream);
/* out[k] = max |p_k[0 .. n_k)| for `count` <= 4 arrays in one launch (the `wamax` scalars of a grouped MLP's layers) */
int sps_tamax4(int count, const float *p0, long long n0, const float *p1, long long n1, const float *p2, long long n2,
               const float *p3, long long n3, float *out, sps_stream_t stream);
int sps_tbn_bwd_finalize(int c, int nparts, double count, const double *partial, float *params, float *dgamma, float *dbeta,
                         sps_stream_t stream);
/* out (b, c, m) = max over the nsample columns of relu(fma(y, scale, shift)), arg = position of the FIRST maximum (a NaN wins
 * and propagates: torch's max_pool2d), yarg = y at that position; nsample in {4, 8, 16, 32, 64}. */
int sps_tpool_fwd(int b, int c, int m, int nsample, const float *y, const float *params, float *out, unsigned char *arg,
                  float *yarg, sps_stream_t stream);
/* the BatchNorm-backward sums of the LAST layer from the pooled gradient: partial (b, c, 2) doubles = b parts; amax_out
 * (zeroed device scalar, may be NULL) receives max |gout| */
int sps_tpool_bwd_stats(int b, int c, int m, const float *yarg, const float *gout, const float *params, double *partial,
                        float *amax_out, sps_stream_t stream);
/* The same stack WITHOUT a pool (aggregation / confidence layers: Conv1d + BatchNorm1d + ReLU on (b, c, l) tensors,
 * pointnet2_modules.py:213-245, 449-455): out = relu(fma(y, scale, shift)), and the last layer's BatchNorm-backward sums from a
 * dense incoming gradient dA (partial (b, c, 2) doubles = b parts; amax_out receives max |dA|).  l a multiple of 4. */
int sps_tbn_apply_relu(int b, int c, long long l, const float *y, const float *params, float *out, sps_stream_t stream);
int sps_tbn_bwd_stats(int b, int c, long long l, const float *y, const float *dA, const float *params, double *partial,
                      float *amax_out, sps_stream_t stream);
/* dw (co, ci) = sum over all columns of dY (x) T(x): dY from (dA, y, pd) as in_mode 2 / 3 of sps_tconv (dmode), T = identity
 * (xmode 0) or relu(fma(x, scale, shift)) with px (xmode 1); co, ci <= 256, l a multiple of 32; work =
 * sps_twgrad_workspace_floats floats. */
/* Arithmetic of every sps_tconv / sps_twgrad launch and of sps_mlp_train_forward / _backward (process-wide, like
 * sps_set_mlp_precision): 0 = split-fp16 (every fp32 operand as hi + lo halves behind an exact power-of-two scaling, three
 * v_mfma_f32_16x16x32_f16 per product block; the library default at this level), 1 = EXACT fp32 on v_mfma_f32_16x16x4_f32 --
 * the reference's arithmetic (fp32 Conv2d / BatchNorm2d, pointnet2_modules.py:203-211): nothing is scaled, split, range-checked
 * or poisoned, wamax / amax_in / amax_out may be NULL and the overflow flag is never raised.  Returns the previous mode.
 * (spsnet_amd.fused.TRAIN_PRECISION, whose default IS "fp32", sets it in front of every launch.) */
int sps_set_train_precision(int mode);
long long sps_twgrad_workspace_floats(int b, int co, int ci, long long l);
int sps_twgrad(int b, int co, int ci, long long l, int dmode, int xmode, const float *dA, const float *y, const float *gout,
               const unsigned char *arg, int nsample, int m, const float *pd, const float *x, const float *px,
               const float *amax_in, float *dw, float *work, int *overflow, sps_stream_t stream);

/* ---- one call per grouped MLP of a TRAINING step (csrc/mlp_train.hip; pointnet2_modules.py:432-444 of the reference in train()
 * mode: [Conv2d 1x1, BatchNorm2d on batch statistics, ReLU] x n + max-pool, and its backward).  The launch-by-launch entry points
 * above composed in C: a backbone step issues ~220 of them and is host-bound from Python.  Every buffer is the caller's:
 *   x (b, c[0], m, ns); w[k] (c[k+1], c[k]); y[k] (b, c[k+1], m ns) pre-BatchNorm outputs; params[k] (c[k+1], 8);
 *   wamax (n); partial = sps_mlp_train_partial_doubles(desc) doubles; out / yarg (b, c[n], m) floats, arg the same in bytes;
 *   backward: gout (b, c[n], m); dA[k] (b, c[k], m ns) for k = 1 .. n-1, dA[0] = dx or NULL; dw[k] (c[k+1], c[k]) or NULL;
 *   dgamma[k], dbeta[k] (c[k+1]); amax (n); work = max_k sps_twgrad_workspace_floats(b, c[k+1], c[k], m ns) floats.
 * ns = 0: the same stack WITHOUT a pool on (b, c, m) tensors -- an aggregation / confidence / vote stack, [Conv1d, BatchNorm1d,
 * ReLU] x n (pointnet2_modules.py:213-245, 449-455, 470-478): x (b, c[0], m), y[k] / dA[k] (b, c[k], m), out (b, c[n], m) =
 * relu(bn_n(...)) dense, gout dense, arg / yarg unused (may be NULL); m a multiple of 64.
 * Arithmetic: sps_set_train_precision (exact fp32: wamax / amax are not read).
 * running_mean / running_var / num_batches_tracked are updated as nn.BatchNorm2d does.  Nothing is allocated or synchronised. */
typedef struct sps_mlp_train_desc {
    int n, b, m, ns;
    int c[5];
    const float *w[4];
    const float *gamma[4], *beta[4];
    float eps[4], momentum[4];
    float *running_mean[4], *running_var[4];
    long long *num_batches_tracked[4];
    const float *x;
    float *y[4];
    float *params[4];
    float *wamax;
    double *partial;
    float *out;
    unsigned char *arg;
    float *yarg;
    int *overflow;
    const float *gout;
    float *dA[4];
    float *dw[4];
    float *dgamma[4], *dbeta[4];
    float *amax;
    float *work;
} sps_mlp_train_desc;
long long sps_mlp_train_partial_doubles(const sps_mlp_train_desc *desc);
int sps_mlp_train_forward(const sps_mlp_train_desc *desc, sps_stream_t stream);
int sps_mlp_train_backward(const sps_mlp_train_desc *desc, sps_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPSNET_SA_H */
