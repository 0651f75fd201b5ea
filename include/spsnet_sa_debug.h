/*
 * spsnet_sa_debug.h -- DIAGNOSTIC entry points of libspsnet_sa.so.  NOT part of the drop-in boundary (include/spsnet_sa.h):
 * nothing here has a counterpart in the reference's pointnet2_batch_cuda extension and nothing on the product path calls
 * it.  Used by tests/ (to force the correct-or-redo paths) and tools/ (per-phase cycle profiles, CU-mask probes).
 */
#ifndef SPSNET_SA_DEBUG_H
#define SPSNET_SA_DEBUG_H

#include "spsnet_sa.h"

#ifdef __cplusplus
extern "C" {
#endif

/* s_memtime-instrumented build of the pruned FPS kernel (csrc/fps_pruned.hip, the PROF instantiation).
 * dbg (B, 8 waves, 12) u64 receives per-wave cycle sums of the loop segments, touched-bucket / tie-path counts and why the
 * accepted prefix of a round ended (lowered, hidden, nothing rejected; picks). */
int sps_debug_fps_profile(int b, int n, int m, const float *dataset, float *temp, int *idxs,
                          unsigned long long *dbg, sps_stream_t stream);

/* one {XCC_ID, HW_ID} register pair per workgroup -> out (blocks, 2) u32; tools/cumask_probe.py uses it to print which
 * physical compute units a CU-masked stream reaches. */
int sps_debug_where(int blocks, int threads, int spin, unsigned *out, sps_stream_t stream);

/* sps_debug_set_wait_spins: the spin bound of sps_wait_progress (tests force the redo path with it; 0xFFFFFFFF: every wait
 * gives up without looking at the counter; 0 restores the default).  Returns the previous bound.
 * sps_debug_set_exchange_spins: the spin bound of the cross-workgroup polls inside the FPS kernels (the K-way sort of the
 * pre-pass / clustered kernel and the clustered kernel's record exchange).  Same encoding. */
unsigned sps_debug_set_wait_spins(unsigned spins);
unsigned sps_debug_set_exchange_spins(unsigned spins);

#ifdef __cplusplus
}
#endif
#endif /* SPSNET_SA_DEBUG_H */
