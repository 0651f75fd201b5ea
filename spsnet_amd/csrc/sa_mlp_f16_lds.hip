// sa_mlp_f16_lds.hip -- the split-fp16 grouped-MLP kernel of sa_mlp_f16.hip with the weight stream SHARED by the four
// waves of a workgroup through LDS.
//
// In sa_mlp_f16.hip every wave streams every weight fragment from L2 for its own 32 columns: the widest scale
// (131 -> 128 -> 256 -> 256, 8 x 512 centroids x 32 samples) pulls 456 KiB per 32 columns = 1.9 GB per launch, ~10 TB/s,
// and that L2 -> CU traffic, not the matrix cores, bounds it (18 % of the split-fp16 MFMA rate).  Here a workgroup's four
// waves work on four units in lockstep and consume ONE stream:
//   * the matrix instruction is v_mfma_f32_16x16x32_f16 (K = 32; the K = 16 form of sa_mlp_f16.hip issues at a quarter
//     of its rate on gfx950).  Its B operand wants 8 consecutive k per lane; the D tiles of the previous layer give a
//     lane rows 4q..4q+3 of two 16-row tiles, so k-slot (q, j) of step s is channel 32s + 16(j/4) + 4q + j%4 -- the
//     host packs the weights in that order and activations still chain register to register;
//   * layers 2 and 3 are interleaved: as soon as 32 channels of layer 2 exist (two 16-row tiles = one k32-step of
//     layer 3) they are multiplied into ALL layer-3 output tiles, whose accumulators sit in the otherwise idle AGPRs --
//     the full layer-2 activation (128 VGPRs at 256 channels x 32 columns, which hipcc spilled to scratch) never exists;
//   * the host concatenates the fragments in exactly the order the kernel consumes them: layer 1 as [k32][tile], then
//     for every k32-step s of layer 3: layer-2 tiles 2s, 2s+1 as [tile][k32], layer-3 fragments (tile, s) for all
//     tiles; a fragment is 2 KiB = [hi x8 per lane | lo x8 per lane], a CHUNK = 4 fragments = 8 KiB;
//   * chunk g is brought in by LDS-DMA (global_load_lds_dwordx4: wave w fetches fragment 4g+w, two 1-KiB pieces, no
//     VGPRs) into a ring of RING slots, RING-1 chunks ahead of its use -- the stream simply wraps around from the end
//     of one unit to the start of the next, so the pipeline never drains inside the launch;
//   * per chunk: counted s_waitcnt vmcnt (my pieces of chunk g have landed), ONE raw s_barrier (everybody's have,
//     everybody is done with chunk g-1), refill the slot chunk g-1 occupied, 8 x ds_read_b128, 24 MFMAs.
// L2 traffic drops 4x; LDS carries 2 KiB per wave and fragment.
// Biases sit in LDS for the whole launch and results are written once per unit, so that no ordinary vector load or store
// is outstanding while the counted waits are relied upon (loads and stores retire out of order with each other).
// Arithmetic, operand split and results are those of sa_mlp_f16.hip (tests compare both).
#include "sps_common.h"
#include "sa_mlp_args.h"

namespace sps {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ f32x4 mfma32h(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// Split 4 fp32 values into halves OFF..OFF+3 of the hi / lo operand vectors.  The value is clamped to the fp16 range
// first (RELU: to [0, 65504], which is the layer's ReLU as well), so hi is finite and lo = fp16(c - hi) is tiny: no
// inf/NaN can arise.  |x| <= 65504 splits exactly to 22 bits; `mx` tracks the largest magnitude seen so that the
// launch can report operands beyond that (fused.check_overflow) -- never silent.
template <int OFF, bool RELU>
__device__ __forceinline__ void split4(const f32x4 v, h8 &hi, h8 &lo, float &mx) {
    mx = __builtin_amdgcn_fmed3f(mx, INFINITY, fmaxf(fabsf(v[0]), fabsf(v[1])));  // max3(mx, |v0|, |v1|) for mx >= 0
    mx = __builtin_amdgcn_fmed3f(mx, INFINITY, fmaxf(fabsf(v[2]), fabsf(v[3])));
    if constexpr (!RELU) {   // gathered inputs: NaN / Inf are caught by 0 * v (v_max drops a NaN); see sa_mlp_f16.hip
        const float z = __builtin_fmaf(v[0], 0.f, __builtin_fmaf(v[1], 0.f, __builtin_fmaf(v[2], 0.f, v[3] * 0.f)));
        mx = (z == 0.f) ? mx : INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float c = __builtin_amdgcn_fmed3f(v[r], RELU ? 0.f : -65504.f, 65504.f);
        const _Float16 h = (_Float16)c;
        hi[OFF + r] = h;
        lo[OFF + r] = (_Float16)(c - (float)h);
    }
}

struct WFrag { h8 hi, lo; };


// LDS slots of one chunk (8 KiB) each.  (12 slots = 88 KiB in flight for the one-workgroup-per-CU variants: no change, 370 vs
// 375 us at layer 5 -- the DMA stream is not what they wait for.)
template <int C3> constexpr int ring_slots() { return 6; }
constexpr int FRAG_BYTES = 2048;    // [hi x8 | lo x8] per lane, as two lane-linear 1-KiB pieces
constexpr int CHUNK_BYTES = 4 * FRAG_BYTES;  // a chunk = 4 fragments = 8 pieces of 1 KiB, fetched by 4 or 8 waves

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace

// a.w1 = the concatenated fragment stream; a.ks1 = layer-1 k-steps of 32 grouped channels.
template <int C1, int C2, int C3, int NT, int NS, int WAVES, int PIPE_MODE = 2>
__global__ __launch_bounds__(64 * WAVES, (NT == 1 && C3 <= 256) ? 2 : 1) void sa_group_mlp_f16_lds_kernel(SaMlpArgs a) {
    if (a.run_if && *a.run_if == 0) return;   // workgroup-uniform, before any barrier or DMA request
    constexpr int PIECES = 8 / WAVES;  // 1-KiB pieces of a chunk each wave fetches
    constexpr int RING = ring_slots<C3>();
    static_assert(WAVES == 4 || WAVES == 8, "4 waves x 2 pieces or 8 waves x 1 piece");
    constexpr int T1 = C1 / 16, T2 = C2 / 16, MT3 = C3 / 16;   // 16-row output tiles
    constexpr int S1 = T1 / 2, S2 = T2 / 2;                    // k32-steps over the previous layer's channels
    constexpr int UNIT = 16 * NT;
    constexpr int CPP = UNIT >= NS ? UNIT / NS : 1;    // whole centroids per unit ...
    constexpr int SPLIT = UNIT >= NS ? 1 : NS / UNIT;  // ... or consecutive waves (units) per centroid
    constexpr int BLOCK_FRAGS = 2 * S1 + MT3;                  // fragments of one (layer-2 pair, layer-3 k-step) block
    static_assert((UNIT % NS == 0 || NS % UNIT == 0) && (NS % 16) == 0 && WAVES % SPLIT == 0, "units and centroids must nest");
    static_assert(T1 % 4 == 0 && T2 % 2 == 0 && BLOCK_FRAGS % 4 == 0, "chunks of 4 fragments must tile the stream");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ring = smem;                                        // RING x 8 KiB
    float *bias_l = reinterpret_cast<float *>(smem + RING * CHUNK_BYTES);             // C1 + C2 + c3 floats
    float *stage = bias_l + C1 + C2 + C3;                     // [WAVES][CPP][C3] pooled outputs of the current unit

    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, c = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nchunks = (a.ks1 * T1 + S2 * BLOCK_FRAGS) / 4;  // per unit

    for (int i = threadIdx.x; i < C1; i += 64 * WAVES) bias_l[i] = a.b1[i];
    for (int i = threadIdx.x; i < C2; i += 64 * WAVES) bias_l[C1 + i] = a.b2[i];
    for (int i = threadIdx.x; i < C3; i += 64 * WAVES) bias_l[C1 + C2 + i] = a.b3[i];
    __syncthreads();
    const float *b1l = bias_l, *b2l = bias_l + C1, *b3l = bias_l + C1 + C2;
    float *my_stage = stage + (size_t)wv * CPP * C3;

    // ---- the fragment stream ----
    const char *wsrc = reinterpret_cast<const char *>(a.w1) + (size_t)wv * (PIECES * 1024) + lane * 16;  // my pieces of chunk 0
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)ring;
    int prod = 0, prod_slot = 0, cons_slot = 0;  // next chunk to request / its slot / slot of the next chunk to consume
    auto request = [&]() {
        // LDS-DMA as inline asm: through the builtin hipcc treats the pending LDS write as an alias of every later ring
        // read and drains it (vmcnt(0)) before each one.  M0 = wave-uniform LDS destination; saved and restored.
        const char *gsrc = wsrc + (size_t)prod * CHUNK_BYTES;
        const unsigned dst = ring_base + (unsigned)(prod_slot * CHUNK_BYTES + wv * (PIECES * 1024));
        unsigned keep;
        if constexpr (PIECES == 2) {
            const char *gsrc2 = gsrc + 1024;  // (no instruction offset: it would also move the LDS address)
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %3\n\t"
                         "s_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\t"
                         "s_add_u32 m0, m0, 0x400\n\t"
                         "s_nop 0\n\t"
                         "global_load_lds_dwordx4 %2, off\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(gsrc), "v"(gsrc2), "s"(dst)
                         : "memory", "scc");
        } else {
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %2\n\t"
                         "s_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(gsrc), "s"(dst)
                         : "memory");
        }
        prod = (prod + 1 == nchunks) ? 0 : prod + 1;
        prod_slot = (prod_slot + 1 == RING) ? 0 : prod_slot + 1;
    };
    for (int i = 0; i < RING - 1; ++i) request();
    // PIPE: the ring reads run ONE CHUNK AHEAD of the matrix instructions.  next_chunk hands out the fragments it asked for
    // during the previous call and asks for the following chunk's, which then land while this chunk's MFMAs issue -- with
    // one or two waves per SIMD nobody else hides the LDS latency (~130 cycles + 8 x 4 array cycles per chunk against 12
    // MFMAs = 192 cycles).  Not for the 1024-row variant, whose 511 registers leave no room for the second fragment set.
    // The bias of the layer-2 tile that BEGINS with the next chunk travels with that chunk's reads (a ninth ds_read): read by
    // the compiler it drew an s_waitcnt lgkmcnt(0) right behind the eight reads in flight.  Needs tiles that begin on chunk
    // boundaries: S1 == 4.
    // (the NT = 2 variants would drop to one wave per SIMD; the 512- / 1024-row variants of IA-SSD layer 5, one wave per SIMD
    //  anyway: 86.0 -> 80.9 us and 385 -> 370 us)
    constexpr bool PIPE = PIPE_MODE > 0 && NT == 1 && (S1 == 4 || C3 > 256);
    constexpr bool PIPE_BIAS = PIPE && PIPE_MODE > 1 && S1 == 4;
    const unsigned bias2_base = ring_base + (unsigned)(RING * CHUNK_BYTES + (C1 + 4 * q) * 4);   // &b2l[4 q] as an LDS address
    f32x4 nbias = {0.f, 0.f, 0.f, 0.f}, cur_bias = {0.f, 0.f, 0.f, 0.f};
    i32x4 n0, n1, n2, n3, n4, n5, n6, n7;   // the fragments in flight (PIPE)
    auto issue_reads = [&](int next_tile) {   // next_tile: the layer-2 tile whose k-steps fill the chunk being read, or -1
        if (PIPE_BIAS && next_tile >= 0) {
            const unsigned baddr = bias2_base + (unsigned)(64 * next_tile);
            asm volatile("ds_read_b128 %0, %1" : "=&v"(nbias) : "v"(baddr) : "memory");
        }
        // asm: hipcc hoisted plain LDS loads above the waits and the barrier (it does not treat them as ordered against
        // the asm statements), i.e. before the data was guaranteed to be there
        const unsigned src = ring_base + (unsigned)(cons_slot * CHUNK_BYTES + lane * 16);
        asm volatile("ds_read_b128 %0, %8\n\t"
                     "ds_read_b128 %1, %8 offset:1024\n\t"
                     "ds_read_b128 %2, %8 offset:2048\n\t"
                     "ds_read_b128 %3, %8 offset:3072\n\t"
                     "ds_read_b128 %4, %8 offset:4096\n\t"
                     "ds_read_b128 %5, %8 offset:5120\n\t"
                     "ds_read_b128 %6, %8 offset:6144\n\t"
                     "ds_read_b128 %7, %8 offset:7168"
                     : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3), "=&v"(n4), "=&v"(n5), "=&v"(n6), "=&v"(n7)
                     : "v"(src)
                     : "memory");
        cons_slot = (cons_slot + 1 == RING) ? 0 : cons_slot + 1;
    };
    auto land_reads = [&](WFrag (&w)[4]) {   // the registers are only valid behind this wait: it is their last writer
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6), "+v"(n7), "+v"(nbias)
                     :
                     : "memory");
        cur_bias = nbias;
        union U { i32x4 i; h8 h; };
        U a0{n0}, a1{n1}, a2{n2}, a3{n3}, a4{n4}, a5{n5}, a6{n6}, a7{n7};
        w[0] = WFrag{a0.h, a1.h}; w[1] = WFrag{a2.h, a3.h}; w[2] = WFrag{a4.h, a5.h}; w[3] = WFrag{a6.h, a7.h};
    };
    if constexpr (PIPE) {   // prime: chunk 0 on its way to the registers, RING-1 chunks behind it on their way to the ring
        wait_vm<PIECES * (RING - 2)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_reads(-1);
        request();
    }
    auto next_chunk = [&](WFrag (&w)[4], int next_tile) {
        if constexpr (PIPE) {
            land_reads(w);                          // chunk g: asked for during the previous call; my reads of it are over
            wait_vm<PIECES * (RING - 2)>();         // my pieces of chunk g+1 have landed
            __builtin_amdgcn_s_barrier();           // ... and everybody else's; everybody holds chunk g in registers
            asm volatile("" ::: "memory");
            issue_reads(next_tile);                 // chunk g+1, consumed by the next call
            request();                              // refill the slot chunk g occupied
        } else {
            wait_vm<PIECES * (RING - 2)>();         // my pieces of the chunk about to be read have landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // ... and everybody else's; everybody is done with the previous chunk
            asm volatile("" ::: "memory");
            request();                              // refill the slot the previous chunk occupied
            const unsigned src = ring_base + (unsigned)(cons_slot * CHUNK_BYTES + lane * 16);
            i32x4 r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b128 %0, %8\n\t"
                         "ds_read_b128 %1, %8 offset:1024\n\t"
                         "ds_read_b128 %2, %8 offset:2048\n\t"
                         "ds_read_b128 %3, %8 offset:3072\n\t"
                         "ds_read_b128 %4, %8 offset:4096\n\t"
                         "ds_read_b128 %5, %8 offset:5120\n\t"
                         "ds_read_b128 %6, %8 offset:6144\n\t"
                         "ds_read_b128 %7, %8 offset:7168\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                         : "v"(src)
                         : "memory");
            union U { i32x4 i; h8 h; };
            U a0{r0}, a1{r1}, a2{r2}, a3{r3}, a4{r4}, a5{r5}, a6{r6}, a7{r7};
            w[0] = WFrag{a0.h, a1.h}; w[1] = WFrag{a2.h, a3.h}; w[2] = WFrag{a4.h, a5.h}; w[3] = WFrag{a6.h, a7.h};
            cons_slot = (cons_slot + 1 == RING) ? 0 : cons_slot + 1;
        }
    };
    // hi*hi + hi*lo + lo*hi for all NT column tiles
    auto mac = [&](const WFrag &w, const h8 (&xh)[NT], const h8 (&xl)[NT], f32x4 (&acc)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32h(w.hi, xh[nt], acc[nt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32h(w.hi, xl[nt], acc[nt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32h(w.lo, xh[nt], acc[nt]);
    };

    bool any_bad = false;   // some unit of this wave met an operand beyond the representable range (or NaN / Inf)
    // units are handed out per workgroup so that its four waves stay in lockstep on the shared stream; a wave without a
    // unit of its own recomputes the last one and writes nothing
    const MlpRange rg = mlp_range(a);
    const int ngroups = (rg.units + WAVES - 1) / WAVES;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int unit_raw = grp * WAVES + wv;
        const bool valid = unit_raw < rg.units;
        const int unit = valid ? unit_raw : rg.units - 1;
        const int ub = unit / rg.ups;
        const long long col0 = ((long long)ub * a.m + rg.j0) * NS + (long long)(unit - ub * rg.ups) * UNIT;
        float mx = 0.f;  // largest operand magnitude this lane has split in this unit
        {
            int src[NT];
            long long bj[NT];
            int bb[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const long long e = col0 + nt * 16 + c;
                bj[nt] = e / NS;
                bb[nt] = ub;  // a unit never straddles scenes
                src[nt] = a.idx[e];
            }
            // ---------------- layer 1: k-steps of 32 gathered channels, stream order [k32][tile] ----------------
            f32x4 acc1[T1][NT];
#pragma unroll
            for (int t = 0; t < T1; ++t) {
                const f32x4 bias = *reinterpret_cast<const f32x4 *>(b1l + 16 * t + 4 * q);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc1[t][nt] = bias;
            }
            auto gather4 = [&](int k16, int nt) -> f32x4 {
                f32x4 v;
                if (a.feat_pm) {
                    // point-major features (B, N, C), C % 4 == 0, grouped channel order [features, xyz, pad]: a lane's
                    // four channels are one 16-byte load, the four q-lanes of a column read 64 contiguous bytes
                    const int ch0 = 16 * k16 + 4 * q;
                    if (ch0 < a.c_feat) {
                        v = *reinterpret_cast<const f32x4 *>(a.feat + ((size_t)bb[nt] * a.n + src[nt]) * a.c_feat + ch0);
                    } else {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int ax = ch0 + jj - a.c_feat;
                            v[jj] = ax < 3 ? a.xyz[((size_t)bb[nt] * a.n + src[nt]) * 3 + ax] - a.new_xyz[(size_t)bj[nt] * 3 + ax] : 0.f;
                        }
                    }
                    return v;
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int ch = 16 * k16 + 4 * q + jj;  // grouped channel: 0..2 centred xyz, 3.. features
                    float x;
                    if (ch < 3) {
                        x = a.xyz[((size_t)bb[nt] * a.n + src[nt]) * 3 + ch] - a.new_xyz[(size_t)bj[nt] * 3 + ch];
                    } else if (a.c_feat == 0) {
                        x = 0.f;
                    } else {
                        int cf = ch - 3;
                        cf = cf < a.c_feat ? cf : a.c_feat - 1;  // padded channel: finite data times a zero weight
                        x = a.feat[((size_t)bb[nt] * a.c_feat + cf) * a.n + src[nt]];
                    }
                    v[jj] = x;
                }
                return v;
            };
            f32x4 xcur[2][NT], xnext[2][NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { xcur[0][nt] = gather4(0, nt); xcur[1][nt] = gather4(1, nt); }
            for (int ks = 0; ks < a.ks1; ++ks) {
                const bool more = ks + 1 < a.ks1;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    xnext[0][nt] = more ? gather4(2 * ks + 2, nt) : xcur[0][nt];
                    xnext[1][nt] = more ? gather4(2 * ks + 3, nt) : xcur[1][nt];
                }
                h8 xhi[NT], xlo[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    split4<0, false>(xcur[0][nt], xhi[nt], xlo[nt], mx);
                    split4<4, false>(xcur[1][nt], xhi[nt], xlo[nt], mx);
                }
#pragma unroll
                for (int tc = 0; tc < T1 / 4; ++tc) {
                    WFrag w[4];
                    next_chunk(w, tc == T1 / 4 - 1 ? 0 : -1);   // (behind the last k-step: layer-2 tile 0 is next)
#pragma unroll
                    for (int u = 0; u < 4; ++u) mac(w[u], xhi, xlo, acc1[tc * 4 + u]);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { xcur[0][nt] = xnext[0][nt]; xcur[1][nt] = xnext[1][nt]; }
            }
            h8 h1hi[S1][NT], h1lo[S1][NT];
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (t % 2 == 0) split4<0, true>(acc1[t][nt], h1hi[t / 2][nt], h1lo[t / 2][nt], mx);
                    else split4<4, true>(acc1[t][nt], h1hi[t / 2][nt], h1lo[t / 2][nt], mx);
                }

            // ---------------- layers 2 and 3, interleaved by layer-3 k-step ----------------
            f32x4 acc3[MT3][NT];
#pragma unroll
            for (int mt = 0; mt < MT3; ++mt) {
                const f32x4 bias = *reinterpret_cast<const f32x4 *>(b3l + 16 * mt + 4 * q);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc3[mt][nt] = bias;
            }
#pragma unroll
            for (int s = 0; s < S2; ++s) {
                h8 gh[NT], gl[NT];  // layer-2 channels 32s .. 32s+31 of this unit's columns
                f32x4 acc[NT];
                WFrag w[4];
#pragma unroll
                for (int f = 0; f < BLOCK_FRAGS; ++f) {
                    if (f % 4 == 0) {
                        // what the NEXT chunk holds: the second tile of this block, layer-3 fragments, or the next block's first tile
                        const int nf = f + 4;
                        next_chunk(w, nf < 2 * S1 ? 2 * s + nf / S1 : (nf == BLOCK_FRAGS && s + 1 < S2 ? 2 * s + 2 : -1));
                    }
                    if (f < 2 * S1) {
                        const int half = f / S1, s1 = f % S1, mt = 2 * s + half;
                        if (s1 == 0) {
                            const f32x4 bias = PIPE_BIAS ? cur_bias : *reinterpret_cast<const f32x4 *>(b2l + 16 * mt + 4 * q);
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) acc[nt] = bias;
                        }
                        mac(w[f % 4], h1hi[s1], h1lo[s1], acc);
                        if (s1 == S1 - 1) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                if (half == 0) split4<0, true>(acc[nt], gh[nt], gl[nt], mx);
                                else split4<4, true>(acc[nt], gh[nt], gl[nt], mx);
                            }
                        }
                    } else {
                        mac(w[f % 4], gh, gl, acc3[f - 2 * S1]);
                    }
                }
            }
            // ---------------- ReLU + max-pool over the samples of each centroid (one integer max from +0) ----------------
            // A unit that met an unrepresentable operand (or NaN / Inf inputs) is POISONED: NaN rows, never a clamped value.
            const bool poison = __builtin_amdgcn_ballot_w64(mx > 65504.f) != 0ull;
            any_bad |= poison;
#pragma unroll
            for (int mt = 0; mt < MT3; ++mt) {
                f32x4 best[CPP];
#pragma unroll
                for (int cc = 0; cc < CPP; ++cc) best[cc] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int cc = SPLIT > 1 ? 0 : (nt * 16) / NS;
#pragma unroll
                    for (int r = 0; r < 4; ++r) best[cc][r] = imaxf(best[cc][r], acc3[mt][nt][r]);
                }
#pragma unroll
                for (int cc = 0; cc < CPP; ++cc) {
                    f32x4 v = row_allmax4i(best[cc]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = poison ? __int_as_float(0x7fc00000) : v[r];
                    if (c == 0) *reinterpret_cast<f32x4 *>(my_stage + (size_t)cc * C3 + 16 * mt + 4 * q) = v;
                }
            }
        }
        // ---------------- write the unit's pooled rows; nothing of mine may be outstanding but the weight requests ------
        if constexpr (SPLIT > 1) {
            // a centroid's samples are spread over SPLIT consecutive waves: the first of them combines the partial maxima
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (valid && wv % SPLIT == 0) {
            const long long bj0 = col0 / NS;
            for (int e = lane; e < CPP * a.c3_real; e += 64) {
                const int cc = e / a.c3_real, row = e - cc * a.c3_real;
                const long long cen = bj0 + cc;
                const int b = ub, j = (int)(cen - (long long)ub * a.m);  // no 64-bit division: the unit lies inside scene ub
                float v = my_stage[(size_t)cc * C3 + row];
#pragma unroll
                for (int o = 1; o < SPLIT; ++o) v = imaxf(v, my_stage[(size_t)(o * CPP + cc) * C3 + row]);   // NaN-keeping
                if (a.out_pm) a.out[((size_t)b * a.m + j) * a.out_c_total + a.out_c_off + row] = v;   // lanes <-> rows: contiguous
                else a.out[((size_t)b * a.out_c_total + a.out_c_off + row) * a.m + j] = v;
            }
        }
        if constexpr (SPLIT > 1) {  // the partner's stage is read above: nobody may overwrite it before everybody is here
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        wait_vm<0>();  // stores retire out of order with loads: let them (and the requests) finish before counting again
    }
    wait_vm<0>();
    if (any_bad && a.overflow) *a.overflow = 1;
}

template <int C1, int C2, int C3, int NT, int NS, int WAVES, int PIPE_MODE = 2>
static int launch_lds_variant(const SaMlpArgs &a, hipStream_t st) {
    constexpr int UNIT = 16 * NT;
    SaMlpArgs k = a;
    const long long cols_scene = (long long)a.ups * NS;
    if (cols_scene % UNIT != 0)
        return fail(SPS_ERR_INVALID, "sa_group_mlp(f16/lds): centroids*nsample per scene (%lld) not a multiple of %d", cols_scene, UNIT);
    k.ups = (int)(cols_scene / UNIT);
    k.units = a.units * k.ups;
    k.alt_j0 = 0;
    k.alt_ups = (int)((long long)a.m * NS / UNIT);
    k.alt_units = a.units * k.alt_ups;
    k.ks1 = (3 + a.c_feat + 31) / 32;
    const int groups = divup(k.units, WAVES);
    const int resident = WAVES == 8 ? 256 : 512;   // eight waves = two per SIMD fill a CU
    int blocks = groups < resident ? groups : resident;
    const size_t lds = (size_t)ring_slots<C3>() * CHUNK_BYTES + sizeof(float) * ((size_t)C1 + C2 + C3 + (size_t)WAVES * (UNIT >= NS ? UNIT / NS : 1) * C3);
    static LdsLimitOnce raised;  // one per instantiation
    if (lds > 64 * 1024) {
        const int rc = raise_lds_limit((const void *)sa_group_mlp_f16_lds_kernel<C1, C2, C3, NT, NS, WAVES, PIPE_MODE>, 150 * 1024, raised,
                                       "sa_group_mlp(f16/lds)");
        if (rc != SPS_OK) return rc;
    }
    hipLaunchKernelGGL((sa_group_mlp_f16_lds_kernel<C1, C2, C3, NT, NS, WAVES, PIPE_MODE>), dim3(blocks), dim3(64 * WAVES), lds, st, k);
    return check_launch("sa_group_mlp_f16_lds_kernel");
}

// split_fp16 == 2: the weights are ONE concatenated stream in a.w1 (fused._pack_stream)
int launch_sa_mlp_f16_lds(const SaMlpArgs &a, int c1, int c2, int nsample, hipStream_t st) {
#define SPS_MLPL_CASE(C1, C2, C3, NT, NS, W) \
    if (c1 == C1 && c2 == C2 && a.c3 == C3 && nsample == NS) return launch_lds_variant<C1, C2, C3, NT, NS, W>(a, st);
    SPS_MLPL_CASE(64, 64, 128, 2, 16, 4)
    SPS_MLPL_CASE(64, 96, 128, 2, 32, 4)
    SPS_MLPL_CASE(128, 128, 256, 1, 16, 4)
    SPS_MLPL_CASE(128, 256, 256, 1, 32, 4)   // (NT = 2 here and one line up: 512 registers + spills, 2.30 -> 2.35 ms per pass;
                                             //  8 waves per workgroup = half the L2 stream: 111.6 -> 112.4 us, not the bound;
                                             //  ring reads one chunk ahead: 114 -> 108 us, PIPE_MODE 0 / 2 on one box)
    SPS_MLPL_CASE(256, 256, 512, 1, 16, 4)   // IA-SSD layer 5 [259,256,256,512]: 4 waves = 512 registers each
    SPS_MLPL_CASE(256, 512, 1024, 1, 32, 4)  // IA-SSD layer 5 [259,256,512,1024]
    SPS_MLPL_CASE(128, 128, 256, 1, 64, 8)   // nsample 64: a centroid spans four waves
    SPS_MLPL_CASE(128, 256, 256, 1, 64, 8)
#undef SPS_MLPL_CASE
    return fail(SPS_ERR_INVALID, "sa_group_mlp(f16/lds): no kernel for widths (%d, %d, %d) nsample %d", c1, c2, a.c3, nsample);
}

}  // namespace sps
