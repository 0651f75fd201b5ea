// Issue rate of v_mfma_f32_16x16x4_f32 on gfx950 with TWO accumulators: alternating every MFMA (what a "for k: for nt" loop
// emits), in runs of 4 and of 16 on the same accumulator, and one accumulator alone.  One wave per SIMD on every CU (256 blocks
// x 256 threads); wall time from hipEvents -> cycles per MFMA at 2.4 GHz (nominal: 32).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int RUN, int NACC>
__global__ __launch_bounds__(256) void chain(float *out, int iters) {
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = 0.001f * (threadIdx.x + i); b[i] = 0.002f * (i + 1); }
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32 / RUN; ++u)
#pragma unroll
            for (int r = 0; r < RUN; ++r)
                acc[NACC == 1 ? 0 : (u & 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r & 3], b[(r + u) & 3], acc[NACC == 1 ? 0 : (u & 1)], 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[0][1] + acc[0][2] + acc[0][3] + acc[1][0] + acc[1][1] + acc[1][2] + acc[1][3];
}

template <int RUN, int NACC>
void run(float *out, int blocks, const char *what) {
    const int iters = 2048;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((chain<RUN, NACC>), dim3(blocks), dim3(256), 0, 0, out, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain<RUN, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / (iters * 32.0);
    printf("  %-46s %4d blocks: %7.3f ms  %6.2f ns per MFMA = %5.1f cycles at 2.4 GHz\n", what, blocks, ms, per, per * 2.4);
}

int main() {
    float *out;
    hipMalloc(&out, 1024 * 256 * sizeof(float));
    for (int blocks : {256, 8}) {
        run<32, 1>(out, blocks, "one accumulator");
        run<1, 2>(out, blocks, "two accumulators, alternating every MFMA");
        run<2, 2>(out, blocks, "two accumulators, runs of 2");
        run<4, 2>(out, blocks, "two accumulators, runs of 4");
        run<16, 2>(out, blocks, "two accumulators, runs of 16");
    }
    return 0;
}
