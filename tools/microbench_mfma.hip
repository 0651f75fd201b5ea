// Issue rate of v_mfma_f32_16x16x32_f16 chains on gfx950: the same accumulator back to back (what a K loop over one output
// tile does) against 2, 3, 4 and 8 accumulators in rotation.  One wave per SIMD (256 blocks x 256 threads) and one wave
// alone; cycles from s_memtime (100 MHz reference -> reported as ns per MFMA) and wall time from hipEvents.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void chain(float *out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
    f32x4 acc[NACC];
    for (int k = 0; k < NACC; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24 / NACC; ++u)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(float *out, int blocks) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(chain<NACC>, dim3(blocks), dim3(256), 0, 0, out, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / (iters * 24.0);   // ns per MFMA per wave (one wave per SIMD)
    printf("  %d accumulator(s) in rotation, %4d blocks: %7.3f ms  %6.2f ns per MFMA = %5.1f cycles at 2.4 GHz\n", NACC, blocks, ms, per, per * 2.4);
}

int main() {
    float *out;
    hipMalloc(&out, 1024 * 256 * sizeof(float));
    for (int blocks : {256, 512}) {
        run<1>(out, blocks); run<2>(out, blocks); run<3>(out, blocks); run<4>(out, blocks); run<8>(out, blocks);
    }
    return 0;
}
