// microbench_valu.hip -- VALU issue rates on gfx950 that decide the FPS inner-loop design:
// v_fma_f32 vs v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, v_min_f32, v_max3_f32, v_cndmask,
// at 1, 2 and 4 waves per SIMD (256 / 512 / 1024-thread workgroups on one CU).
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_valu.hip -o tools/microbench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ void k(unsigned long long *out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {  // v_fma_f32, 8 independent chains
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (OP == 1) {  // v_pk_fma_f32, 4 independent chains (8 floats)
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        } else if (OP == 2) {  // v_pk_mul_f32
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                               "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));)
        } else if (OP == 3) {  // v_pk_add_f32
            REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                               "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));)
        } else if (OP == 4) {  // v_min_f32
            REP16(asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                               "v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (OP == 5) {  // v_max3_f32
            REP16(asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                               "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (OP == 6) {  // v_cmp_gt_f32 + v_cndmask pair (the argmax tracking idiom)
            REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                               "v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if (OP == 7) {  // v_sub_f32 with an SGPR operand
            REP16(asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                               "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (s == 123.456f) out[1000] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char *name, unsigned long long *d) {
    const int iters = 200;
    const int instr = iters * 16 * 8;
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(16);
        hipMemcpy(h.data(), d, 16 * 8, hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
        const double waves_per_simd = threads / 64 >= 4 ? threads / 256.0 : 1.0;
        // s_memtime counts at 100 MHz-independent "shader cycle"? report both raw ticks and per-instr
        printf("%-14s threads=%4d  ticks=%8llu  ticks/instr/wave=%.3f  ticks per instr per SIMD=%.3f\n", name, threads, mx,
               (double)mx / instr, (double)mx / instr / waves_per_simd);
    }
}

int main() {
    unsigned long long *d;
    hipMalloc(&d, 4096 * 8);
    run<0>("v_fma_f32", d);
    run<1>("v_pk_fma_f32", d);
    run<2>("v_pk_mul_f32", d);
    run<3>("v_pk_add_f32", d);
    run<4>("v_min_f32", d);
    run<5>("v_max3_f32", d);
    run<6>("cmp+cndmask", d);
    run<7>("v_sub_f32", d);
    return 0;
}
