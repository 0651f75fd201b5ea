// microbench_valu_pk.hip -- gfx950 issue rate of the PACKED fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32:
// two IEEE fp32 operations per lane and instruction, the same rounding as their scalar forms) beside v_add_f32 / v_fma_f32, for a
// pure VALU stream with 1, 2 and 4 waves per SIMD.  Question: do they issue at the full VALU rate, i.e. do
// they halve the instruction count of the FPS kernels' sub / mul / fma chains for free?  (hipcc already emits them where SLP
// vectorisation finds pairs: the accept phase of fps_pruned_kernel holds v_pk_mul_f32 / v_pk_fma_f32.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
typedef float f2 __attribute__((ext_vector_type(2)));
#define PK8(ins) ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8\n"
#define PK8_3(ins) ins " %0, %0, %8, %0\n" ins " %1, %1, %8, %1\n" ins " %2, %2, %8, %2\n" ins " %3, %3, %8, %3\n" ins " %4, %4, %8, %4\n" ins " %5, %5, %8, %5\n" ins " %6, %6, %8, %6\n" ins " %7, %7, %8, %7\n"
#define REGS2 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)

template <int OP>
__global__ void k(unsigned long long *out, int iters) {
    f2 a0 = {1.f + threadIdx.x, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 b = {1.0000001f, 0.9999999f};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile(PK8("v_pk_add_f32") REGS2);) }
        else if (OP == 1) { REP16(asm volatile(PK8("v_pk_mul_f32") REGS2);) }
        else if (OP == 2) { REP16(asm volatile(PK8_3("v_pk_fma_f32") REGS2);) }
        else if (OP == 3) {   // scalar forms on the low halves of the same register pairs (the baseline on this harness)
            REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                               "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                               : "+v"(a0[0]), "+v"(a1[0]), "+v"(a2[0]), "+v"(a3[0]), "+v"(a4[0]), "+v"(a5[0]), "+v"(a6[0]), "+v"(a7[0]) : "v"(b[0]));)
        } else if (OP == 4) {
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                               "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                               : "+v"(a0[0]), "+v"(a1[0]), "+v"(a2[0]), "+v"(a3[0]), "+v"(a4[0]), "+v"(a5[0]), "+v"(a6[0]), "+v"(a7[0]) : "v"(b[0]));)
        } else if (OP == 5) {   // a DEPENDENT chain of v_pk_fma_f32 on one register pair (latency of the packed form)
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %0, %0, %8, %0\n"
                               "v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %0, %0, %8, %0\n" REGS2);)
        } else if (OP == 6) {   // ... and of v_fma_f32
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n"
                               "v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n"
                               : "+v"(a0[0]) : "v"(b[0]));)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s[0] + s[1] == 123456789.f) out[1000] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char *name, unsigned long long *d) {
    const int iters = 200;
    const int instr = iters * 16 * 8;
    printf("%-22s", name);
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(16);
        (void)hipMemcpy(h.data(), d, 16 * 8, hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
        // (the same figure as tools/microbench_valu2.hip prints: s_memtime ticks per instruction and SIMD)
        printf("  %dw/SIMD: %.2f cyc/instr/SIMD", threads / 256, (double)mx / instr / (threads / 256.0));
    }
    printf("\n");
}

int main() {
    unsigned long long *d;
    (void)hipMalloc(&d, 4096 * 8);
    run<3>("v_add_f32", d); run<0>("v_pk_add_f32", d); run<1>("v_pk_mul_f32", d);
    run<4>("v_fma_f32", d); run<2>("v_pk_fma_f32", d);
    run<6>("v_fma_f32 chain", d); run<5>("v_pk_fma_f32 chain", d);
    return 0;
}
