// microbench_valu2.hip -- more gfx950 VALU issue rates (integer min/max, compares, selects, DPP, readlane)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
#define OPS8(ins) ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8\n"
#define OPS8_3(ins) ins " %0, %0, %8, %9\n" ins " %1, %1, %8, %9\n" ins " %2, %2, %8, %9\n" ins " %3, %3, %8, %9\n" ins " %4, %4, %8, %9\n" ins " %5, %5, %8, %9\n" ins " %6, %6, %8, %9\n" ins " %7, %7, %8, %9\n"
#define REGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)

template <int OP>
__global__ void k(unsigned long long *out, int iters) {
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int b = 12345, c = 777;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile(OPS8("v_min_u32") REGS);) }
        else if (OP == 1) { REP16(asm volatile(OPS8("v_max_i32") REGS);) }
        else if (OP == 2) { REP16(asm volatile(OPS8("v_max_f32") REGS);) }
        else if (OP == 3) { REP16(asm volatile(OPS8("v_add_u32") REGS);) }
        else if (OP == 4) { REP16(asm volatile(OPS8("v_and_b32") REGS);) }
        else if (OP == 5) { REP16(asm volatile(OPS8_3("v_max3_i32") REGS);) }
        else if (OP == 6) { REP16(asm volatile(OPS8_3("v_min3_u32") REGS);) }
        else if (OP == 7) { REP16(asm volatile(OPS8_3("v_lshl_or_b32") REGS);) }
        else if (OP == 8) { REP16(asm volatile(OPS8_3("v_and_or_b32") REGS);) }
        else if (OP == 9) { REP16(asm volatile(OPS8("v_mul_f32") REGS);) }
        else if (OP == 10) { REP16(asm volatile(OPS8("v_add_f32") REGS);) }
        else if (OP == 11) { REP16(asm volatile(OPS8_3("v_med3_f32") REGS);) }
        else if (OP == 12) {  // compare only (VOPC to vcc)
            REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n"
                               "v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n" REGS : "vcc");)
        } else if (OP == 13) {  // select only
            REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                               "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n" REGS : "vcc");)
        } else if (OP == 14) {  // dependent DPP max chain (latency): v_max_i32_dpp row_shr:1
            REP16(asm volatile("v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                               "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                               "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
                               "s_nop 1\n s_nop 1\n" REGS);)
        } else if (OP == 15) {  // v_mov_b32 independent
            REP16(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n" REGS);)
        } else if (OP == 16) {  // v_cmp_eq_u32 to SGPR pair + s_ff1 (VOP3 compare with scalar destination)
            REP16(asm volatile("v_cmp_eq_u32 s[40:41], %0, %8\n v_cmp_eq_u32 s[42:43], %1, %8\n v_cmp_eq_u32 s[44:45], %2, %8\n v_cmp_eq_u32 s[46:47], %3, %8\n"
                               "v_cmp_eq_u32 s[48:49], %4, %8\n v_cmp_eq_u32 s[50:51], %5, %8\n v_cmp_eq_u32 s[52:53], %6, %8\n v_cmp_eq_u32 s[54:55], %7, %8\n" REGS
                               : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");)
        } else if (OP == 17) { REP16(asm volatile(OPS8("v_fmac_f32") REGS);) }
        else if (OP == 18) { REP16(asm volatile(OPS8("v_sub_u32") REGS);) }
        else if (OP == 19) { REP16(asm volatile(OPS8("v_lshlrev_b32") REGS);) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123456789) out[1000] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char *name, unsigned long long *d) {
    const int iters = 200;
    const int instr = iters * 16 * 8;
    printf("%-16s", name);
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(16);
        (void)hipMemcpy(h.data(), d, 16 * 8, hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
        printf("  %dw/SIMD: %.2f cyc/instr/SIMD", threads / 256, (double)mx / instr / (threads / 256.0));
    }
    printf("\n");
}

int main() {
    unsigned long long *d;
    (void)hipMalloc(&d, 4096 * 8);
    run<0>("v_min_u32", d); run<1>("v_max_i32", d); run<2>("v_max_f32", d); run<3>("v_add_u32", d); run<4>("v_and_b32", d);
    run<5>("v_max3_i32", d); run<6>("v_min3_u32", d); run<7>("v_lshl_or_b32", d); run<8>("v_and_or_b32", d);
    run<9>("v_mul_f32", d); run<10>("v_add_f32", d); run<11>("v_med3_f32", d); run<12>("v_cmp_gt_f32", d);
    run<13>("v_cndmask_b32", d); run<14>("dpp6chain(8)", d); run<15>("v_mov_b32", d); run<16>("v_cmp->sgpr", d);
    run<17>("v_fmac_f32", d); run<18>("v_sub_u32", d); run<19>("v_lshlrev_b32", d);
    return 0;
}
