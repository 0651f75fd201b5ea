// agpr_idx_probe.hip -- does VGPR-index mode (s_set_gpr_idx_on) apply to the ACC-register operand of v_accvgpr_read_b32 /
// v_accvgpr_write_b32 on gfx950?  (Question behind the "one wave per SIMD, upper bucket slots in AGPRs" FPS variant of
// a wave that holds 64 bucket slots needs x, y, z beyond the 256 architectural VGPRs, selected by a wave-uniform
// runtime index.)   build: hipcc --offload-arch=gfx950 -O2 -o tools/agpr_idx_probe tools/agpr_idx_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(int *out, int idx) {
    int lane = threadIdx.x;
    int r_plain, r_src, r_dst;
    asm volatile(
        "v_add_u32 %0, 100, %3\n\t v_accvgpr_write_b32 a0, %0\n\t"
        "v_add_u32 %0, 101, %3\n\t v_accvgpr_write_b32 a1, %0\n\t"
        "v_add_u32 %0, 102, %3\n\t v_accvgpr_write_b32 a2, %0\n\t"
        "v_add_u32 %0, 103, %3\n\t v_accvgpr_write_b32 a3, %0\n\t"
        "v_add_u32 %0, 104, %3\n\t v_accvgpr_write_b32 a4, %0\n\t"
        "v_add_u32 %0, 105, %3\n\t v_accvgpr_write_b32 a5, %0\n\t"
        "v_add_u32 %0, 106, %3\n\t v_accvgpr_write_b32 a6, %0\n\t"
        "v_add_u32 %0, 107, %3\n\t v_accvgpr_write_b32 a7, %0\n\t"
        "s_nop 4\n\t"
        "v_accvgpr_read_b32 %0, a0\n\t"                      // plain: 100 + lane
        "s_set_gpr_idx_on %4, gpr_idx(SRC0)\n\t"
        "v_accvgpr_read_b32 %1, a0\n\t"                      // indexed source: 100 + idx + lane if the mode applies
        "s_set_gpr_idx_off\n\t"
        "v_mov_b32 %2, 999\n\t"
        "s_set_gpr_idx_on %4, gpr_idx(DST)\n\t"
        "v_accvgpr_write_b32 a0, %2\n\t"                     // indexed destination: a[idx] = 999 if the mode applies
        "s_set_gpr_idx_off\n\t"
        "s_nop 4\n\t"
        "v_accvgpr_read_b32 %2, a3\n\t"                      // (idx = 3 in the run below)
        : "=&v"(r_plain), "=&v"(r_src), "=&v"(r_dst)
        : "v"(lane), "s"(idx)
        : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "memory");
    out[lane * 3 + 0] = r_plain;
    out[lane * 3 + 1] = r_src;
    out[lane * 3 + 2] = r_dst;
}

int main() {
    int *d = nullptr, h[192];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) { printf("no device\n"); return 1; }
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 3);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 1; }
    printf("lane 0: plain read a0 = %d (want 100); indexed-source read (idx 3) = %d (103 if the mode applies to ACC sources, 100 if not); "
           "a3 after an indexed-destination write of 999 to a0 = %d (999 if the mode applies to ACC destinations, 103 if not)\n",
           h[0], h[1], h[2]);
    printf("lane 5: %d %d %d\n", h[15], h[16], h[17]);
    return 0;
}
