// valu_rate.hip — issue cost (clk per wave64 instruction per SIMD) of candidate address-forming VALU ops on gfx950,
// 4 waves per SIMD, 16 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(1024) void k_valu(int iters, uint32_t* out) {
    const int tid = threadIdx.x;
    uint32_t p[16], a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { p[k] = tid * 7 + k; a[k] = tid * 13 + 5 * k; }
    uint32_t c8 = 8;
    asm volatile("" : "+v"(c8));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (OP == 0) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(a[k]), "s"(0x07020500 + it));
            if (OP == 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(a[k]), "v"(c8));
            if (OP == 2) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(p[k]) : "v"(a[k]));
            if (OP == 3) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(a[k]), "v"(c8));
            if (OP == 4) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(c8), "v"(a[k]));
            if (OP == 5) asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(p[k]) : "v"(a[k]));
            if (OP == 6) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(p[k]) : "v"(c8), "v"(a[k]));
            if (OP == 7) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(p[k]) : "v"(a[k]));
            if (OP == 8) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(c8), "v"(a[k]));
            if (OP == 9) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(p[k]) : "v"(a[k]), "v"(c8));
            if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(p[k]) : "v"(a[k]));
            if (OP == 11) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(p[k]) : "v"(c8), "v"(a[k]));
            if (OP == 12) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(a[k]), "v"(c8));
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) acc ^= p[k];
    if (acc == 0x12345678u) out[0] = acc;
}
template <int OP>
static void run(const char* nm, uint32_t* dout) {
    const int iters = 4000, grid = 256 * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_valu<OP>), dim3(grid), dim3(1024), 0, 0, iters, dout);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    // per SIMD: grid/256 WGs x 4 waves x iters x 16 instr
    printf("%-52s %.3f ms  -> %.2f clk per instruction per SIMD @2.4GHz\n", nm, best, best * 1e-3 * 2.4e9 / ((double)grid / 256 * 4 * iters * 16));
}
int main() {
    uint32_t* dout; CK(hipMalloc(&dout, 64));
    run<0>("v_perm_b32 (sgpr selector)", dout);
    run<1>("v_perm_b32 (vgpr selector)", dout);
    run<2>("v_mov_b32_sdwa dst BYTE_1 preserve", dout);
    run<3>("v_and_or_b32", dout);
    run<4>("v_lshl_or_b32", dout);
    run<5>("v_or_b32 (VOP2)", dout);
    run<6>("v_lshlrev_b32_sdwa src byte", dout);
    run<7>("v_add_u32 (VOP2)", dout);
    run<8>("v_bfi_b32", dout);
    run<9>("v_or_b32_sdwa src0 byte", dout);
    run<10>("v_fma_f32", dout);
    run<11>("v_mad_u32_u24", dout);
    run<12>("v_alignbyte_b32", dout);
    return 0;
}
