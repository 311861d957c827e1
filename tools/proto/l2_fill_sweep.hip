// l2_fill_sweep.hip — VERDICT r5 item 6: why does a CU fill at ~44 B/clk when every CU reads the SAME lines and at ~12 B/clk when each
// re-reads a private tile (profiles/r02_lds_fill_rate.txt)?  One workgroup per CU (8 waves, global_load_dwordx4 of 128-byte row pieces,
// k_flat_gemm2's pattern); the tile a workgroup re-reads is shared by `share` workgroups of its XCD and is `rows` x 1536 B large.
//   footprint per XCD = (32 / share) x rows x 1536 B   — below / above the XCD's 4 MiB L2
//   share = 1 .. 32: how many CUs of the XCD want the same line (not in lock-step: they run free)
// Prints bytes per clock per CU at 2.4 GHz.  Build: hipcc --offload-arch=gfx950 -O3 -o l2_fill_sweep l2_fill_sweep.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_fill(const char* base, int rows, int share, int ksteps, uint32_t* out) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = 1536;
    const int x = blockIdx.x & 7, s = blockIdx.x >> 3;
    const int tiles_per_xcd = 32 / share;
    const size_t tile_bytes = (size_t)rows * ld;
    const char* wg_base = base + ((size_t)x * tiles_per_xcd + (s / share)) * tile_bytes;
    // a stage = 512 row pieces of 128 B (like two 256-row operands): wave w takes pieces [64 w, 64 w + 64), 8 per instruction
    uint32_t acc = 0;
    int kt = 0, r0 = 0;
    v4u r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = v4u{0, 0, 0, 0};
    for (int g = 0; g < ksteps; g++) {
        v4u n[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int R = r0 + (8 * w + j) * 8 + (lane >> 3);
            R = R % rows;
            n[j] = *reinterpret_cast<const v4u*>(wg_base + (size_t)R * ld + kt * 128 + (lane & 7) * 16);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc ^= r[j].x;
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = n[j];
        __builtin_amdgcn_s_barrier();
        if (++kt == 12) { kt = 0; r0 = (r0 + 512) % rows; }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) acc ^= r[j].y;
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t total = (size_t)256 * 4096 * 1536;          // room for 256 private tiles of 4096 rows
    char* buf; CK(hipMalloc(&buf, total));
    CK(hipMemset(buf, 1, total));
    uint32_t* dout; CK(hipMalloc(&dout, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("rows  tile KiB  share  footprint per XCD (MiB)   B/clk/CU   TB/s chip\n");
    const int ksteps = 3000;
    for (int rows : {32, 64, 128, 256, 512, 1024, 4096})
        for (int share : {1, 2, 4, 8, 32}) {
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_fill, dim3(256), dim3(512), 0, 0, buf, rows, share, ksteps, dout);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }
            const double bytes_cu = (double)ksteps * 65536;
            printf("%5d  %7.0f  %5d  %10.2f  %22.1f  %8.2f\n", rows, rows * 1536 / 1024.0, share, (32.0 / share) * rows * 1536 / 1048576.0,
                   bytes_cu / (best * 1e-3 * 2.4e9), bytes_cu * 256 / (best * 1e-3) / 1e12);
            fflush(stdout);
        }
    return 0;
}
