// lds_gather_rate.hip — how many LDS cycles does a wave-level ds_read_b32 cost on gfx950 when the 64 lanes hit
// (a) 32 distinct banks in the SAME 128-byte row, (b) 32 distinct banks in DIFFERENT rows (the rotated PQ table
// gather), (c) random banks?  Pure LDS stream: 16 ds_read_b32 per s_waitcnt, nothing else in the loop.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_lds(int iters, uint32_t* out) {
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int e = tid; e < 32768; e += WAVES * 64) lds[e] = e;
    __syncthreads();
    uint32_t a[16];
    uint32_t h = tid * 2654435761u + 12345u;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        h = h * 1664525u + 1013904223u;
        const uint32_t row = (h >> 8) & 255u;          // 256 rows of 128 B... 512 B apart
        if (MODE == 0) a[k] = (uint32_t)(k * 256 + (lane & 31) * 4);                  // same row, distinct banks
        if (MODE == 1) a[k] = row * 256 + (uint32_t)((lane + k) & 31) * 4;            // random rows, distinct banks
        if (MODE == 2) a[k] = row * 256 + ((h >> 20) & 31u) * 4;                      // random rows, random banks
        if (MODE == 3) a[k] = row * 512 + (uint32_t)((lane + k) & 31) * 8;            // b64: random rows, distinct bank pairs
    }
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t r[16];
        if (MODE != 3) {
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("ds_read_b32 %0, %1" : "=v"(r[k]) : "v"(a[k]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            uint64_t r2[16];
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("ds_read_b64 %0, %1" : "=v"(r2[k]) : "v"(a[k]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 16; k++) r[k] = (uint32_t)r2[k];
        }
        acc ^= r[it & 15];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// overlap test: per iteration 16 conflict-free gathers (L), 16 independent v_perm (V), 4 independent i8 MFMAs (X)
typedef int v4i __attribute__((ext_vector_type(4)));
template <int WHAT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mix(int iters, uint32_t* out) {
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int e = tid; e < 32768; e += WAVES * 64) lds[e] = e;
    __syncthreads();
    uint32_t a[16];
    uint32_t h = tid * 2654435761u + 12345u;
#pragma unroll
    for (int k = 0; k < 16; k++) { h = h * 1664525u + 1013904223u; a[k] = ((h >> 8) & 255u) * 256 + (uint32_t)((lane + k) & 31) * 4; }
    uint32_t p[16];
#pragma unroll
    for (int k = 0; k < 16; k++) p[k] = h + k;
    v4i C0 = {0, 0, 0, 0}, C1 = C0, C2 = C0, C3 = C0;
    const v4i Bm = {1, 1, 1, 1};
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t r[16];
        if (WHAT & 1) {
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("ds_read_b32 %0, %1" : "=v"(r[k]) : "v"(a[k]));
        }
        if (WHAT & 2) {
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(p[k]) : "v"(p[k]), "v"(a[k]), "s"(0x07020500 + it));
        }
        if (WHAT & 4) {
            const v4i A = {(int)p[0], (int)p[1], (int)p[2], (int)p[3]};
            asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(C0) : "v"(A), "v"(Bm));
            asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(C1) : "v"(A), "v"(Bm));
            asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(C2) : "v"(A), "v"(Bm));
            asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(C3) : "v"(A), "v"(Bm));
        }
        if (WHAT & 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc ^= r[it & 15]; }
    }
    acc ^= p[3] ^ (uint32_t)(C0[0] + C1[1] + C2[2] + C3[3]);
    if (acc == 0x12345678u) out[0] = acc;
}
template <int WHAT, int WAVES>
static void runmix(const char* nm, uint32_t* dout) {
    const int iters = 4000, grid = 256 * 4;
    CK(hipFuncSetAttribute((const void*)k_mix<WHAT, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_mix<WHAT, WAVES>), dim3(grid), dim3(WAVES * 64), 131072, 0, iters, dout);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    printf("%-44s waves/CU %2d: %.3f ms  -> %.1f clk per iteration per SIMD-wave-slot @2.4GHz\n", nm, WAVES, best,
           best * 1e-3 * 2.4e9 / ((double)grid / 256 * iters) / (WAVES / 4));
}

template <int MODE, int WAVES>
static void run(const char* nm, uint32_t* dout) {
    const int iters = 4000, grid = 256 * 4;
    size_t shm = MODE == 3 ? 131072 : 65536 + 1024;
    CK(hipFuncSetAttribute((const void*)k_lds<MODE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_lds<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), shm > 131072 ? 131072 : 131072, 0, iters, dout);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    const double instr_per_cu = (double)grid / 256 * WAVES * iters * 16;
    printf("%-44s waves/CU %2d: %.3f ms  -> %.2f clk per wave-level read per CU @2.4GHz\n", nm, WAVES, best, best * 1e-3 * 2.4e9 / instr_per_cu);
}

int main() {
    uint32_t* dout; CK(hipMalloc(&dout, 64));
    run<0, 16>("b32 same row, distinct banks", dout);
    run<1, 16>("b32 random rows, distinct banks (rotated)", dout);
    run<2, 16>("b32 random rows, random banks", dout);
    run<3, 16>("b64 random rows, distinct bank pairs", dout);
    run<1, 8>("b32 random rows, distinct banks (rotated)", dout);
    run<1, 4>("b32 random rows, distinct banks (rotated)", dout);
    run<2, 8>("b32 random rows, random banks", dout);
    runmix<1, 16>("mix: 16 gathers", dout);
    runmix<2, 16>("mix: 16 v_perm", dout);
    runmix<4, 16>("mix: 4 mfma", dout);
    runmix<3, 16>("mix: 16 gathers + 16 v_perm", dout);
    runmix<5, 16>("mix: 16 gathers + 4 mfma", dout);
    runmix<6, 16>("mix: 16 v_perm + 4 mfma", dout);
    runmix<7, 16>("mix: 16 gathers + 16 v_perm + 4 mfma", dout);
    return 0;
}
