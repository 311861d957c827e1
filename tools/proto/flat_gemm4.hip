// flat_gemm4.hip — prototype: k_flat_gemm2's tile (256 queries x 256 db rows per workgroup, 8 waves = 2 x 4, 128 x 64 per wave, 128 fp32
// accumulators per lane) with BK = 32 and FOUR 32 KiB LDS stages, THREE of them in flight (96 KiB of L2 -> LDS requests outstanding per CU
// instead of at most 64 KiB part of the time).  Why: profiles/r02_flat_gemm_experiments.md — the fill stream alone takes 75 % of
// k_flat_gemm2 at ~16 B/clk/CU, i.e. it is bound by bytes in flight / L2 latency, not by the L2's bandwidth.
// build: hipcc --offload-arch=gfx950 -O3 -o flat_gemm4 flat_gemm4.hip ;  run: ./flat_gemm4 [n_rows] [check] [variant]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define STAGE 32768            // A 256 x 64 B | B 256 x 64 B
#ifndef NSTAGE
#define NSTAGE 4
#endif
#define AHEAD (NSTAGE - 1)     // stages in flight

__device__ __forceinline__ void dma16(const void* g, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

template <bool STORE>
__global__ __launch_bounds__(512) void k_fg4(const __half* __restrict__ Q, const __half* __restrict__ X, int64_t nv, int ld, int qt,
                                             int64_t ntiles, float* out, int64_t ostride, float thr, unsigned long long* cnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    // walking map: workgroup b on XCD b % 8, slot s = b / 8 keeps query tile s % qt and walks db tiles x + 8 (s / qt) + 8 (S / qt) n
    const int x = blockIdx.x & 7, s = blockIdx.x >> 3, ngq = (int)(gridDim.x >> 3) / qt;
    const int qi = s % qt;
    const int64_t t0 = x + 8 * (s / qt), tstep = 8 * (int64_t)ngq;
    if (t0 >= ntiles) return;
    const int nitems = (int)((ntiles - 1 - t0) / tstep) + 1;
    const int KT = ld / 32;
    const int G = nitems * KT;

    // DMA: a wave instruction moves 16 rows x 64 B.  A: instructions 2w, 2w + 1; B: 2w, 2w + 1.  lane j: row j >> 2, LDS chunk j & 3;
    // the global chunk is XOR-swizzled so that the fragment reads below are conflict-free
    const int jr = lane >> 2, jc = lane & 3;
    const char* srcA[2]; int rowB[2]; int cB[2];
    uint32_t ldsA[2], ldsB[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int R = (2 * w + i) * 16 + jr;
        const int c = jc ^ ((R >> 1) & 3);
        srcA[i] = reinterpret_cast<const char*>(Q + ((int64_t)qi * 256 + R) * ld) + c * 16;
        ldsA[i] = (uint32_t)((2 * w + i) * 1024);
        rowB[i] = R; cB[i] = c * 16;
        ldsB[i] = (uint32_t)(16384 + (2 * w + i) * 1024);
    }
    const int64_t vlast = nv - 1;
    int i_kt = 0; int64_t i_vt = t0 * 256;
    auto issue = [&](int buf) {
        const uint32_t base = (uint32_t)(buf * STAGE);
#pragma unroll
        for (int i = 0; i < 2; i++) dma16(srcA[i] + (int64_t)i_kt * 64, base + ldsA[i]);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            int64_t r = i_vt + rowB[i]; r = r > vlast ? vlast : r;
            dma16(reinterpret_cast<const char*>(X) + r * (int64_t)ld * 2 + (int64_t)i_kt * 64 + cB[i], base + ldsB[i]);
        }
        i_kt++;
        if (i_kt == KT) { i_kt = 0; i_vt += tstep * 256; }
    };

    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

    // fragment of a 32 x 32 x 16 MFMA: lane (li = lane & 31, kh = lane >> 5) holds row li, k = 8 kh .. 8 kh + 7 of the 16: chunk 2 ss + kh
    const int li = lane & 31, kh = lane >> 5;
    int offs[2];
#pragma unroll
    for (int ss = 0; ss < 2; ss++) offs[ss] = li * 64 + (((2 * ss + kh) ^ ((li >> 1) & 3)) << 4);

#pragma unroll
    for (int p = 0; p < AHEAD; p++) if (p < G) issue(p);
    int kt = 0;
    int64_t vt = t0 * 256;
    for (int g = 0; g < G; g++) {
        // my pieces of stage g have landed: the stages g + 1 .. g + AHEAD - 1 issued after it (4 DMAs each) may still be in flight
        const int younger = (G - 1 - g) < (AHEAD - 1) ? (G - 1 - g) : (AHEAD - 1);
        if (younger >= 2) wait_vm<8>(); else if (younger == 1) wait_vm<4>(); else wait_vm<0>();
        __syncthreads();                         // stage g complete for everyone; everyone is done with stage g - 1
        if (g + AHEAD < G) issue((g + AHEAD) % NSTAGE);
        const unsigned char* As = smem + (g % NSTAGE) * STAGE + wr * (128 * 64);
        const unsigned char* Bs = smem + (g % NSTAGE) * STAGE + 16384 + wc * (64 * 64);
        half8 fa[2][4], fb[2][2];
#pragma unroll
        for (int ss = 0; ss < 2; ss++) {
#pragma unroll
            for (int t = 0; t < 4; t++) fa[ss][t] = *reinterpret_cast<const half8*>(As + t * 2048 + offs[ss]);
#pragma unroll
            for (int t = 0; t < 2; t++) fb[ss][t] = *reinterpret_cast<const half8*>(Bs + t * 2048 + offs[ss]);
        }
#pragma unroll
        for (int ss = 0; ss < 2; ss++)
#pragma unroll
            for (int ti = 0; ti < 4; ti++)
#pragma unroll
                for (int tj = 0; tj < 2; tj++)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ss][ti], fb[ss][tj], acc[ti][tj], 0, 0, 0);
        if (++kt < KT) continue;
        kt = 0;
        // ---- epilogue of a (query tile, db tile) item
        const int lj = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int tj = 0; tj < 2; tj++) {
            const int64_t col = vt + wc * 64 + tj * 32 + lj;
#pragma unroll
            for (int ti = 0; ti < 4; ti++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int ql = wr * 128 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float v = acc[ti][tj][r];
                    if (STORE) { if (col < nv) out[((int64_t)qi * 256 + ql) * ostride + col] = v; }
                    else if (__any(v >= thr)) { if (v >= thr && col < nv) atomicAdd(cnt, 1ull); }
                    acc[ti][tj][r] = 0.0f;
                }
        }
        vt += tstep * 256;
    }
}

int main(int argc, char** argv) {
    const int64_t nv = argc > 1 ? atoll(argv[1]) : 10000000;
    const int check = argc > 2 ? atoi(argv[2]) : 0;
    const int nq = 1024, d = 768, qt = nq / 256;
    std::vector<__half> hq((size_t)nq * d), hx((size_t)nv * d);
    uint64_t st = 12345;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hq) v = __float2half(rnd());
    for (size_t i = 0; i < hx.size(); i++) hx[i] = __float2half(check ? rnd() : (float)((i * 2654435761u) & 255) / 256.0f - 0.5f);
    __half *dq, *dx; float* dout = nullptr; unsigned long long* dcnt;
    hipMalloc(&dq, hq.size() * 2); hipMalloc(&dx, hx.size() * 2); hipMalloc(&dcnt, 8);
    hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dcnt, 0, 8);
    const int64_t ntiles = (nv + 255) / 256;
    const size_t shm = NSTAGE * STAGE;
    hipFuncSetAttribute((const void*)k_fg4<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipFuncSetAttribute((const void*)k_fg4<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    int ncu = 256;
    const int slots = ((ncu / 8) / qt) * qt;
    unsigned grid = 8u * (unsigned)slots;
    if (check) {
        hipMalloc(&dout, (size_t)nq * nv * 4);
        hipLaunchKernelGGL((k_fg4<true>), dim3(grid), dim3(512), shm, 0, dq, dx, nv, d, qt, ntiles, dout, nv, 0.0f, dcnt);
        hipDeviceSynchronize();
        std::vector<float> ho((size_t)nq * nv);
        hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0; int64_t bad = 0;
        for (int q = 0; q < nq; q += 37)
            for (int64_t v = 0; v < nv; v += 13) {
                double ref = 0;
                for (int k = 0; k < d; k++) ref += (double)__half2float(hq[(size_t)q * d + k]) * (double)__half2float(hx[(size_t)v * d + k]);
                const double e = fabs(ref - ho[(size_t)q * nv + v]);
                if (e > maxerr) maxerr = e;
                if (e > 1e-2) bad++;
            }
        printf("check nv=%lld: max |err| = %g, bad = %lld, err = %s\n", (long long)nv, maxerr, (long long)bad, hipGetErrorString(hipGetLastError()));
        return bad != 0;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k_fg4<false>), dim3(grid), dim3(512), shm, 0, dq, dx, nv, d, qt, ntiles, (float*)nullptr, (int64_t)0, 1e30f, dcnt);
    hipEventRecord(e0);
    const int reps = 5;
    for (int rep = 0; rep < reps; rep++) hipLaunchKernelGGL((k_fg4<false>), dim3(grid), dim3(512), shm, 0, dq, dx, nv, d, qt, ntiles, (float*)nullptr, (int64_t)0, 1e30f, dcnt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double fl = 2.0 * nq * (double)nv * d;
    printf("NSTAGE=%d nv=%lld: %.3f ms per launch, %.1f TFLOP/s (%.3f of 2500), err = %s\n", NSTAGE, (long long)nv, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0, hipGetErrorString(hipGetLastError()));
    return 0;
}
