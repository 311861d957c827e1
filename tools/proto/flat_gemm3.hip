// flat_gemm3.hip — prototype of the round-3 Flat scan kernel (S = Q X^T, fp16 in, fp32 accumulate) for gfx950.
//   256 queries x 512 db rows per workgroup, 8 waves (2 x 4), 128 x 128 per wave (256 fp32 accumulators per lane),
//   BK = 32, THREE 48 KiB LDS stages filled by LDS-DMA issued from inline assembly (two stages in flight), one barrier
//   per K step, v_mfma_f32_32x32x16_f16.  Persistent workgroups keep a query tile and walk db tiles (4 sharers per XCD).
// Why: k_flat_gemm2 (256 x 256, two 64 KiB stages) moves 123 GB L2 -> LDS per 10M x 1024 launch and its fill and MFMA
// times add up (profiles/r02_flat_gemm_experiments.md).  This shape moves 90 GB and keeps two stages in flight.
// build: hipcc --offload-arch=gfx950 -O3 -o flat_gemm3 flat_gemm3.hip ;  run: ./flat_gemm3 [n_rows] [check]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define STAGE 49152            // A 256 x 64 B | B 512 x 64 B
#define NSTAGE 3

__device__ __forceinline__ void dma16(const void* g, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds) : "memory");
}

template <bool STORE>
__global__ __launch_bounds__(512) void k_fg3(const __half* __restrict__ Q, const __half* __restrict__ X, int64_t nv, int ld, int qt,
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

    // DMA sources.  A: instructions 2w, 2w+1 (16 rows each); B: 4w .. 4w+3.  lane j: row j >> 2 of the instruction, LDS chunk j & 3
    const int jr = lane >> 2, jc = lane & 3;
    const char* srcA[2]; int rowB[4]; int cB[4];
    uint32_t ldsA[2], ldsB[4];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int R = (2 * w + i) * 16 + jr;
        const int c = jc ^ ((R >> 2) & 3);
        srcA[i] = reinterpret_cast<const char*>(Q + ((int64_t)qi * 256 + R) * ld) + c * 16;
        ldsA[i] = (uint32_t)((2 * w + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int R = (4 * w + i) * 16 + jr;
        rowB[i] = R; cB[i] = (jc ^ ((R >> 2) & 3)) * 16;
        ldsB[i] = (uint32_t)(16384 + (4 * w + i) * 1024);
    }
    const int64_t vlast = nv - 1;
    int i_kt = 0; int64_t i_vt = t0 * 512;
    auto issue = [&](int buf) {
        const uint32_t base = (uint32_t)(buf * STAGE);
#pragma unroll
        for (int i = 0; i < 2; i++) dma16(srcA[i] + (int64_t)i_kt * 64, base + ldsA[i]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int64_t r = i_vt + rowB[i]; r = r > vlast ? vlast : r;
            dma16(reinterpret_cast<const char*>(X) + r * (int64_t)ld * 2 + (int64_t)i_kt * 64 + cB[i], base + ldsB[i]);
        }
        i_kt++;
        if (i_kt == KT) { i_kt = 0; i_vt += tstep * 512; }
    };

    floatx16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

    const int li = lane & 31, kh = lane >> 5;
    int offs[2];
#pragma unroll
    for (int ss = 0; ss < 2; ss++) offs[ss] = li * 64 + (((2 * ss + kh) ^ ((li >> 2) & 3)) << 4);

    issue(0);
    if (G > 1) issue(1);
    int kt = 0;
    int64_t vt = t0 * 512;
    for (int g = 0; g < G; g++) {
        // my pieces of stage g have landed (the 6 DMAs of stage g + 1 may still be in flight)
        if (g + 1 < G) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                         // stage g complete for everyone; everyone is done with stage g - 1
        if (g + 2 < G) issue((g + 2) % NSTAGE);
        const unsigned char* As = smem + (g % NSTAGE) * STAGE + wr * (128 * 64);
        const unsigned char* Bs = smem + (g % NSTAGE) * STAGE + 16384 + wc * (128 * 64);
#pragma unroll
        for (int ss = 0; ss < 2; ss++) {
            half8 fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; t++) fa[t] = *reinterpret_cast<const half8*>(As + t * 2048 + offs[ss]);
#pragma unroll
            for (int t = 0; t < 4; t++) fb[t] = *reinterpret_cast<const half8*>(Bs + t * 2048 + offs[ss]);
#pragma unroll
            for (int ti = 0; ti < 4; ti++)
#pragma unroll
                for (int tj = 0; tj < 4; tj++)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ti], fb[tj], acc[ti][tj], 0, 0, 0);
        }
        if (++kt < KT) continue;
        kt = 0;
        // ---- epilogue of a (query tile, db tile) item
        const int lj = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int tj = 0; tj < 4; tj++) {
            const int64_t col = vt + wc * 128 + tj * 32 + lj;
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
        vt += tstep * 512;
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
    const int64_t ntiles = (nv + 511) / 512;
    const size_t shm = NSTAGE * STAGE;
    hipFuncSetAttribute((const void*)k_fg3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipFuncSetAttribute((const void*)k_fg3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    int ncu = 256;
    const int slots = ((ncu / 8) / qt) * qt;
    unsigned grid = 8u * (unsigned)slots;
    if (check) {
        hipMalloc(&dout, (size_t)nq * nv * 4);
        hipLaunchKernelGGL((k_fg3<true>), dim3(grid), dim3(512), shm, 0, dq, dx, nv, d, qt, ntiles, dout, nv, 0.0f, dcnt);
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
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k_fg3<false>), dim3(grid), dim3(512), shm, 0, dq, dx, nv, d, qt, ntiles, (float*)nullptr, (int64_t)0, 1e30f, dcnt);
    hipEventRecord(e0);
    const int reps = 5;
    for (int rep = 0; rep < reps; rep++) hipLaunchKernelGGL((k_fg3<false>), dim3(grid), dim3(512), shm, 0, dq, dx, nv, d, qt, ntiles, (float*)nullptr, (int64_t)0, 1e30f, dcnt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double fl = 2.0 * nq * (double)nv * d;
    printf("nv=%lld: %.3f ms per launch, %.1f TFLOP/s (%.3f of 2500), err = %s\n", (long long)nv, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0, hipGetErrorString(hipGetLastError()));
    return 0;
}
