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
#ifndef VARIANT
#define VARIANT 0
#endif

__device__ __forceinline__ void dma16(const void* g, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds) : "memory");
}
#ifdef TRACE
__device__ unsigned long long g_tr[8 * 64 * 8];      // [wave][stage 200..263][mark]
#define MARK(i) do { if (blockIdx.x == 0 && g >= 200 && g < 264 && lane == 0) g_tr[(w * 64 + (g - 200)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MARK(i)
#endif
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
        const int c = jc ^ ((R >> 3) & 3);
#if VARIANT >= 2      // strip-major tiles: [tile][k strip of 32][row 0..255][64 B] — a stage's operand is ONE contiguous 16 KiB block
        srcA[i] = reinterpret_cast<const char*>(Q) + ((int64_t)qi * (ld / 32) * 256 + R) * 64 + c * 16;
#else
        srcA[i] = reinterpret_cast<const char*>(Q + ((int64_t)qi * 256 + R) * ld) + c * 16;
#endif
        ldsA[i] = (uint32_t)((2 * w + i) * 1024);
        rowB[i] = R; cB[i] = c * 16;
        ldsB[i] = (uint32_t)(16384 + (2 * w + i) * 1024);
    }
    const int64_t vlast = nv - 1;
    int i_kt = 0; int64_t i_vt = t0 * 256;
    auto issue = [&](int buf) {
        const uint32_t base = (uint32_t)(buf * STAGE);
#if VARIANT >= 2
#pragma unroll
        for (int i = 0; i < 2; i++) dma16(srcA[i] + (int64_t)i_kt * (256 * 64), base + ldsA[i]);
#pragma unroll
        for (int i = 0; i < 2; i++)      // (tiles are whole: the last one is padded)
            dma16(reinterpret_cast<const char*>(X) + ((i_vt >> 8) * (int64_t)(ld / 32) + i_kt) * (256 * 64) + rowB[i] * 64 + cB[i], base + ldsB[i]);
#else
#pragma unroll
        for (int i = 0; i < 2; i++) dma16(srcA[i] + (int64_t)i_kt * 64, base + ldsA[i]);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            int64_t r = i_vt + rowB[i]; r = r > vlast ? vlast : r;
            dma16(reinterpret_cast<const char*>(X) + r * (int64_t)ld * 2 + (int64_t)i_kt * 64 + cB[i], base + ldsB[i]);
        }
#endif
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

    // fragment of a 32 x 32 x 16 MFMA: lane (li = lane & 31, kh = lane >> 5) holds row li, k = 8 kh .. 8 kh + 7 of the 16: chunk 2 ss + kh.
    // Swizzle: ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) of MI355X_MICROARCH.md, 16 lanes over
    // the 64 banks; with 64-byte rows a lane's 16-byte slot of the 256-byte bank row is (li & 3) * 4 + chunk position, so the four quads
    // of a group (li >> 2 in {0,3,5,6} or {1,2,4,7}) need four different chunk positions: XOR with (li >> 3) & 3 (first try, (li >> 1) & 3:
    // PMC SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE)
    const int li = lane & 31, kh = lane >> 5;
    int offs[2];
#pragma unroll
    for (int ss = 0; ss < 2; ss++) offs[ss] = li * 64 + (((2 * ss + kh) ^ ((li >> 3) & 3)) << 4);

#if VARIANT == 3
    // ---- ping-pong (MI355X guide, K-loop recipe): waves 0-3 (group 0) and 4-7 (group 1) share the SIMDs pairwise; group 1 runs one
    // barrier behind, so in every barrier-to-barrier slot one wave of a SIMD issues its 8 MFMAs of a K = 16 sub-step while the other
    // forms DMA addresses and reads the next sub-step's fragments.  PP_AHEAD stages ahead of the one being computed are in flight.
#ifndef PP_AHEAD
#define PP_AHEAD 2
#endif
    auto issue_half = [&](int buf, int half) {      // half 0: the A pieces, half 1: the B pieces (and the advance)
        const uint32_t base = (uint32_t)(buf * STAGE);
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < 2; i++) dma16(srcA[i] + (int64_t)i_kt * (256 * 64), base + ldsA[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 2; i++)
                dma16(reinterpret_cast<const char*>(X) + ((i_vt >> 8) * (int64_t)(ld / 32) + i_kt) * (256 * 64) + rowB[i] * 64 + cB[i], base + ldsB[i]);
            i_kt++;
            if (i_kt == KT) { i_kt = 0; i_vt += tstep * 256; }
        }
    };
    const int grp = w >> 2;
#pragma unroll
    for (int p = 0; p < PP_AHEAD; p++) if (p < G) { issue_half(p, 0); issue_half(p, 1); }
    if (G > PP_AHEAD - 1) wait_vm<4 * (PP_AHEAD - 1)>(); else wait_vm<0>();
    __syncthreads();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    int kt = 0;
    int64_t vt = t0 * 256;
    for (int g = 0; g < G; g++) {
        const unsigned char* As = smem + (g % NSTAGE) * STAGE + wr * (128 * 64);
        const unsigned char* Bs = smem + (g % NSTAGE) * STAGE + 16384 + wc * (64 * 64);
        const bool more = g + PP_AHEAD < G;
#pragma unroll
        for (int ss = 0; ss < 2; ss++) {
            half8 fa[4], fb[2];
            if (more) issue_half((g + PP_AHEAD) % NSTAGE, ss);
#pragma unroll
            for (int t = 0; t < 4; t++) fa[t] = *reinterpret_cast<const half8*>(As + t * 2048 + offs[ss]);
#pragma unroll
            for (int t = 0; t < 2; t++) fb[t] = *reinterpret_cast<const half8*>(Bs + t * 2048 + offs[ss]);
            // stage g + 1 must be complete before the barrier that ends slot 4 g + 3: group 1 is in its second load phase there
            if (ss == 1 && grp == 1) { if (more) wait_vm<4 * (PP_AHEAD - 1)>(); else wait_vm<0>(); }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ti = 0; ti < 4; ti++)
#pragma unroll
                for (int tj = 0; tj < 2; tj++)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ti], fb[tj], acc[ti][tj], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            // ... group 0 in its second MFMA phase
            if (ss == 1 && grp == 0) { if (more) wait_vm<4 * (PP_AHEAD - 1)>(); else wait_vm<0>(); }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
#elif VARIANT == 4
    // ---- software pipeline inside every wave: the barrier at the top of stage g certifies stage g + 1 (two more stages stay in flight), so
    // the first fragments of stage g + 1 are read during the last MFMAs of stage g; the four DMA pieces of stage g + 3 and the fragment
    // reads of the second K = 16 sub-step are placed between the MFMAs (the matrix pipe runs 32 cycles per MFMA, the issue slot is free)
    auto dma_piece = [&](int buf, int pc) {      // pieces 0, 1: A rows; 2, 3: B rows (+ the advance)
        const uint32_t base = (uint32_t)(buf * STAGE);
        if (pc < 2) dma16(srcA[pc] + (int64_t)i_kt * (256 * 64), base + ldsA[pc]);
        else {
            dma16(reinterpret_cast<const char*>(X) + ((i_vt >> 8) * (int64_t)(ld / 32) + i_kt) * (256 * 64) + rowB[pc - 2] * 64 + cB[pc - 2], base + ldsB[pc - 2]);
            if (pc == 3) { i_kt++; if (i_kt == KT) { i_kt = 0; i_vt += tstep * 256; } }
        }
    };
#pragma unroll
    for (int p = 0; p < 3; p++) if (p < G) { dma_piece(p, 0); dma_piece(p, 1); dma_piece(p, 2); dma_piece(p, 3); }
    if (G >= 3) wait_vm<8>(); else wait_vm<0>();
    __syncthreads();
    half8 f0a[4], f0b[2], f1a[4], f1b[2];
    {
        const unsigned char* As = smem + wr * (128 * 64);
        const unsigned char* Bs = smem + 16384 + wc * (64 * 64);
#pragma unroll
        for (int t = 0; t < 4; t++) f0a[t] = *reinterpret_cast<const half8*>(As + t * 2048 + offs[0]);
#pragma unroll
        for (int t = 0; t < 2; t++) f0b[t] = *reinterpret_cast<const half8*>(Bs + t * 2048 + offs[0]);
    }
    int kt = 0;
    int64_t vt = t0 * 256;
#define MM(F, ti, tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F##a[ti], F##b[tj], acc[ti][tj], 0, 0, 0)
#define PIN __builtin_amdgcn_sched_barrier(0)
    for (int g = 0; g < G; g++) {
        if (g + 2 < G) wait_vm<4>(); else wait_vm<0>();          // my pieces of stage g + 1 (stage g + 2 may be in flight)
        __builtin_amdgcn_s_barrier();                            // stage g + 1 complete; everyone has read all of stage g - 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // F0 (read at the end of the previous iteration)
        const bool more = g + 3 < G;
        const int nb = (g + 3) % NSTAGE;
        const unsigned char* As = smem + (g % NSTAGE) * STAGE + wr * (128 * 64);
        const unsigned char* Bs = smem + (g % NSTAGE) * STAGE + 16384 + wc * (64 * 64);
        const unsigned char* An = smem + ((g + 1) % NSTAGE) * STAGE + wr * (128 * 64);
        const unsigned char* Bn = smem + ((g + 1) % NSTAGE) * STAGE + 16384 + wc * (64 * 64);
        PIN;
        MM(f0, 0, 0); MM(f0, 0, 1); PIN;
        if (more) dma_piece(nb, 0);
        PIN; MM(f0, 1, 0); MM(f0, 1, 1); PIN;
        if (more) dma_piece(nb, 1);
        PIN; MM(f0, 2, 0); MM(f0, 2, 1); PIN;
#pragma unroll
        for (int t = 0; t < 4; t++) f1a[t] = *reinterpret_cast<const half8*>(As + t * 2048 + offs[1]);
#pragma unroll
        for (int t = 0; t < 2; t++) f1b[t] = *reinterpret_cast<const half8*>(Bs + t * 2048 + offs[1]);
        PIN; MM(f0, 3, 0); MM(f0, 3, 1); PIN;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PIN; MM(f1, 0, 0); MM(f1, 0, 1); PIN;
        if (more) dma_piece(nb, 2);
        PIN; MM(f1, 1, 0); MM(f1, 1, 1); PIN;
        if (more) dma_piece(nb, 3);
        PIN; MM(f1, 2, 0); MM(f1, 2, 1); PIN;
        if (g + 1 < G) {
#pragma unroll
            for (int t = 0; t < 4; t++) f0a[t] = *reinterpret_cast<const half8*>(An + t * 2048 + offs[0]);
#pragma unroll
            for (int t = 0; t < 2; t++) f0b[t] = *reinterpret_cast<const half8*>(Bn + t * 2048 + offs[0]);
        }
        PIN; MM(f1, 3, 0); MM(f1, 3, 1); PIN;
#else
#pragma unroll
    for (int p = 0; p < AHEAD; p++) if (p < G) issue(p);
    int kt = 0;
    int64_t vt = t0 * 256;
    for (int g = 0; g < G; g++) {
        // my pieces of stage g have landed: the stages g + 1 .. g + AHEAD - 1 issued after it (4 DMAs each) may still be in flight
        const int younger = (G - 1 - g) < (AHEAD - 1) ? (G - 1 - g) : (AHEAD - 1);
        MARK(0);
        if (younger >= 2) wait_vm<8>(); else if (younger == 1) wait_vm<4>(); else wait_vm<0>();
        MARK(1);
        __syncthreads();                         // stage g complete for everyone; everyone is done with stage g - 1
        MARK(2);
#if VARIANT != 1
        if (g + AHEAD < G) issue((g + AHEAD) % NSTAGE);
#endif
        MARK(3);
        const unsigned char* As = smem + (g % NSTAGE) * STAGE + wr * (128 * 64);
        const unsigned char* Bs = smem + (g % NSTAGE) * STAGE + 16384 + wc * (64 * 64);
        half8 fa[2][4], fb[2][2];
#pragma unroll
        for (int ss = 0; ss < 2; ss++) {
#pragma unroll
            for (int t = 0; t < 4; t++) fa[ss][t] = *reinterpret_cast<const half8*>(As + t * 2048 + offs[ss]);
#pragma unroll
            for (int t = 0; t < 2; t++) fb[ss][t] = *reinterpret_cast<const half8*>(Bs + t * 2048 + offs[ss]);
#if VARIANT == 1
            if (ss == 0) {      // the first sub-step's fragment reads are in flight before the DMA addresses are formed
                __builtin_amdgcn_sched_barrier(0);
                if (g + AHEAD < G) issue((g + AHEAD) % NSTAGE);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
#ifdef TRACE
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        MARK(4);
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int ss = 0; ss < 2; ss++)
#pragma unroll
            for (int ti = 0; ti < 4; ti++)
#pragma unroll
                for (int tj = 0; tj < 2; tj++)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ss][ti], fb[ss][tj], acc[ti][tj], 0, 0, 0);
#ifdef TRACE
        __builtin_amdgcn_sched_barrier(0);
        MARK(5);
#endif
#endif
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
#if VARIANT == 3
    if (grp == 0) __builtin_amdgcn_s_barrier();
#endif
}

int main(int argc, char** argv) {
    const int64_t nv = argc > 1 ? atoll(argv[1]) : 10000000;
    const int check = argc > 2 ? atoi(argv[2]) : 0;
    const int nq = 1024, d = 768, qt = nq / 256;
    std::vector<__half> hq((size_t)nq * d), hx((size_t)nv * d);
    uint64_t st = 12345;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hq) v = __float2half(rnd());
    const int realistic = argc > 3 ? atoi(argv[3]) : 0;      // 1: random mantissas (the matrix pipe's power draw depends on the data)
    for (size_t i = 0; i < hx.size(); i++) hx[i] = __float2half((check || realistic) ? rnd() : (float)((i * 2654435761u) & 255) / 256.0f - 0.5f);
    __half *dq, *dx; float* dout = nullptr; unsigned long long* dcnt;
    const int64_t nvp = (nv + 255) / 256 * 256;
    hipMalloc(&dq, hq.size() * 2); hipMalloc(&dx, (size_t)nvp * d * 2); hipMalloc(&dcnt, 8);
#if VARIANT >= 2
    {
        auto permute = [&](const std::vector<__half>& src, int64_t rows, int64_t rows_pad) {
            std::vector<__half> o((size_t)rows_pad * d);
            const int KT = d / 32;
            for (int64_t r = 0; r < rows_pad; r++) {
                const int64_t rs = r < rows ? r : rows - 1, T = r >> 8, rr = r & 255;
                for (int kt = 0; kt < KT; kt++)
                    for (int e = 0; e < 32; e++) o[(size_t)(((T * KT + kt) * 256 + rr) * 32 + e)] = src[(size_t)rs * d + kt * 32 + e];
            }
            return o;
        };
        if (check) {
            std::vector<__half> pq = permute(hq, nq, nq), px = permute(hx, nv, nvp);
            hipMemcpy(dq, pq.data(), pq.size() * 2, hipMemcpyHostToDevice);
            hipMemcpy(dx, px.data(), px.size() * 2, hipMemcpyHostToDevice);
        } else {      // timing only: the values do not matter
            hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
            hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
        }
    }
#else
    hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
#endif
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
#ifdef TRACE
    {
        std::vector<unsigned long long> tr(8 * 64 * 8);
        hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_tr), tr.size() * 8);
        for (int wv = 0; wv < 8; wv += 1) {
            double d[6] = {0, 0, 0, 0, 0, 0};
            for (int st = 0; st < 63; st++) {
                const unsigned long long* a = &tr[(wv * 64 + st) * 8]; const unsigned long long* b = &tr[(wv * 64 + st + 1) * 8];
                d[0] += (double)(a[1] - a[0]); d[1] += (double)(a[2] - a[1]); d[2] += (double)(a[3] - a[2]); d[3] += (double)(a[4] - a[3]);
                d[4] += (double)(a[5] - a[4]); d[5] += (double)(b[0] - a[0]);
            }
            printf("wave %d (100 MHz ticks x 24 = clk): vmwait %.1f  barrier %.1f  dma-issue %.1f  frag-reads %.1f  mfma %.1f | stage %.1f clk\n", wv,
                   d[0] / 63 * 24, d[1] / 63 * 24, d[2] / 63 * 24, d[3] / 63 * 24, d[4] / 63 * 24, d[5] / 63 * 24);
        }
    }
#endif
    printf("NSTAGE=%d nv=%lld: %.3f ms per launch, %.1f TFLOP/s (%.3f of 2500), err = %s\n", NSTAGE, (long long)nv, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0, hipGetErrorString(hipGetLastError()));
    return 0;
}
