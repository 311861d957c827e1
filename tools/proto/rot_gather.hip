// rot_gather.hip — stand-alone prototype of the IVF-PQ "rotated" fast scan inner loop (gfx950).
//
// Question answered before the engine is rewritten around it: what does a bank-conflict-free LDS table
// gather + MFMA-i8 accumulation cost per (vector, sub-quantiser, 4 queries) on MI355X?
//
//   * 16 vectors x 96 sub-quantisers per block; lane (g, i) = (lane >> 4, lane & 15) owns vector i and, at step
//     s, sub-quantiser m = 16 g + ((i + s) & 15) (64-wide phase) or 64 + 16 (g & 1) + ((i + s + 8 (g >> 1)) & 15)
//     (32-wide phase).  The code bytes are STORED in that order, so a lane reads 16 + 8 contiguous bytes.
//   * table [code][m] with 256-byte rows: bank = m % 32, and the 32 lanes of a half-wave always hold 32
//     different m % 32 -> every ds_read_b32 is conflict-free whatever the codes are.
//   * address = (code << 8) | rot byte: ONE v_perm_b32.
//   * the gathered dword (4 queries' int8 entries) is an A operand of v_mfma_i32_16x16x64_i8 against a constant
//     one-hot B: the matrix core adds 16 gathers x 16 vectors per instruction; VALU does no accumulation.
//
// Modes: 0 = rotated conflict-free (the design), 1 = same instruction stream, random banks (all lanes same m).
// Build: hipcc --offload-arch=gfx950 -O3 -o rot_gather rot_gather.hip ; run: ./rot_gather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int M = 96;
constexpr int BLK_BYTES = 16 * M;          // 1536
constexpr int BLOCKS_PER_TILE = 512;       // 8192 vectors
constexpr int TAB_BYTES = 128 * 1024;      // plane A 64 KiB + plane B (half rows) 64 KiB

__device__ __forceinline__ uint32_t lds_rd(uint32_t addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>((uintptr_t)addr);
}

template <int MODE, bool OUT, int VAR = 0>
__global__ __launch_bounds__(1024) void k_rot(const uint8_t* __restrict__ codes, int64_t ntiles_mod,
                                              const uint32_t* __restrict__ img /* LDS image */, int thr, int nblk,
                                              int* __restrict__ out /* optional [tile][512][16][4] */,
                                              unsigned long long* __restrict__ hits) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    // stage the table: item = (code, 4 consecutive m) -> one ds_write_b128 (conflict-free, 16-byte slots)
    for (int e = tid; e < TAB_BYTES / 16; e += 1024)
        reinterpret_cast<uint4*>(lds)[e] = reinterpret_cast<const uint4*>(img)[e];
    // per-lane rotation bytes
    uint32_t RA[4], RB[3];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int s = r * 4 + b;
            const uint32_t rot = MODE == 0 ? (uint32_t)(64 * g + 4 * ((i + s) & 15)) : (uint32_t)(4 * s + 64 * (g & 1));
            v |= rot << (8 * b);
        }
        RA[r] = v;
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
        uint32_t v = 0x01000000u;
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const int s = r * 3 + b;
            if (s < 8) {
                const uint32_t rot = MODE == 0 ? (uint32_t)(64 * (g & 1) + 4 * ((i + s + 8 * (g >> 1)) & 15)) : (uint32_t)(4 * s);
                v |= rot << (8 * b);
            }
        }
        RB[r] = v;
    }
    // one-hot B: column n = lane & 15 picks byte n of every dword
    const int n = lane & 15;
    const int bsel = n < 4 ? (1 << (8 * n)) : 0;
    const v4i Bm = {bsel, bsel, bsel, bsel};
    const int mythr = n < 4 ? thr : 0x7fffffff;
    __syncthreads();

    const int64_t tile = (int64_t)blockIdx.x % ntiles_mod;
    const uint8_t* tp = codes + tile * (int64_t)BLOCKS_PER_TILE * BLK_BYTES;
    unsigned long long nh = 0;
    constexpr int D = 4;                       // code blocks in flight per wave (1.5 KiB each)
    uint4 ca[D]; uint2 cb[D];
#pragma unroll
    for (int dd = 0; dd < D; dd++) {
        int b = w + 16 * dd; b = b < nblk ? b : nblk - 1;
        const uint8_t* bp = tp + (int64_t)b * BLK_BYTES;
        ca[dd] = *reinterpret_cast<const uint4*>(bp + lane * 16);
        cb[dd] = *reinterpret_cast<const uint2*>(bp + 1024 + lane * 8);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the loop header then only inherits the back edge's counted waits
#pragma unroll 1
    for (int b0 = w; b0 < nblk; b0 += 16 * D) {
#pragma unroll
        for (int dd = 0; dd < D; dd++) {
            const int b = b0 + 16 * dd;
            const uint32_t cw[6] = {ca[dd].x, ca[dd].y, ca[dd].z, ca[dd].w, cb[dd].x, cb[dd].y};
            uint32_t gv[24];
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const uint32_t sel = 0x0c0c0000u | ((uint32_t)(4 + (s & 3)) << 8) | (uint32_t)(s & 3);
                gv[s] = __builtin_amdgcn_perm(cw[s >> 2], RA[s >> 2], sel);
            }
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const uint32_t sel = 0x0c030000u | ((uint32_t)(4 + (s & 3)) << 8) | (uint32_t)(s % 3);
                gv[16 + s] = __builtin_amdgcn_perm(cw[4 + (s >> 2)], RB[s / 3], sel);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(VAR & 4)) {   // the slot's code registers are dead now: refill them in place (clamped, branch-free)
                int bn = b + 16 * D; bn = bn < nblk ? bn : nblk - 1;
                const uint8_t* bp = tp + (int64_t)bn * BLK_BYTES;
                ca[dd] = *reinterpret_cast<const uint4*>(bp + lane * 16);
                cb[dd] = *reinterpret_cast<const uint2*>(bp + 1024 + lane * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int thr_b = b < nblk ? mythr : 0x7fffffff;
#pragma unroll
            for (int s = 0; s < 24; s++) if (!(VAR & 2)) gv[s] = lds_rd(gv[s]);
            v4i C = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 6; t++) {
                const v4i A = {(int)gv[4 * t], (int)gv[4 * t + 1], (int)gv[4 * t + 2], (int)gv[4 * t + 3]};
                if (VAR & 1) { C[0] += A[0]; C[1] += A[1]; C[2] += A[2]; C[3] += A[3]; }
                else if (VAR & 8) { v4i Z = {0, 0, 0, 0}; v4i P = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, Bm, Z, 0, 0, 0); C[0] += P[0]; C[1] += P[1]; C[2] += P[2]; C[3] += P[3]; }
                else C = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, Bm, C, 0, 0, 0);
            }
            if (OUT) {
                if (n < 4 && b < nblk) {
#pragma unroll
                    for (int r = 0; r < 4; r++) out[(((int64_t)blockIdx.x * BLOCKS_PER_TILE + b) * 16 + (4 * g + r)) * 4 + n] = C[r];
                }
            } else {
                const bool hit = (C[0] >= thr_b) | (C[1] >= thr_b) | (C[2] >= thr_b) | (C[3] >= thr_b);
                if (__builtin_amdgcn_ballot_w64(hit)) {
#pragma unroll
                    for (int r = 0; r < 4; r++) nh += (C[r] >= thr_b) ? 1 : 0;
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) nh += __shfl_xor(nh, off);
    if (lane == 0 && nh) atomicAdd(hits + (blockIdx.x & 255), nh);
}


// ---- software-pipelined form: block j's MFMAs are interleaved with block j+1's address perms and LDS gathers, so every
// wave's own instruction stream keeps VALU, LDS and the matrix core busy together (waves of a SIMD run in lockstep
// after the barrier; without this the three units take turns).
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
template <int PIN, int NOLOAD = 0>
__global__ __launch_bounds__(1024) void k_rot2(const uint8_t* __restrict__ codes, int64_t ntiles_mod,
                                               const uint32_t* __restrict__ img, int thr, int nblk,
                                               int* __restrict__ out, unsigned long long* __restrict__ hits) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    for (int e = tid; e < TAB_BYTES / 16; e += 1024)
        reinterpret_cast<uint4*>(lds)[e] = reinterpret_cast<const uint4*>(img)[e];
    uint32_t RA[4], RB[3];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) v |= (uint32_t)(64 * g + 4 * ((i + r * 4 + b) & 15)) << (8 * b);
        RA[r] = v;
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
        uint32_t v = 0x01000000u;
#pragma unroll
        for (int b = 0; b < 3; b++) { const int s = r * 3 + b; if (s < 8) v |= (uint32_t)(64 * (g & 1) + 4 * ((i + s + 8 * (g >> 1)) & 15)) << (8 * b); }
        RB[r] = v;
    }
    const int n = lane & 15;
    const int bsel = n < 4 ? (1 << (8 * n)) : 0;
    const v4i Bm = {bsel, bsel, bsel, bsel};
    const int mythr = n < 4 ? thr : 0x7fffffff;
    __syncthreads();
    const int64_t tile = (int64_t)blockIdx.x % ntiles_mod;
    const uint8_t* tp = codes + tile * (int64_t)BLOCKS_PER_TILE * BLK_BYTES;
    unsigned long long nh = 0;
    constexpr int D = 4;
    uint4 ca[D]; uint2 cb[D];
#pragma unroll
    for (int dd = 0; dd < D; dd++) {
        int b = w + 16 * dd; b = b < nblk ? b : nblk - 1;
        const uint8_t* bp = tp + (int64_t)b * BLK_BYTES;
        ca[dd] = *reinterpret_cast<const uint4*>(bp + lane * 16);
        cb[dd] = *reinterpret_cast<const uint2*>(bp + 1024 + lane * 8);
    }
    uint32_t G[2][24];
    auto addr = [&](int dd, int s) -> uint32_t {   // LDS address of gather s of the block in slot dd
        if (s < 16) return __builtin_amdgcn_perm(((const uint32_t*)&ca[dd])[s >> 2], RA[s >> 2], 0x0c0c0000u | ((uint32_t)(4 + (s & 3)) << 8) | (uint32_t)(s & 3));
        const int s2 = s - 16;
        return __builtin_amdgcn_perm(((const uint32_t*)&cb[dd])[s2 >> 2], RB[s2 / 3], 0x0c030000u | ((uint32_t)(4 + (s2 & 3)) << 8) | (uint32_t)(s2 % 3));
    };
    // prologue: gathers of block 0 (slot 0), then refill slot 0
#pragma unroll
    for (int s = 0; s < 24; s++) G[0][s] = lds_rd(addr(0, s));
    {
        int bn = w + 16 * D; bn = bn < nblk ? bn : nblk - 1;
        const uint8_t* bp = tp + (int64_t)bn * BLK_BYTES;
        ca[0] = *reinterpret_cast<const uint4*>(bp + lane * 16);
        cb[0] = *reinterpret_cast<const uint2*>(bp + 1024 + lane * 8);
    }
#pragma unroll 1
    for (int b0 = w; b0 < nblk; b0 += 16 * D) {
#pragma unroll
        for (int dd = 0; dd < D; dd++) {
            const int b = b0 + 16 * dd;             // block whose gathers sit in G[dd & 1]
            const int nd = (dd + 1) & 3;            // slot of block b + 16
            v4i C = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 6; t++) {
                uint32_t a0 = addr(nd, 4 * t), a1 = addr(nd, 4 * t + 1), a2 = addr(nd, 4 * t + 2), a3 = addr(nd, 4 * t + 3);
                G[(dd + 1) & 1][4 * t] = lds_rd(a0); G[(dd + 1) & 1][4 * t + 1] = lds_rd(a1);
                G[(dd + 1) & 1][4 * t + 2] = lds_rd(a2); G[(dd + 1) & 1][4 * t + 3] = lds_rd(a3);
                const v4i A = {(int)G[dd & 1][4 * t], (int)G[dd & 1][4 * t + 1], (int)G[dd & 1][4 * t + 2], (int)G[dd & 1][4 * t + 3]};
                C = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, Bm, C, 0, 0, 0);
                if (PIN) { SGB(0x002, 4); SGB(0x100, 4); SGB(0x008, 1); }
            }
            if (!NOLOAD) {   // slot nd is consumed: refill it with block b + 16 + 16 D
                int bn = b + 16 + 16 * D; bn = bn < nblk ? bn : nblk - 1;
                const uint8_t* bp = tp + (int64_t)bn * BLK_BYTES;
                ca[nd] = *reinterpret_cast<const uint4*>(bp + lane * 16);
                cb[nd] = *reinterpret_cast<const uint2*>(bp + 1024 + lane * 8);
            }
            const int thr_b = b < nblk ? mythr : 0x7fffffff;
            if (out) {
                if (n < 4 && b < nblk) {
#pragma unroll
                    for (int r = 0; r < 4; r++) out[(((int64_t)blockIdx.x * BLOCKS_PER_TILE + b) * 16 + (4 * g + r)) * 4 + n] = C[r];
                }
            } else {
                const bool hit = (C[0] >= thr_b) | (C[1] >= thr_b) | (C[2] >= thr_b) | (C[3] >= thr_b);
                if (__builtin_amdgcn_ballot_w64(hit)) {
#pragma unroll
                    for (int r = 0; r < 4; r++) nh += (C[r] >= thr_b) ? 1 : 0;
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) nh += __shfl_xor(nh, off);
    if (lane == 0 && nh) atomicAdd(hits + (blockIdx.x & 255), nh);
}

int main(int argc, char** argv) {
    const int ntiles = 1024;                 // 1024 x 786 KB = 805 MB of codes (well past the 256 MB MALL)
    const int grid = argc > 1 ? atoi(argv[1]) : 8192;
    std::vector<uint8_t> hc((size_t)ntiles * BLOCKS_PER_TILE * BLK_BYTES);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t k = 0; k < hc.size(); k += 8) { uint64_t v = rnd(); memcpy(&hc[k], &v, 8); }
    std::vector<uint32_t> ht(96 * 256);
    for (auto& v : ht) v = (uint32_t)rnd();
    uint8_t* dc; uint32_t* dt; int* dout; unsigned long long* dh;
    const int REP = 8; CK(hipMalloc(&dc, hc.size() * REP)); CK(hipMalloc(&dt, ht.size() * 4)); CK(hipMalloc(&dh, 8 * 256));
    for (int r = 0; r < REP; r++) CK(hipMemcpy(dc + (size_t)r * hc.size(), hc.data(), hc.size(), hipMemcpyHostToDevice));
    std::vector<uint32_t> himg(TAB_BYTES / 4, 0);
    for (int m = 0; m < 96; m++)
        for (int c = 0; c < 256; c++)
            himg[(m < 64 ? c * 256 + m * 4 : 65536 + c * 256 + (m - 64) * 4) / 4] = ht[m * 256 + c];
    CK(hipFree(dt)); CK(hipMalloc(&dt, TAB_BYTES));
    CK(hipMemcpy(dt, himg.data(), TAB_BYTES, hipMemcpyHostToDevice));
    CK(hipMemset(dh, 0, 8 * 256));
    CK(hipFuncSetAttribute((const void*)k_rot<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, TAB_BYTES));
    CK(hipFuncSetAttribute((const void*)k_rot<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, TAB_BYTES));

    // ---- correctness: 4 tiles, every (vector, query) sum against the host
    {
        const int vt = 4;
        size_t no = (size_t)vt * BLOCKS_PER_TILE * 16 * 4;
        CK(hipMalloc(&dout, no * 4));
        CK(hipFuncSetAttribute((const void*)k_rot<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, TAB_BYTES));
        hipLaunchKernelGGL((k_rot<0, true>), dim3(vt), dim3(1024), TAB_BYTES, 0, dc, (int64_t)ntiles, dt, 0, BLOCKS_PER_TILE, dout, dh);
        CK(hipDeviceSynchronize());
        std::vector<int> ho(no);
        CK(hipMemcpy(ho.data(), dout, no * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (int t = 0; t < vt; t++)
            for (int b = 0; b < BLOCKS_PER_TILE; b++) {
                const uint8_t* bp = &hc[((size_t)t * BLOCKS_PER_TILE + b) * BLK_BYTES];
                for (int i = 0; i < 16; i++) {
                    int sum[4] = {0, 0, 0, 0};
                    for (int g = 0; g < 4; g++) {
                        const int lane = 16 * g + i;
                        for (int st = 0; st < 16; st++) {
                            const int m = 16 * g + ((i + st) & 15);
                            const uint32_t e = ht[m * 256 + bp[lane * 16 + st]];
                            for (int q = 0; q < 4; q++) sum[q] += (int8_t)(e >> (8 * q));
                        }
                        for (int st = 0; st < 8; st++) {
                            const int m = 64 + 16 * (g & 1) + ((i + st + 8 * (g >> 1)) & 15);
                            const uint32_t e = ht[m * 256 + bp[1024 + lane * 8 + st]];
                            for (int q = 0; q < 4; q++) sum[q] += (int8_t)(e >> (8 * q));
                        }
                    }
                    for (int q = 0; q < 4; q++)
                        if (ho[(((size_t)t * BLOCKS_PER_TILE + b) * 16 + i) * 4 + q] != sum[q]) bad++;
                }
            }
        printf("correctness: %zu mismatches of %zu\n", bad, no);
        CK(hipFree(dout));
    }
    // ---- timing
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {
        int64_t tmod = (int64_t)ntiles * REP;
        auto run = [&](const char* nm, void (*kern)(const uint8_t*, int64_t, const uint32_t*, int, int, int*, unsigned long long*)) {
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TAB_BYTES));
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), TAB_BYTES, 0, dc, tmod, dt, 2200, 512, nullptr, dh);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }
            printf("variant %-28s %.3f ms  (%.2f clk/CU per wave-gather @2.4GHz)\n", nm, best, best * 1e-3 * 2.4e9 * 256 / ((double)grid * 8192 * 96 / 64));
        };
        for (int64_t tm : {(int64_t)32, (int64_t)256, (int64_t)ntiles * REP}) {
            tmod = tm; printf("-- %lld distinct tiles (%.0f MB)\n", (long long)tm, tm * 0.786432);
            run("full", k_rot<0, false, 0>);
            run("pipelined", k_rot2<0>);
            run("pipelined+pinned", k_rot2<1>);
            run("no-mfma (v_add)", k_rot<0, false, 1>);
            run("no-lds-read", k_rot<0, false, 2>);
            run("no-mfma, no-lds", k_rot<0, false, 3>);
        }
        tmod = (int64_t)ntiles * REP;
        run("full", k_rot<0, false, 0>);
        run("pipelined", k_rot2<0>);
        run("pipelined+pinned", k_rot2<1>);
        run("pipelined, no-global", k_rot2<0, 1>);
        run("pipelined+pinned, no-global", k_rot2<1, 1>);
        run("no-mfma (v_add)", k_rot<0, false, 1>);
        run("no-lds-read", k_rot<0, false, 2>);
        run("no-global-reload", k_rot<0, false, 4>);
        run("independent mfma + v_add", k_rot<0, false, 8>);
        run("no-mfma, no-lds", k_rot<0, false, 3>);
        run("no-global, no-mfma", k_rot<0, false, 5>);
        run("no-global, no-lds", k_rot<0, false, 6>);
        run("nothing but perms", k_rot<0, false, 7>);
    }
    const int nblks[4] = {512, 0, 256, 512};
    for (int mode = 0; mode < 1; mode++) {
        for (int rep = 0; rep < 1; rep++) {
            const int nblk = nblks[rep];
            CK(hipMemset(dh, 0, 8 * 256));
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL((k_rot<0, false>), dim3(grid), dim3(1024), TAB_BYTES, 0, dc, (int64_t)ntiles * REP, dt, 2200, nblk, nullptr, dh);
            else hipLaunchKernelGGL((k_rot<1, false>), dim3(grid), dim3(1024), TAB_BYTES, 0, dc, (int64_t)ntiles * REP, dt, 2200, nblk, nullptr, dh);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long nhv[256], nh = 0; CK(hipMemcpy(nhv, dh, 8 * 256, hipMemcpyDeviceToHost)); for (int z = 0; z < 256; z++) nh += nhv[z];
            const double vec = (double)grid * 16 * (nblk > 0 ? nblk : 1), gathers = vec * 96 / 64;   // wave-level ds_read_b32
            printf("mode %d nblk %d: %.3f ms  %.1f Gvec*4q/s  %.2f TB/s of codes  %.2f clk/CU per wave-gather @2.4GHz  hits %llu\n", mode, nblk, ms,
                   vec / ms * 1e-6, vec * 96 / ms * 1e-9, ms * 1e-3 * 2.4e9 * 256 / gathers, nh);
        }
    }
    return 0;
}
