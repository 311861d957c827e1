// k_gemm.hip — the dense contractions of the search path, on the gfx950 matrix cores.
//
//  k_gemm_exact : S = X · Cᵀ with f32-input MFMA (v_mfma_f32_32x32x2_f32).  On gfx950 that
//                 instruction is bit-for-bit a k-ordered fp32 fmaf chain, so every output equals
//                 the oracle's `s = fmaf(x[t], c[t], s)` loop: coarse-quantiser scores (the
//                 quantizer.search inside IndexIVF*::search — reference ivf_flat.py:143,225,
//                 ivf_pq.py:146,230) and IVF assignment (index.add, ivf_flat.py:180, ivf_pq.py:185)
//                 are exact and reproducible.  1024 x 4096 x 768 is 6.4 GFLOP: off the critical path.
//  k_flat_gemm  : Flat scan Q · Xᵀ, fp16 operands / fp32 accumulate (v_mfma_f32_32x32x16_f16),
//                 128x128 tiles staged through padded LDS (IndexFlatIP::search, flat.py:139).
//                 Candidates only — exact scores come from the fp64 re-rank in k_finalize.
//  k_flat_gemm2 : the same contraction for batches > 128 over fp16 rows: 256 x 256 tiles, LDS-DMA staging
//                 (global_load_lds_dwordx4) into XOR-swizzled unpadded LDS, one barrier per K step, one db tile per
//                 workgroup with every query tile passed over it.
//  k_list_scan2 : k_list_scan for fp16 rows with the row stream staged through per-wave LDS-DMA rings (full
//                 128-byte lines per row, five K steps ahead, no barriers), optional in-kernel candidate filter.
//  k_list_scan  : IVF-Flat list scan, list-major: one work item = (list, <=16 probing queries,
//                 row chunk); the list's fp16 rows stream from HBM straight into MFMA B fragments
//                 (16-byte loads, 64 B contiguous per row per instruction) and are dotted against
//                 the group's queries held in LDS (v_mfma_f32_16x16x32_f16).  HBM-bound by design:
//                 a list is read once per batch for all queries that probe it.
#include "rsx_internal.h"

namespace rsx {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// =======================================================================================
// exact f32 GEMM
// =======================================================================================
template <bool XF16>
__device__ inline void load16_f32(const void* X, int64_t row, int ldx, int t0, int d, bool vec_ok, float* out) {
    if (XF16) {
        const __half* p = (const __half*)X + row * ldx + t0;
        if (vec_ok && t0 + 16 <= d) {
            uint4 u0 = *reinterpret_cast<const uint4*>(p);
            uint4 u1 = *reinterpret_cast<const uint4*>(p + 8);
            const __half* h0 = reinterpret_cast<const __half*>(&u0);
            const __half* h1 = reinterpret_cast<const __half*>(&u1);
#pragma unroll
            for (int e = 0; e < 8; e++) { out[e] = __half2float(h0[e]); out[8 + e] = __half2float(h1[e]); }
        } else {
#pragma unroll
            for (int e = 0; e < 16; e++) out[e] = (t0 + e < d) ? __half2float(p[e]) : 0.0f;
        }
    } else {
        const float* p = (const float*)X + row * ldx + t0;
        if (vec_ok && t0 + 16 <= d) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float4 f = *reinterpret_cast<const float4*>(p + 4 * e);
                out[4 * e] = f.x; out[4 * e + 1] = f.y; out[4 * e + 2] = f.z; out[4 * e + 3] = f.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; e++) out[e] = (t0 + e < d) ? p[e] : 0.0f;
        }
    }
}

template <bool XF16, bool ARGMAX>
__global__ __launch_bounds__(256) void k_gemm_exact(const void* X, int64_t n, int ldx, int x_vec_ok, const float* C,
                                                    int nc, int d, int c_vec_ok, float* S, int64_t lds_,
                                                    uint64_t* partial, int npart) {
    __shared__ float As[128][33];
    __shared__ float Bs[128][33];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int64_t row0 = (int64_t)blockIdx.y * 128;
    const int col0 = blockIdx.x * 128;

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

    const int sr = tid >> 1, sh = tid & 1;
    int64_t xr = row0 + sr; if (xr > n - 1) xr = n - 1;
    int cr = col0 + sr; if (cr > nc - 1) cr = nc - 1;

    float xa[16], cb[16];
    load16_f32<XF16>(X, xr, ldx, 16 * sh, d, x_vec_ok, xa);
    load16_f32<false>(C, cr, d, 16 * sh, d, c_vec_ok, cb);

    for (int k0 = 0; k0 < d; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; e++) { As[sr][16 * sh + e] = xa[e]; Bs[sr][16 * sh + e] = cb[e]; }
        __syncthreads();
        if (k0 + 32 < d) {
            load16_f32<XF16>(X, xr, ldx, k0 + 32 + 16 * sh, d, x_vec_ok, xa);
            load16_f32<false>(C, cr, d, k0 + 32 + 16 * sh, d, c_vec_ok, cb);
        }
        const int li = lane & 31, lk = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            const int kc = 2 * kk + lk;
            float a0 = As[wr * 64 + li][kc], a1 = As[wr * 64 + 32 + li][kc];
            float b0 = Bs[wc * 64 + li][kc], b1 = Bs[wc * 64 + 32 + li][kc];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    // C/D layout of 32x32 MFMA: reg r of lane l holds (row i = (r&3) + 8*(r>>2) + 4*(l>>5), col j = l&31)
    const int lj = lane & 31, lh = lane >> 5;
    if (!ARGMAX) {
#pragma unroll
        for (int ti = 0; ti < 2; ti++)
#pragma unroll
            for (int tj = 0; tj < 2; tj++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    int64_t row = row0 + wr * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    int col = col0 + wc * 64 + tj * 32 + lj;
                    if (row < n && col < nc) S[row * lds_ + col] = acc[ti][tj][r];
                }
    } else {
#pragma unroll
        for (int ti = 0; ti < 2; ti++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int64_t row = row0 + wr * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                uint64_t best = 0;
#pragma unroll
                for (int tj = 0; tj < 2; tj++) {
                    int col = col0 + wc * 64 + tj * 32 + lj;
                    uint64_t key = (col < nc) ? make_key(acc[ti][tj][r], (uint32_t)col) : 0ull;
                    best = key > best ? key : best;
                }
#pragma unroll
                for (int off = 1; off < 32; off <<= 1) {
                    uint32_t lo = __shfl_xor((uint32_t)(best & 0xffffffffull), off);
                    uint32_t hi = __shfl_xor((uint32_t)(best >> 32), off);
                    uint64_t o = ((uint64_t)hi << 32) | lo;
                    best = o > best ? o : best;
                }
                if (lj == 0 && row < n) partial[row * npart + blockIdx.x * 2 + wc] = best;
            }
    }
}

__global__ void k_argmax_reduce(const uint64_t* partial, int64_t n, int npart, int32_t* assign, float* best) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t b = 0;
    for (int p = 0; p < npart; p++) { uint64_t k = partial[i * npart + p]; b = k > b ? k : b; }
    assign[i] = b ? (int32_t)key_idx(b) : 0;
    if (best) best[i] = b ? key_score(b) : -__builtin_inff();
}

static inline int aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// The same scores for a handful of rows (single-query / small-batch latency path): the 128 x 128 MFMA tile would run
// nc / 128 workgroups with 127 idle rows each, and a thread walking its own centroid row has one cache line in flight.
// Here a workgroup takes 16 centroids: all 256 threads pull the 16 rows into LDS with every load of the tile in flight at
// once (coalesced 256-byte runs), then one thread per (row of X, centroid) runs the fp32 fmaf chain in k order — the same
// bits as the MFMA kernel and the oracle.
#define SS_ROWS 16
#define SS_UNR 12
template <int NQ>
__global__ __launch_bounds__(256) void k_scores_small(const float* __restrict__ X, int n, int ldx, const float* __restrict__ C, int nc,
                                                     int d, float* __restrict__ S, int64_t lds_) {
    extern __shared__ __attribute__((aligned(16))) float ss_mem[];
    const int rs = d + 4;                       // LDS row stride of the centroid tile (floats): 16-byte aligned rows
    float* ss_c = ss_mem;                       // [SS_ROWS][rs]
    float* ss_x = ss_mem + SS_ROWS * rs;        // [NQ][d]
    const int t = threadIdx.x;
    const int c0 = blockIdx.x * SS_ROWS;
    const int row = t >> 4, l16 = t & 15, d4 = d >> 2;
    const int crow = c0 + row < nc ? c0 + row : nc - 1;
    const float4* src = reinterpret_cast<const float4*>(C + (int64_t)crow * d);
    for (int base = 0; base < d4; base += 16 * SS_UNR) {
        float4 v[SS_UNR];
#pragma unroll
        for (int i = 0; i < SS_UNR; i++) { const int k4 = base + l16 + 16 * i; v[i] = k4 < d4 ? src[k4] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int i = 0; i < SS_UNR; i++) { const int k4 = base + l16 + 16 * i; if (k4 < d4) *reinterpret_cast<float4*>(ss_c + row * rs + 4 * k4) = v[i]; }
    }
    for (int e = t; e < NQ * d; e += 256) { const int q = e / d, k = e - q * d; ss_x[e] = q < n ? X[(int64_t)q * ldx + k] : 0.0f; }
    __syncthreads();
    if (t < SS_ROWS * NQ) {
        const int c = t & (SS_ROWS - 1), q = t >> 4;
        const float* cr = ss_c + c * rs;
        const float* xr = ss_x + q * d;
        float acc = 0.0f;
#pragma unroll 4
        for (int k4 = 0; k4 < d4; k4++) {
            const float4 v = *reinterpret_cast<const float4*>(cr + 4 * k4), x = *reinterpret_cast<const float4*>(xr + 4 * k4);
            acc = __fmaf_rn(x.x, v.x, acc); acc = __fmaf_rn(x.y, v.y, acc); acc = __fmaf_rn(x.z, v.z, acc); acc = __fmaf_rn(x.w, v.w, acc);
        }
        if (c0 + c < nc && q < n) S[(int64_t)q * lds_ + c0 + c] = acc;
    }
}

void launch_gemm_exact_scores(const void* X, int x_f16, int64_t n, int ldx, const float* C, int nc, int d,
                              float* S, int64_t lds_, hipStream_t st) {
    if (n <= 0 || nc <= 0) return;
    if (!x_f16 && n <= 8 && d % 4 == 0 && aligned16(C) && (size_t)(SS_ROWS * (d + 4) + 8 * d) * 4 <= 160 * 1024 - 1024) {
        const unsigned g = (unsigned)((nc + SS_ROWS - 1) / SS_ROWS);
        const float* Xf = (const float*)X;
        const int NQ = n == 1 ? 1 : n == 2 ? 2 : n <= 4 ? 4 : 8;
        const size_t shm = (size_t)(SS_ROWS * (d + 4) + NQ * d) * 4;
        auto kern = NQ == 1 ? k_scores_small<1> : NQ == 2 ? k_scores_small<2> : NQ == 4 ? k_scores_small<4> : k_scores_small<8>;
        static DevSize big;
        if (shm > 48 * 1024)
            big.grow(shm, [&] {
                (void)hipFuncSetAttribute((const void*)k_scores_small<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
                (void)hipFuncSetAttribute((const void*)k_scores_small<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
                (void)hipFuncSetAttribute((const void*)k_scores_small<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
                (void)hipFuncSetAttribute((const void*)k_scores_small<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
            });
        hipLaunchKernelGGL(kern, dim3(g), dim3(256), shm, st, Xf, (int)n, ldx, C, nc, d, S, lds_);
        return;
    }
    dim3 grid((nc + 127) / 128, (unsigned)((n + 127) / 128));
    int xv = aligned16(X) && (ldx % 8 == 0);
    int cv = aligned16(C) && (d % 4 == 0);
    // measurement builds: RSX_COARSE_SPREAD = dynamic LDS (KiB) requested on top of the kernel's 34 KB of tiles, so that only one
    // workgroup fits a CU and the dispatcher has to spread a grid of ~#CUs workgroups over all CUs (profiles/r05_coarse_gemm.md)
    static const int spread_kib = measure_env("RSX_COARSE_SPREAD", 0);
    const size_t dyn = (size_t)spread_kib * 1024;
    if (dyn > 0) {
        static DevOnce once;
        once.once([&] {
            (void)hipFuncSetAttribute((const void*)k_gemm_exact<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            (void)hipFuncSetAttribute((const void*)k_gemm_exact<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        });
    }
    if (x_f16) hipLaunchKernelGGL((k_gemm_exact<true, false>), grid, dim3(256), dyn, st, X, n, ldx, xv, C, nc, d, cv, S, lds_, (uint64_t*)nullptr, 0);
    else hipLaunchKernelGGL((k_gemm_exact<false, false>), grid, dim3(256), dyn, st, X, n, ldx, xv, C, nc, d, cv, S, lds_, (uint64_t*)nullptr, 0);
}

void launch_gemm_exact_argmax(const void* X, int x_f16, int64_t n, int ldx, const float* C, int nc, int d,
                              uint64_t* partial, int32_t* assign, float* best, hipStream_t st) {
    if (n <= 0 || nc <= 0) return;
    // grid.y is limited to 65535 row tiles per launch
    const int64_t rows_per_launch = 65535ll * 128;
    int ct = (nc + 127) / 128;
    int xv = aligned16(X) && (ldx % 8 == 0);
    int cv = aligned16(C) && (d % 4 == 0);
    size_t esz = x_f16 ? 2 : 4;
    for (int64_t r0 = 0; r0 < n; r0 += rows_per_launch) {
        int64_t nr = n - r0 < rows_per_launch ? n - r0 : rows_per_launch;
        dim3 grid(ct, (unsigned)((nr + 127) / 128));
        const void* Xp = (const char*)X + (size_t)r0 * ldx * esz;
        if (x_f16) hipLaunchKernelGGL((k_gemm_exact<true, true>), grid, dim3(256), 0, st, Xp, nr, ldx, xv, C, nc, d, cv, (float*)nullptr, (int64_t)0, partial + r0 * 2 * ct, 2 * ct);
        else hipLaunchKernelGGL((k_gemm_exact<false, true>), grid, dim3(256), 0, st, Xp, nr, ldx, xv, C, nc, d, cv, (float*)nullptr, (int64_t)0, partial + r0 * 2 * ct, 2 * ct);
    }
    hipLaunchKernelGGL(k_argmax_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, n, 2 * ct, assign, best);
}

// =======================================================================================
// Flat scan GEMM (fp16 MFMA)
// =======================================================================================
template <bool XF16>
__device__ inline uint4 load8_as_half(const void* X, int64_t row, int ld, int t0) {
    if (XF16) {
        return *reinterpret_cast<const uint4*>((const __half*)X + row * ld + t0);
    } else {
        const float* p = (const float*)X + row * ld + t0;
        float4 f0 = *reinterpret_cast<const float4*>(p);
        float4 f1 = *reinterpret_cast<const float4*>(p + 4);
        union { uint4 u; __half h[8]; } c;
        c.h[0] = __float2half_rn(f0.x); c.h[1] = __float2half_rn(f0.y);
        c.h[2] = __float2half_rn(f0.z); c.h[3] = __float2half_rn(f0.w);
        c.h[4] = __float2half_rn(f1.x); c.h[5] = __float2half_rn(f1.y);
        c.h[6] = __float2half_rn(f1.z); c.h[7] = __float2half_rn(f1.w);
        return c.u;
    }
}

#define FG_STRIDE 144  // bytes per LDS row: 64 halfs + 16 B pad -> conflict-free ds_read_b128

// FILTER = false: temp[q, v0 + col] = score (one chunk, 2-D grid: x = query tile, y = db tile).
// FILTER = true : ONE launch over all remaining db tiles (1-D grid); instead of storing scores the epilogue
//                 appends the keys that beat the query's running K'-th best key (tau, from the chunks
//                 already selected) to a per-query candidate buffer.  The 1-D block id is decoded so that
//                 the QT query tiles of one db tile run on the same XCD (b % 8) back to back: the db tile is
//                 pulled from HBM once and re-read from that XCD's L2.
struct FlatFilterArgs {
    const uint64_t* tau; int64_t tau_stride;     // tau[q * tau_stride]
    uint64_t* cand; unsigned long long* cand_cnt; int cand_cap;
    int nq; int qt; int64_t ntiles;              // real queries, query tiles, db tiles in [v0, v0 + nv)
    int walk;                                    // k_flat_gemm2: persistent workgroups, fixed query tile, walking db tiles
    // k_flat_gemm, unfiltered form only (launch_coarse_approx): grid rows >= cv_y0 do not multiply — they widen cv_n fp16 values at cv_src to
    // fp32 at cv_dst (the query batch's fp32 copy the later kernels read, riding in this launch instead of one of its own)
    const __half* cv_src; float* cv_dst; int64_t cv_n; int cv_y0;
};

template <bool XF16, bool FILTER>
__global__ __launch_bounds__(256) void k_flat_gemm(const __half* Q16, const void* X, int64_t v0, int64_t nv, int ld,
                                                   const float* bias, float* temp, int64_t tstride, FlatFilterArgs F) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 128 * FG_STRIDE];
    __shared__ uint64_t s_tau[128];
    unsigned char* As = smem;                    // queries tile
    unsigned char* Bs = smem + 128 * FG_STRIDE;  // db rows tile
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    int64_t q0, vt0;
    if (FILTER) {
        const int64_t b = blockIdx.x;
        const int64_t i = b >> 3;
        const int64_t vt = (b & 7) + 8 * (i / F.qt);
        if (vt >= F.ntiles) return;
        q0 = (i % F.qt) * 128;
        vt0 = vt * 128;
        if (tid < 128) s_tau[tid] = (q0 + tid < F.nq) ? F.tau[(q0 + tid) * F.tau_stride] : ~0ull;
    } else {
        if (F.cv_n > 0 && (int)blockIdx.y >= F.cv_y0) {
            const int64_t nb = (int64_t)gridDim.x * (gridDim.y - F.cv_y0), b = (int64_t)(blockIdx.y - F.cv_y0) * gridDim.x + blockIdx.x;
            for (int64_t i8 = b * 256 + tid; i8 < F.cv_n / 8; i8 += nb * 256) {
                const uint4 u = *reinterpret_cast<const uint4*>(F.cv_src + 8 * i8);
                const __half* hv = reinterpret_cast<const __half*>(&u);
                *reinterpret_cast<float4*>(F.cv_dst + 8 * i8) = make_float4(__half2float(hv[0]), __half2float(hv[1]), __half2float(hv[2]), __half2float(hv[3]));
                *reinterpret_cast<float4*>(F.cv_dst + 8 * i8 + 4) = make_float4(__half2float(hv[4]), __half2float(hv[5]), __half2float(hv[6]), __half2float(hv[7]));
            }
            return;
        }
        q0 = (int64_t)blockIdx.x * 128;
        vt0 = (int64_t)blockIdx.y * 128;  // column offset inside this chunk
    }

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

    uint4 ra[4], rb[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        int p = tid + 256 * u, row = p >> 3, slot = p & 7;
        ra[u] = *reinterpret_cast<const uint4*>(Q16 + (q0 + row) * ld + slot * 8);
        rb[u] = load8_as_half<XF16>(X, v0 + vt0 + row, ld, slot * 8);
    }
    for (int k0 = 0; k0 < ld; k0 += 64) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int p = tid + 256 * u, row = p >> 3, slot = p & 7;
            *reinterpret_cast<uint4*>(As + row * FG_STRIDE + slot * 16) = ra[u];
            *reinterpret_cast<uint4*>(Bs + row * FG_STRIDE + slot * 16) = rb[u];
        }
        __syncthreads();
        if (k0 + 64 < ld) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int p = tid + 256 * u, row = p >> 3, slot = p & 7;
                ra[u] = *reinterpret_cast<const uint4*>(Q16 + (q0 + row) * ld + k0 + 64 + slot * 8);
                rb[u] = load8_as_half<XF16>(X, v0 + vt0 + row, ld, k0 + 64 + slot * 8);
            }
        }
        const int li = lane & 31, kh = lane >> 5;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            half8 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                a[t] = *reinterpret_cast<const half8*>(As + (wr * 64 + t * 32 + li) * FG_STRIDE + (2 * s + kh) * 16);
                b[t] = *reinterpret_cast<const half8*>(Bs + (wc * 64 + t * 32 + li) * FG_STRIDE + (2 * s + kh) * 16);
            }
#pragma unroll
            for (int ti = 0; ti < 2; ti++)
#pragma unroll
                for (int tj = 0; tj < 2; tj++)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
        }
    }
    // A index i = query, B index j = db row
    const int lj = lane & 31, lh = lane >> 5;
    if (!FILTER) {
#pragma unroll
        for (int tj = 0; tj < 2; tj++) {
            int64_t col = vt0 + wc * 64 + tj * 32 + lj;
            float bv = (bias && col < nv) ? bias[v0 + col] : 0.0f;
#pragma unroll
            for (int ti = 0; ti < 2; ti++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    int64_t qrow = q0 + wr * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (col < nv) temp[qrow * tstride + col] = acc[ti][tj][r] + bv;
                }
        }
    } else {
        float bv[2]; int64_t colv[2];
#pragma unroll
        for (int tj = 0; tj < 2; tj++) {
            colv[tj] = vt0 + wc * 64 + tj * 32 + lj;
            bv[tj] = (bias && colv[tj] < nv) ? bias[v0 + colv[tj]] : 0.0f;
        }
#pragma unroll
        for (int ti = 0; ti < 2; ti++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ql = wr * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;   // query inside the tile
                const uint64_t tau = s_tau[ql];          // ~0 for padding queries: nothing passes
                const float tf = key_score(tau);
#pragma unroll
                for (int tj = 0; tj < 2; tj++) {
                    const float sc = acc[ti][tj][r] + bv[tj];
                    // cheap wave-level reject on the float score; exact decision on the full key
                    if (!__any(sc >= tf && colv[tj] < nv)) continue;
                    const uint64_t key = (colv[tj] < nv) ? make_key(sc, (uint32_t)(v0 + colv[tj])) : 0ull;
                    const bool pass = key > tau;
                    const uint64_t mask = __ballot(pass);
                    const uint64_t mine = lh ? (mask >> 32) : (mask & 0xffffffffull);   // my half-wave = my query
                    if (mine) {
                        unsigned long long base = 0;
                        const int leader = (__ffsll((unsigned long long)mine) - 1) + 32 * lh;
                        if (lane == leader) base = atomicAdd(&F.cand_cnt[(int64_t)(q0 + ql) * CCS], (unsigned long long)__popcll(mine));
                        base = __shfl(base, leader);
                        const unsigned long long slot = base + __popcll(mine & ((1ull << lj) - 1ull));
                        if (pass && slot < (unsigned long long)F.cand_cap) F.cand[(q0 + ql) * F.cand_cap + slot] = key;
                    }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------
// k_flat_gemm2: the large-batch Flat scan.  256 queries x 256 db rows per workgroup, 8 waves (2 x 4, each
// 128 x 64 = 4 x 2 MFMA 32x32x16 tiles -> 6 ds_read_b128 per 8 MFMAs), BK = 64, two 64 KiB LDS stages
// filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), one barrier per K step.
//
// LDS layout per operand: row R at R*128 B; its 16-byte chunk c sits in slot c ^ ((R >> 1) & 7).  A DMA
// instruction lands 64 lanes x 16 B linearly (8 rows), so lane (r8, p) FETCHES chunk p ^ swz(R); the
// MFMA operand reads (16 lanes of a ds_read_b128 group = 16 different rows, same chunk) then hit 16
// different 16-byte slots of the 256-byte bank window: conflict-free without padding.
// ---------------------------------------------------------------------------------------
#define FG2_STAGE 65536
#define FG2_QCAP 192        // entries of a wave's queue of passing keys (filtered epilogue); flushed beyond QCAP - 128

// Cache policy of the LDS-DMA row streams (aux bits of the load; 2 = nt, "streamed once"), measured per kernel on one box
// (profiles/r04_nt_loads.md): the IVF-Flat / small-batch row stream is read by ONE workgroup once per launch — nt: +5 % at
// nprobe 32, +1 % at nprobe 128; a Flat db tile is re-read by the 4 workgroups that walk it on one XCD — nt: -12 %; the
// IVF-PQ code stream is re-read by sibling query groups out of the L2 — nt: -9 % (k_pq_rot.hip keeps the default policy).
#ifndef RSX_NT_FLAT
#define RSX_NT_FLAT 0
#endif
#ifndef RSX_NT_LIST
#define RSX_NT_LIST 2
#endif
template <int AUX = 0>
__device__ __forceinline__ void fg2_dma16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, AUX);
}

template <bool FILTER>
__global__ __launch_bounds__(512) void k_flat_gemm2(const __half* Q16, int nq_pad, const __half* X, int64_t v0, int64_t nv,
                                                    int ld, const float* bias, float* temp, int64_t tstride,
                                                    FlatFilterArgs F) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fg2_smem[];   // 2 stages x (A 32 KiB | B 32 KiB), tau[2][256]
    uint64_t* s_tau = reinterpret_cast<uint64_t*>(fg2_smem + 2 * FG2_STAGE);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr0 = __builtin_amdgcn_readfirstlane(w >> 2), wc0 = __builtin_amdgcn_readfirstlane(w & 3);
    // Work items of a workgroup = (query tile, db tile) pairs run through ONE software pipeline of K steps:
    //  !FILTER       : one pair per workgroup (2-D grid).
    //  FILTER, walk  : PERSISTENT workgroups (one per CU).  Workgroup b sits on XCD b % 8; slot s = b / 8 of that XCD keeps
    //                  query tile s % qt and walks the db tiles  x + 8 (s / qt) + 8 (S / qt) n,  n = 0, 1, ...  The qt slots
    //                  that share s / qt walk the SAME db tiles at the same pace, so a db tile comes from HBM once per XCD
    //                  and its other qt - 1 readers hit that XCD's L2 (PMC FETCH_SIZE: 15.5 GB per 10M x 768 launch = the
    //                  db itself, against 67.6 GB for the pass form below); the query tiles stay L2 resident.
    //  FILTER, !walk : the workgroup owns ONE db tile and passes every query tile over it (small launches).
    const bool walk = FILTER && F.walk != 0;
    int qi0, nitems;
    int64_t vt0, vt_step;
    if (!FILTER) { vt0 = (int64_t)blockIdx.y * 256; vt_step = 0; qi0 = (int)blockIdx.x; nitems = 1; }
    else if (!walk) { vt0 = (int64_t)blockIdx.x * 256; vt_step = 0; qi0 = 0; nitems = F.qt; }
    else {
        const int x = blockIdx.x & 7, s = blockIdx.x >> 3, ngq = (int)(gridDim.x >> 3) / F.qt;
        qi0 = s % F.qt;
        const int64_t t0 = x + 8 * (s / F.qt), tstep = 8 * (int64_t)ngq;
        vt0 = t0 * 256; vt_step = tstep * 256;
        nitems = t0 < F.ntiles ? (int)((F.ntiles - 1 - t0) / tstep) + 1 : 0;
        if (nitems == 0) return;
    }
    float* s_tf = reinterpret_cast<float*>(s_tau + 512);                        // tf[2][256]: the thresholds as floats
    uint64_t* s_wq_key = reinterpret_cast<uint64_t*>(s_tf + 512);               // (FILTER) per-wave queue of passing keys [8][FG2_QCAP] ...
    uint16_t* s_wq_q = reinterpret_cast<uint16_t*>(s_wq_key + 8 * FG2_QCAP);    // ... and their queries inside the tile
    if (FILTER && tid < 256) {
        const int64_t qn = (int64_t)qi0 * 256 + tid;
        const uint64_t t = (qn < F.nq) ? F.tau[qn * F.tau_stride] : ~0ull;
        s_tau[(qi0 & 1) * 256 + tid] = t;
        s_tf[(qi0 & 1) * 256 + tid] = (qn < F.nq) ? key_score(t) : __builtin_inff();
    }

    // DMA sources: wave w copies blocks [4w, 4w + 4) (8 rows each) of both operands, every stage.  The db rows are
    // addressed from the walking tile base on the fly (one v_mad_u64_u32 per piece), clamped to the last row.
    const char* srcA[4];
    const char* xb[2];                           // X + v0 rows + the swizzled chunk of this lane (j even / odd)
    const int r8 = lane >> 3;
    {
        const int p = lane & 7;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int R = (4 * w + j) * 8 + r8;
            const int c = p ^ ((R >> 1) & 7);
            srcA[j] = reinterpret_cast<const char*>(Q16 + ((int64_t)qi0 * 256 + R) * ld) + c * 16;   // nq_pad % 256 == 0
            if (j < 2) xb[j] = reinterpret_cast<const char*>(X + v0 * ld) + c * 16;                    // c depends on j & 1 only
        }
    }
    const int KT = ld / 64;
    const int G = nitems * KT;                   // K steps of all work items, one software pipeline
    const int64_t a_step = walk ? 0 : (int64_t)256 * ld * 2;     // the next item's query tile ...
    const int64_t i_vstep = vt_step;                              // ... or db tile
    const uint32_t ld2 = (uint32_t)ld * 2u;
    const int64_t vlast = nv - 1;
    int i_kt = 0; int64_t i_aoff = 0, i_vt = vt0;    // position of the next stage to issue
    auto issue = [&](int buf, bool advance) {
        unsigned char* sa = fg2_smem + buf * FG2_STAGE + (4 * w) * 1024;
        unsigned char* sb = sa + 32768;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int64_t xrow = i_vt + ((4 * w + j) * 8 + r8);
            xrow = xrow > vlast ? vlast : xrow;
            fg2_dma16(srcA[j] + i_aoff + (int64_t)i_kt * 128, sa + j * 1024);
            fg2_dma16<RSX_NT_FLAT>(xb[j & 1] + (uint64_t)(uint32_t)xrow * ld2 + (int64_t)i_kt * 128, sb + j * 1024);
        }
        // branch-free advance (a branch here would split the scheduling region the interleave below pins)
        i_kt += advance ? 1 : 0;
        const bool wrap = i_kt == KT;
        i_kt = wrap ? 0 : i_kt;
        i_aoff += wrap ? a_step : 0;
        i_vt += wrap ? i_vstep : 0;
    };

    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

    const int li = lane & 31, kh = lane >> 5;
    const int sw = (li >> 1) & 7;
    int offs[4];
#pragma unroll
    for (int s = 0; s < 4; s++) offs[s] = li * 128 + (((2 * s + kh) ^ sw) << 4);

    issue(0, G > 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // operand fragments are double-buffered in registers: the reads of k sub-step s+1 are in flight while
    // the 8 MFMAs of sub-step s issue
    half8 fa[2][4], fb[2][2];
#define FG2_LOAD_FRAGS(BUF, S, SLOT)                                                                       \
    {                                                                                                        \
        const unsigned char* As_ = fg2_smem + (BUF) * FG2_STAGE + wr0 * (128 * 128);                         \
        const unsigned char* Bs_ = fg2_smem + (BUF) * FG2_STAGE + 32768 + wc0 * (64 * 128);                  \
        _Pragma("unroll") for (int t = 0; t < 4; t++)                                                        \
            fa[SLOT][t] = *reinterpret_cast<const half8*>(As_ + t * 4096 + offs[S]);                         \
        _Pragma("unroll") for (int t = 0; t < 2; t++)                                                        \
            fb[SLOT][t] = *reinterpret_cast<const half8*>(Bs_ + t * 4096 + offs[S]);                         \
    }
#define FG2_MFMAS(SLOT)                                                                                     \
    _Pragma("unroll") for (int ti = 0; ti < 4; ti++)                                                         \
        _Pragma("unroll") for (int tj = 0; tj < 2; tj++)                                                     \
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[SLOT][ti], fb[SLOT][tj], acc[ti][tj], 0, 0, 0);
    FG2_LOAD_FRAGS(0, 0, 0)
    const int lj0 = lane & 31, lh0 = lane >> 5;
    int kt = 0, qi = qi0;
    int64_t vt = vt0;                            // db tile of the item being computed (the issue position runs ahead)
    for (int g = 0; g < G; g++) {
        // sub-step 0 carries the DMA of the NEXT stage (into the buffer every wave left at the last barrier);
        // the last step re-fetches its own stage into the idle buffer so the body stays branch-free.
        issue((g + 1) & 1, g + 2 < G);
        FG2_LOAD_FRAGS(g & 1, 1, 1)
        FG2_MFMAS(0)
        // pin the interleave: 1 MFMA : 1 DMA (+ 1 fragment read) — each DMA issue and LDS latency is then
        // covered by a 32-cycle matrix-pipe slot instead of stalling in front of the MFMAs
#pragma unroll
        for (int i = 0; i < 6; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
        FG2_LOAD_FRAGS(g & 1, 2, 0)
        FG2_MFMAS(1)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
        FG2_LOAD_FRAGS(g & 1, 3, 1)
        FG2_MFMAS(0)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
        FG2_MFMAS(1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of stage g+1 have landed
        __syncthreads();                                    // everyone: done with stage g, stage g+1 complete
        if (++kt < KT) {
            FG2_LOAD_FRAGS((g + 1) & 1, 0, 0)
            continue;
        }
        kt = 0;
        // ---- end of a work item: the 256 x 256 scores of (query tile qi, db tile vt) are complete
        const int64_t q0 = (int64_t)qi * 256;
        // launder the lane coordinates: everything derived from them below would otherwise be hoisted out of
        // the K loop as loop-invariant and held in ~70 VGPRs across the MFMA phase (-> spills)
        int lj = lj0, lh = lh0, wr = wr0, wc = wc0;
        asm volatile("" : "+v"(lj), "+v"(lh), "+s"(wr), "+s"(wc));
        if (!FILTER) {
#pragma unroll
            for (int tj = 0; tj < 2; tj++) {
                int64_t col = vt + wc * 64 + tj * 32 + lj;
                float bv = (bias && col < nv) ? bias[v0 + col] : 0.0f;
#pragma unroll
                for (int ti = 0; ti < 4; ti++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        int64_t qrow = q0 + wr * 128 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (col < nv && qrow < nq_pad) temp[qrow * tstride + col] = acc[ti][tj][r] + bv;
                    }
            }
        } else {
            const uint64_t* taus = s_tau + (qi & 1) * 256;
            const float* tfs = s_tf + (qi & 1) * 256;
            float bv[2]; int64_t colv[2];
#pragma unroll
            for (int tj = 0; tj < 2; tj++) {
                colv[tj] = vt + wc * 64 + tj * 32 + lj;
                bv[tj] = (bias && colv[tj] < nv) ? bias[v0 + colv[tj]] : 0.0f;
            }
            const bool in0 = colv[0] < nv, in1 = colv[1] < nv;
            // Round 6: the keys that pass go to a queue of this wave in LDS (ballot + prefix: no atomics) and leave it 64 at a time —
            // ONE atomic round trip per flush for all of them.  Before, every row pair with a hit paid its own atomicAdd-with-return
            // in the middle of the epilogue (~13 serial L2 round trips per tile at the reference's n_docs = 1000: 3.4 ms of a 20 ms batch).
            // Slots inside a candidate row come out in a different order; the selection behind does not depend on it (keys are unique).
            uint64_t* wq_key = s_wq_key + w * FG2_QCAP;
            uint16_t* wq_q = s_wq_q + w * FG2_QCAP;
            int qn = 0;                                   // wave-uniform
            auto flush = [&]() {
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < qn; i += 64) {
                    const uint64_t key = wq_key[i];
                    const int64_t qg = q0 + wq_q[i];
                    const unsigned long long slot = atomicAdd(&F.cand_cnt[qg * CCS], 1ull);
                    if (slot < (unsigned long long)F.cand_cap) F.cand[qg * F.cand_cap + slot] = key;
                }
                __builtin_amdgcn_wave_barrier();
                qn = 0;
            };
#pragma unroll
            for (int ti = 0; ti < 4; ti++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int ql = wr * 128 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;   // query inside the tile
                    // fast reject of the row pair on the float threshold (+inf for padding queries);
                    // the exact decision below is on the full key
                    const float s0 = acc[ti][0][r] + bv[0], s1 = acc[ti][1][r] + bv[1];
                    const float tf = tfs[ql];
                    if (!__any((in0 && s0 >= tf) || (in1 && s1 >= tf))) continue;
                    const uint64_t tau = taus[ql];
#pragma unroll
                    for (int tj = 0; tj < 2; tj++) {
                        const float sc = tj ? s1 : s0;
                        const uint64_t key = (colv[tj] < nv) ? make_key(sc, (uint32_t)(v0 + colv[tj])) : 0ull;
                        const bool pass = key > tau;
                        const uint64_t mask = __ballot(pass);
                        if (mask) {
                            const int pos = qn + (int)__popcll(mask & ((1ull << lane) - 1ull));
                            if (pass) { wq_key[pos] = key; wq_q[pos] = (uint16_t)ql; }
                            qn += (int)__popcll(mask);
                        }
                    }
                    if (qn > FG2_QCAP - 128) flush();
                }
            if (qn) flush();
            // thresholds of the next pass (read after at least one more barrier)
            if (!walk && qi + 1 < nitems && tid < 256) {
                const int64_t qn = (int64_t)(qi + 1) * 256 + tid;
                const uint64_t t = (qn < F.nq) ? F.tau[qn * F.tau_stride] : ~0ull;
                s_tau[((qi + 1) & 1) * 256 + tid] = t;
                s_tf[((qi + 1) & 1) * 256 + tid] = (qn < F.nq) ? key_score(t) : __builtin_inff();
            }
            if (KT == 1) __syncthreads();            // no K-loop barrier would order them before the next epilogue
        }
        qi += walk ? 0 : 1;
        vt += vt_step;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;
        if (g + 1 < G) FG2_LOAD_FRAGS((g + 1) & 1, 0, 0)
    }
#undef FG2_MFMAS
#undef FG2_LOAD_FRAGS
}

static const size_t FG2_SHM = 2 * FG2_STAGE + 2 * 256 * 8 + 2 * 256 * 4;
static const size_t FG2_SHM_FILTER = FG2_SHM + 8 * FG2_QCAP * (8 + 2);
template <bool FILTER>
static bool fg2_ready() {
    static DevOnce once;
    static std::atomic<int> failed{0};
    once.once([&] {
        if (hipFuncSetAttribute((const void*)k_flat_gemm2<FILTER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FILTER ? FG2_SHM_FILTER : FG2_SHM)) != hipSuccess) failed = 1;
    });
    return !failed;
}
// the 256 x 256 LDS-DMA kernel needs fp16 storage, K a multiple of 64 and enough queries to fill its tile
static bool fg2_applies(int nq_pad, int x_f16, int ld) {
    static const int off = measure_env("RSX_FLAT_GEMM_V1", 0);
    return !off && x_f16 && nq_pad % 256 == 0 && ld % 64 == 0;
}

// Coarse quantiser, fast form (round 6): APPROXIMATE scores of all centroids from the fp16 MFMA tile kernel — 6.4 GFLOP in ~15 us
// where the bit-exact f32-input chain (k_gemm_exact) needs 65; k_coarse_pick (k_select.hip) re-scores the few centroids that can be
// among a query's nprobe best with the exact chain.  Q16: [nq_pad][ld] (nq_pad % 128 == 0), C16: [round_up(nlist, 128)][ld] fp16
// copies, zero padded; S: [nq_pad][lds_] fp32.
// widen_n > 0: Q16 is the caller's own fp16 batch (contiguous rows of ld values, widen_n = rows x ld, a multiple of 8): eight extra grid rows
// write its fp32 copy to Q32 in the same launch.
void launch_coarse_approx(const __half* Q16, int64_t nq_pad, const __half* C16, int nlist, int ld, float* S, int64_t lds_, hipStream_t st,
                          float* Q32, int64_t widen_n) {
    if (nq_pad <= 0 || nlist <= 0) return;
    FlatFilterArgs F{};
    const unsigned gy = (unsigned)((nlist + 127) / 128);
    if (widen_n > 0 && Q32) { F.cv_src = Q16; F.cv_dst = Q32; F.cv_n = widen_n; F.cv_y0 = (int)gy; }
    hipLaunchKernelGGL((k_flat_gemm<true, false>), dim3((unsigned)(nq_pad / 128), gy + (F.cv_n > 0 ? 8u : 0u)), dim3(256), 0, st, Q16, (const void*)C16,
                       (int64_t)0, (int64_t)nlist, ld, (const float*)nullptr, S, lds_, F);
}

void launch_flat_gemm(const __half* Q16, int nq_pad, const void* X, int x_f16, int64_t v0, int64_t nv, int ld,
                      const float* bias, float* temp, int64_t tstride, hipStream_t st) {
    if (nv <= 0 || nq_pad <= 0) return;
    FlatFilterArgs F{};
    if (fg2_applies(nq_pad, x_f16, ld) && fg2_ready<false>()) {
        dim3 g2((unsigned)(nq_pad / 256), (unsigned)((nv + 255) / 256));
        hipLaunchKernelGGL((k_flat_gemm2<false>), g2, dim3(512), FG2_SHM, st, Q16, nq_pad, (const __half*)X, v0, nv, ld, bias, temp, tstride, F);
        return;
    }
    dim3 grid(nq_pad / 128, (unsigned)((nv + 127) / 128));
    if (x_f16) hipLaunchKernelGGL((k_flat_gemm<true, false>), grid, dim3(256), 0, st, Q16, X, v0, nv, ld, bias, temp, tstride, F);
    else hipLaunchKernelGGL((k_flat_gemm<false, false>), grid, dim3(256), 0, st, Q16, X, v0, nv, ld, bias, temp, tstride, F);
}

// One launch over db rows [v0, v0 + nv): candidates beating tau[q] go to cand[q][0..cap) (count in cand_cnt[q]).
void launch_flat_gemm_filter(const __half* Q16, int nq_pad, int nq, const void* X, int x_f16, int64_t v0, int64_t nv, int ld,
                             const float* bias, const uint64_t* tau, int64_t tau_stride, uint64_t* cand,
                             unsigned long long* cand_cnt, int cand_cap, hipStream_t st) {
    if (nv <= 0 || nq_pad <= 0) return;
    FlatFilterArgs F{};
    F.tau = tau; F.tau_stride = tau_stride; F.cand = cand; F.cand_cnt = cand_cnt; F.cand_cap = cand_cap;
    if (fg2_applies(nq_pad, x_f16, ld) && fg2_ready<true>()) {
        F.nq = nq; F.qt = nq_pad / 256; F.ntiles = (nv + 255) / 256;
        // persistent walking workgroups (see the kernel) once every slot has a few db tiles to walk; RSX_FG2_WALK=0: off (A/B)
        static const int walk_on = measure_env("RSX_FG2_WALK", 1);
        static int ncu_of[64] = {};
        int& ncu = ncu_of[cur_device()];
        if (ncu == 0) {
            int dev = 0; hipDeviceProp_t pr;
            ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
                      ? pr.multiProcessorCount : 256;
        }
        int slots = ((ncu / 8) / F.qt) * F.qt;                 // per XCD, a multiple of the query tiles
#ifdef FG2_SLOTS                                               // (A/B builds: fewer walking workgroups per XCD = a smaller L2 footprint)
        if (slots > FG2_SLOTS) slots = (FG2_SLOTS / F.qt) * F.qt;
#endif
        unsigned grid = (unsigned)F.ntiles;
        if (walk_on && slots > 0 && nv < ((int64_t)1 << 31) && F.ntiles >= (int64_t)4 * 8 * (slots / F.qt)) {
            F.walk = 1; grid = 8u * (unsigned)slots;
        }
        hipLaunchKernelGGL((k_flat_gemm2<true>), dim3(grid), dim3(512), FG2_SHM_FILTER, st, Q16, nq_pad, (const __half*)X, v0, nv, ld, bias,
                           (float*)nullptr, (int64_t)0, F);
        return;
    }
    F.nq = nq; F.qt = nq_pad / 128; F.ntiles = (nv + 127) / 128;
    int64_t groups = (F.ntiles + 7) / 8;                 // 8 db tiles (one per XCD) x qt query tiles each
    int64_t blocks = groups * F.qt * 8;
    dim3 grid((unsigned)blocks);
    if (x_f16) hipLaunchKernelGGL((k_flat_gemm<true, true>), grid, dim3(256), 0, st, Q16, X, v0, nv, ld, bias, (float*)nullptr, (int64_t)0, F);
    else hipLaunchKernelGGL((k_flat_gemm<false, true>), grid, dim3(256), 0, st, Q16, X, v0, nv, ld, bias, (float*)nullptr, (int64_t)0, F);
}

// =======================================================================================
// List scan (IVF-Flat; Flat at small batch)
// =======================================================================================
template <bool XF16>
__global__ __launch_bounds__(256) void k_list_scan(ListScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ls_smem[];
    const int qstride = (a.ld + 8) * 2;  // bytes; (ld+8)*2/16 is odd for ld % 16 == 0 -> conflict-free b128 reads
    unsigned char* Qs = ls_smem;
    int64_t* segoff = reinterpret_cast<int64_t*>(ls_smem + 16 * qstride);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = blockIdx.x;
    const int chunk = blockIdx.y;

    int64_t base, len;
    int np;
    int l = 0, pair0 = 0;
    if (a.flat_mode) {
        base = 0; len = a.flat_n;
        np = a.nq - 16 * g; if (np > 16) np = 16;
        if (np <= 0) return;
    } else {
        if (g >= *a.total_groups) return;
        int lo = 0, hi = a.nlist;  // largest l with group_off[l] <= g
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (a.group_off[mid] <= g) lo = mid; else hi = mid; }
        l = lo;
        int gi = g - a.group_off[l];
        int cnt = a.pair_off[l + 1] - a.pair_off[l];
        np = cnt - 16 * gi; if (np > 16) np = 16;
        pair0 = a.pair_off[l] + 16 * gi;
        base = a.list_base[l]; len = a.list_len[l];
    }
    const int64_t len_pad = (len + 15) & ~15ll;
    const int64_t c0 = (int64_t)chunk * a.chunk_rows;
    if (c0 >= len_pad) return;
    int64_t c1 = c0 + a.chunk_rows; if (c1 > len_pad) c1 = len_pad;

    // stage the group's queries (16 rows of ld halfs) and their score-buffer offsets
    for (int i = 0; i < 16; i++) {
        int64_t q = -1;
        if (i < np) {
            if (a.flat_mode) q = 16 * (int64_t)g + i;
            else q = a.pairs_sorted[pair0 + i] / a.nprobe;
        }
        for (int t = tid * 8; t < a.ld; t += 256 * 8) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q >= 0) v = *reinterpret_cast<const uint4*>(a.Q16 + q * a.ld + t);
            *reinterpret_cast<uint4*>(Qs + i * qstride + t * 2) = v;
        }
    }
    if (tid < 16) {
        int64_t off = 0;
        if (tid < np) {
            if (a.flat_mode) off = (16 * (int64_t)g + tid) * a.tstride;
            else {
                int pidx = a.pairs_sorted[pair0 + tid];
                int64_t q = pidx / a.nprobe; int j = pidx % a.nprobe;
                off = q * a.tstride + a.seg_start[q * (a.nprobe + 1) + j];
            }
        }
        segoff[tid] = off;
    }
    __syncthreads();

    const int lr = lane & 15, kg = lane >> 4;
    for (int64_t rb = c0 + w * 16; rb < c1; rb += 64) {
        const int64_t row = base + rb + lr;
        floatx4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 8
        for (int k0 = 0; k0 < a.ld; k0 += 32) {
            half8 qa = *reinterpret_cast<const half8*>(Qs + lr * qstride + (k0 + 8 * kg) * 2);
            uint4 xb = load8_as_half<XF16>(a.X, row, a.ld, k0 + 8 * kg);
            half8 xh = *reinterpret_cast<half8*>(&xb);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa, xh, acc, 0, 0, 0);
        }
        // C/D layout of 16x16 MFMA: reg r of lane l holds (row i = 4*(l>>4) + r, col j = l&15)
        //   i = query of the group, j = db row of the block
        const int64_t rloc = rb + lr;
        float bv = (a.bias && rloc < len) ? a.bias[base + rloc] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int qi = kg * 4 + r;
            if (qi < np) {
                float v = (rloc < len) ? acc[r] + bv : -__builtin_inff();
                a.temp[segoff[qi] + rloc] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// k_list_scan2: the same list-major scan for fp16 rows with the row stream staged through LDS by LDS-DMA.
// k_list_scan pulls 16 rows x 64 B per load instruction (half a cache line per row) through VGPRs and tops out
// at ~4.5 TB/s; here every wave owns a ring of LS2_D stages of 16 rows x 128 B (two 1 KiB DMA pieces, 8 rows x
// one full line each, XOR-swizzled like k_flat_gemm2's tiles) that runs LS2_D - 1 K steps ahead of the MFMAs,
// across the 16-row blocks of its chunk.  Waves never synchronise with each other after the queries are staged;
// the scores wait in registers (8 blocks x 4) and are stored once the DMA stream has drained, because loads and
// stores retire out of order with each other and would break the counted vmcnt waits.
// ---------------------------------------------------------------------------------------
#define LS2_D 6
#define LS2_NB 8            // 16-row blocks per wave per chunk -> chunk_rows = 4 waves x 16 x 8 = 512

// QT = 16-query tiles per group (round 3): with many probing queries per list (nlist 2048 / nprobe 128: 64 on average) groups of 16
// pass over every list four times — the rows are re-read from L2 / HBM for each pass (0.20 of HBM on unique bytes).  A group of
// 16 QT queries reads the row stream ONCE and issues QT x 2 MFMAs per K step against the same two row fragments.
// NW waves per workgroup, each with a ring of D stages.  64-query groups take 8 waves x 3 stages: their 97 KiB of queries leave room
// for one workgroup per CU only, and with four waves (one per SIMD) nothing overlapped a wave's DMA issue, LDS latency and MFMAs.
template <bool FILTER, int QT, int NW, int D>
__global__ __launch_bounds__(64 * NW) void k_list_scan2(ListScanArgs a) {
    constexpr int NQG = 16 * QT;
    constexpr int NT = 64 * NW;                  // threads
    constexpr int BR = 16 * NW;                  // rows between a wave's consecutive blocks
    extern __shared__ __attribute__((aligned(16))) unsigned char ls_smem[];
    const int qstride = (a.ld + 8) * 2;
    unsigned char* Qs = ls_smem;
    int64_t* segoff = reinterpret_cast<int64_t*>(ls_smem + NQG * qstride);  // [NQG] score-buffer offset | (FILTER) row column
    int64_t* sq = segoff + NQG;                                             // [NQG] query            (FILTER)
    uint64_t* stau = reinterpret_cast<uint64_t*>(sq + NQG);                 // [NQG] threshold key    (FILTER)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* ring = ls_smem + NQG * qstride + 24 * NQG + w * (D * 2048);
    int g = blockIdx.x;
    int chunk = blockIdx.y;

    int64_t base, len;
    int np;
    int pair0 = 0;
    if (a.flat_mode) {
        base = 0; len = a.flat_n;
        np = a.nq - NQG * g; if (np > NQG) np = NQG;
        if (np <= 0) return;
    } else {
        int l, gi;
        if (a.item_off) {           // 1-D grid over (list, chunk, group) items: the groups of a list chunk share an XCD and a moment
            if (!pq_decode_item(a.item_off, a.group_off, *a.total_items, a.nlist, l, gi, chunk)) return;
            g = a.group_off[l] + gi;
        } else {
            if (g >= *a.total_groups) return;
            int lo = 0, hi = a.nlist;  // largest l with group_off[l] <= g
            while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (a.group_off[mid] <= g) lo = mid; else hi = mid; }
            l = lo;
            gi = g - a.group_off[l];
        }
        int cnt = a.pair_off[l + 1] - a.pair_off[l];
        np = cnt - NQG * gi; if (np > NQG) np = NQG;
        pair0 = a.pair_off[l] + NQG * gi;
        base = a.list_base[l]; len = a.list_len[l];
    }
    const int64_t len_pad = (len + 15) & ~15ll;
    const int64_t c0 = (int64_t)chunk * (BR * LS2_NB);
    if (c0 >= len_pad) return;
    int64_t c1 = c0 + BR * LS2_NB; if (c1 > len_pad) c1 = len_pad;

    // stage the group's queries (NQG rows of ld halfs) and their score-buffer offsets.  Round 3: the query numbers first (one per
    // thread), then every thread's share of the NQG x ld / 8 sixteen-byte pieces with eight loads in flight — the loop used to
    // take the queries one after the other, an index load and a row load each: 2 round trips per query, 32 us per work item for 16
    // queries against the ~40 us its 512 rows take to stream.
    int32_t* sqn = reinterpret_cast<int32_t*>(ls_smem + NQG * qstride + 24 * NQG + NW * D * 2048);     // [NQG] query numbers
    if (tid < NQG) {
        int64_t q = -1;
        if (tid < np) q = a.flat_mode ? NQG * (int64_t)g + tid : a.pairs_sorted[pair0 + tid] / a.nprobe;
        sqn[tid] = (int32_t)q;
    }
    __syncthreads();
    {
        const int ppr = a.ld >> 3;                        // 16-byte pieces per query row
        const int npc = NQG * ppr;
        for (int p0 = tid; p0 < npc; p0 += NT * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int pc = p0 + u * NT;
                const int i = pc < npc ? pc / ppr : 0, t = pc < npc ? pc - i * ppr : 0;
                const int q = sqn[i];
                v[u] = make_uint4(0, 0, 0, 0);
                if (pc < npc && q >= 0) v[u] = *reinterpret_cast<const uint4*>(a.Q16 + (int64_t)q * a.ld + t * 8);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int pc = p0 + u * NT;
                if (pc >= npc) break;
                const int i = pc / ppr, t = pc - i * ppr;
                *reinterpret_cast<uint4*>(Qs + i * qstride + t * 16) = v[u];
            }
        }
    }
    if (tid < NQG) {
        int64_t off = 0;
        if (tid < np) {
            if (a.flat_mode) off = (NQG * (int64_t)g + tid) * a.tstride;
            else {
                int pidx = a.pairs_sorted[pair0 + tid];
                int64_t q = pidx / a.nprobe; int j = pidx % a.nprobe;
                off = q * a.tstride + ((!FILTER && a.pre_stride) ? (int64_t)j * a.pre_stride : a.seg_start[q * (a.nprobe + 1) + j]);
            }
        }
        segoff[tid] = off;
        if (FILTER) {
            int64_t q = 0; uint64_t t = ~0ull;      // padding slot: nothing passes
            if (tid < np) {
                q = a.flat_mode ? NQG * (int64_t)g + tid : a.pairs_sorted[pair0 + tid] / a.nprobe;
                t = a.tau_key[q * a.tau_stride];
                segoff[tid] = off - q * a.tstride;  // the column inside the query's row = the candidate's index
            }
            sq[tid] = q; stau[tid] = t;
        }
    }
    __syncthreads();

    // this wave's blocks: rows c0 + 16 w + BR i, i < nb
    const int64_t r0 = c0 + 16 * w;
    const int nb = r0 < c1 ? (int)((c1 - r0 + BR - 1) / BR) : 0;
    if (nb == 0) return;
    const int KT = a.ld >> 6;
    const int T = nb * KT;
    // DMA source: uniform block base + per-lane offset.  Piece j (0/1) = rows 8j..8j+7 of the block, one 128-B line each.
    const char* xb = reinterpret_cast<const char*>(reinterpret_cast<const __half*>(a.X) + (base + r0) * a.ld);
    uint32_t off0, off1;
    {
        const int r8 = lane >> 3, p = lane & 7;
        const int R0 = r8, R1 = 8 + r8;
        off0 = (uint32_t)R0 * (uint32_t)(a.ld * 2) + ((p ^ ((R0 >> 1) & 7)) << 4);
        off1 = (uint32_t)R1 * (uint32_t)(a.ld * 2) + ((p ^ ((R1 >> 1) & 7)) << 4);
    }
    const int64_t blk_bytes = (int64_t)BR * a.ld * 2;       // the wave's next block is BR rows further
    int i_kt = 0, i_slot = 0; int64_t i_boff = 0; int i_left = T;   // issue position
    auto issue = [&]() {
        const char* gsrc = xb + i_boff + (int64_t)i_kt * 128;
        unsigned char* dst = ring + i_slot * 2048;
        fg2_dma16<RSX_NT_LIST>(gsrc + off0, dst);
        fg2_dma16<RSX_NT_LIST>(gsrc + off1, dst + 1024);
        // branch-free advance; past the last step the last (valid) piece is fetched again into a slot nobody reads
        const bool adv = i_left > 1;
        i_left -= adv ? 1 : 0;
        i_kt += adv ? 1 : 0;
        const bool wrap = i_kt == KT;
        i_kt = wrap ? 0 : i_kt;
        i_boff += wrap ? blk_bytes : 0;
        i_slot = (i_slot + 1 == D) ? 0 : i_slot + 1;
    };
#pragma unroll
    for (int d = 0; d < D - 1; d++) issue();

    const int lr = lane & 15, kg = lane >> 4;
    const int swz = (lr >> 1) & 7;
    const int boff0 = lr * 128 + (((0 + kg) ^ swz) << 4), boff1 = lr * 128 + (((4 + kg) ^ swz) << 4);
    const unsigned char* qa_base = Qs + lr * qstride + (8 * kg) * 2;
    floatx4 acc[QT][LS2_NB];
#pragma unroll
    for (int t = 0; t < QT; t++)
#pragma unroll
        for (int i = 0; i < LS2_NB; i++) acc[t][i] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    int slot = 0;
#pragma unroll
    for (int i = 0; i < LS2_NB; i++) {
        if (i < nb) {
            for (int kt = 0; kt < KT; kt++) {
                issue();
                // all but the newest LS2_D - 1 stages (2 pieces each) have landed -> this stage is in LDS
                // (ring depth, round 5: 4 or 3 stages instead of 6 in the 16-query form — three workgroups per CU — change nothing, 2 instead of
                //  3 in the 64-query form costs 5 %: profiles/r05_ivfflat_wide.md)
                static_assert(D == 6 || D == 3, "the vmcnt literals below are 2 (D - 1)");
                if (D == 6) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                const unsigned char* bs = ring + slot * 2048;
                half8 x0 = *reinterpret_cast<const half8*>(bs + boff0);
                half8 x1 = *reinterpret_cast<const half8*>(bs + boff1);
#pragma unroll
                for (int t = 0; t < QT; t++) {
                    half8 q0 = *reinterpret_cast<const half8*>(qa_base + t * 16 * qstride + (kt * 64) * 2);
                    half8 q1 = *reinterpret_cast<const half8*>(qa_base + t * 16 * qstride + (kt * 64 + 32) * 2);
                    acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q0, x0, acc[t][i], 0, 0, 0);
                    acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q1, x1, acc[t][i], 0, 0, 0);
                }
                slot = (slot + 1 == D) ? 0 : slot + 1;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the tail pieces before any store
    // C/D layout of 16x16 MFMA: reg r of lane l holds (row i = 4*(l>>4) + r, col j = l&15): i = query, j = db row
#pragma unroll
    for (int i = 0; i < LS2_NB; i++) {
        if (i < nb) {
            const int64_t rloc = r0 + BR * i + lr;
            const float bv = (a.bias && rloc < len) ? a.bias[base + rloc] : 0.0f;
#pragma unroll
            for (int t = 0; t < QT; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int qi = t * 16 + kg * 4 + r;
                if (!FILTER) {
                    if (qi < np) a.temp[segoff[qi] + rloc] = (rloc < len) ? acc[t][i][r] + bv : -__builtin_inff();
                } else {
                    // lanes 16 kg .. 16 kg + 15 hold query qi's scores of the block's 16 rows
                    const uint64_t key = (rloc < len && qi < np) ? make_key(acc[t][i][r] + bv, (uint32_t)(segoff[qi] + rloc)) : 0ull;
                    const bool pass = key > stau[qi];
                    const uint64_t mask = __ballot(pass);
                    const uint64_t mine = (mask >> (16 * kg)) & 0xffffull;
                    if (mine) {   // one atomic per query per block
                        const int64_t q = sq[qi];
                        unsigned long long slot0 = 0;
                        const int leader = (__ffsll((unsigned long long)mine) - 1) + 16 * kg;
                        if (lane == leader) slot0 = atomicAdd(&a.cand_cnt[(int64_t)q * CCS], (unsigned long long)__popcll(mine));
                        slot0 = __shfl(slot0, leader);
                        const unsigned long long slot = slot0 + __popcll(mine & ((1ull << lr) - 1ull));
                        if (pass && slot < (unsigned long long)a.cand_cap) a.cand[q * a.cand_cap + slot] = key;
                    }
                }
            }
        }
    }
}

// rows per work item k_list_scan2 is built for (0: it does not apply to this storage)
int list_scan2_chunk_rows(int x_f16, int ld) {
    static const int off = measure_env("RSX_LIST_SCAN_V1", 0);
    return (!off && x_f16 && ld % 64 == 0) ? 64 * LS2_NB : 0;
}

// ---------------------------------------------------------------------------------------
// k_list_scan3 (round 5): the QUERY-STATIONARY form for lists probed by many queries (nprobe >= 64 at the bench's nlist).
// k_list_scan2 keeps a group's queries in LDS (97 KiB for 64) and gives every wave a private ring of rows: a list probed by 65-128
// queries is streamed twice through the CU, and the passes, not HBM latency, are what the 64-query form pays for
// (profiles/r05_ivfflat_wide.md: a prefetch wave and a ring twice as deep both LOSE; time = 3.8 ms of unique bytes + 2.2 ms per
// pass over the 31 GB at nlist 2048 / nprobe 128).  Here a wave keeps ITS 16 queries in registers as MFMA A fragments (2 KT
// x 4 VGPRs = 96 at d = 768), the eight waves of a workgroup share ONE ring of D stages of 128 rows x 128 B (one K step),
// filled by LDS-DMA (wave w fetches rows 16 w .. 16 w + 15 of every stage, the swizzled 1 KiB pieces of k_list_scan2) and read
// by every wave: 128 queries per pass, the LDS holds nothing but the rows in flight (4 stages = 64 KiB).  One s_barrier per K step (stage s
// has landed for everybody / everybody is done with stage s - 1, whose slot takes stage s + D - 1).  Same MFMA, same operand
// order, same K order per (query, row) as k_list_scan2 -> the same bits.
// The scores of a 128-row block leave after its last K step while the DMA stream runs on: vmcnt(2 (D - 2)) stays a SUFFICIENT
// wait with stores in flight (loads retire in order among themselves; outstanding stores can only make it wait for more).
// ---------------------------------------------------------------------------------------
// Ring depth: 4 stages beat 6, 8 and 9 (6.04 / 6.14 / 6.25 / 6.29 ms at nlist 2048 / nprobe 128) and 3 (6.35): the fewer bytes in flight the
// better, down to what covers one barrier.  Requesting the next stage before the barrier instead of after it: +-0.  Stages of 16 WHOLE rows
// (24 KiB contiguous, a wave's 24 MFMAs chained on one accumulator) instead of one K step of 128 rows: 6.27-6.36 against 6.12-6.15 —
// the row stream is not short of DRAM page hits.  2 / 3 / 4 K steps per stage and barrier: 6.16-6.36 against 5.99-6.07 (profiles/r05_ivfflat_wide.md).
#define LS3_D 4
template <bool FILTER, int KT, bool BIAS, int D = LS3_D>
__global__ __launch_bounds__(512) void k_list_scan3(ListScanArgs a) {
    constexpr int NQG = 128, BR = 128, NB = 8;
    constexpr int STAGE = 16384;
    static_assert(KT >= D, "a block's bias slot is rewritten two blocks later: the ring must not reach that far");
    extern __shared__ __attribute__((aligned(16))) unsigned char ls_smem[];
    unsigned char* ring = ls_smem;                                               // [D][8 waves][2 KiB]
    int64_t* segoff = reinterpret_cast<int64_t*>(ls_smem + D * STAGE);          // [NQG] score-buffer offset | (FILTER) row column
    int64_t* sq = segoff + NQG;                                                  // [NQG] query            (FILTER)
    uint64_t* stau = reinterpret_cast<uint64_t*>(sq + NQG);                      // [NQG] threshold key    (FILTER)
    int32_t* sqn = reinterpret_cast<int32_t*>(stau + NQG);                       // [NQG] query numbers, -1 = padding
    float* sbias = reinterpret_cast<float*>(sqn + NQG);                          // [2][BR] the rows' bias of the even / odd blocks (BIAS)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int g = blockIdx.x;
    int chunk = blockIdx.y;
    int l, gi;
    if (a.item_off) {
        if (!pq_decode_item(a.item_off, a.group_off, *a.total_items, a.nlist, l, gi, chunk)) return;
    } else {
        if (g >= *a.total_groups) return;
        int lo = 0, hi = a.nlist;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (a.group_off[mid] <= g) lo = mid; else hi = mid; }
        l = lo;
        gi = g - a.group_off[l];
    }
    const int cnt = a.pair_off[l + 1] - a.pair_off[l];
    int np = cnt - NQG * gi; if (np > NQG) np = NQG;
    const int pair0 = a.pair_off[l] + NQG * gi;
    const int64_t base = a.list_base[l], len = a.list_len[l];
    const int64_t len_pad = (len + 15) & ~15ll;
    const int64_t c0 = (int64_t)chunk * (BR * NB);
    if (c0 >= len_pad) return;
    int64_t c1 = c0 + BR * NB; if (c1 > len_pad) c1 = len_pad;

    if (tid < NQG) {
        int64_t q = -1, off = 0; uint64_t t = ~0ull;      // padding slot: nothing passes
        if (tid < np) {
            const int pidx = a.pairs_sorted[pair0 + tid];
            q = pidx / a.nprobe; const int j = pidx % a.nprobe;
            off = q * a.tstride + ((!FILTER && a.pre_stride) ? (int64_t)j * a.pre_stride : a.seg_start[q * (a.nprobe + 1) + j]);
            if (FILTER) { t = a.tau_key[q * a.tau_stride]; off -= q * a.tstride; }      // the column inside the query's row = the candidate's index
        }
        sqn[tid] = (int32_t)q; segoff[tid] = off;
        if (FILTER) { sq[tid] = q < 0 ? 0 : q; stau[tid] = t; }
    }
    __syncthreads();

    const int lr = lane & 15, kg = lane >> 4;
    const bool active = 16 * w < np;           // waves without queries still fetch their rows of every stage
    half8 qa[2 * KT];
    {
        const int q = sqn[16 * w + lr];
        const __half* qp = a.Q16 + (int64_t)(q < 0 ? 0 : q) * a.ld + 8 * kg;
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
            qa[2 * kt] = q >= 0 ? *reinterpret_cast<const half8*>(qp + kt * 64) : z;
            qa[2 * kt + 1] = q >= 0 ? *reinterpret_cast<const half8*>(qp + kt * 64 + 32) : z;
        }
#pragma unroll
        for (int i = 0; i < 2 * KT; i++) asm volatile("" : "+v"(qa[i]));      // landed before the counted waits below start counting
    }

    const int64_t nrows = c1 - c0;             // a multiple of 16
    const int nblk = (int)((nrows + BR - 1) / BR);
    const int T = nblk * KT;
    const char* xc = reinterpret_cast<const char*>(reinterpret_cast<const __half*>(a.X) + (base + c0) * a.ld);
    const int64_t row_bytes = (int64_t)a.ld * 2;
    uint32_t off0, off1;
    {
        const int r8 = lane >> 3, p = lane & 7;
        const int R0 = r8, R1 = 8 + r8;
        off0 = (uint32_t)R0 * (uint32_t)(a.ld * 2) + ((p ^ ((R0 >> 1) & 7)) << 4);
        off1 = (uint32_t)R1 * (uint32_t)(a.ld * 2) + ((p ^ ((R1 >> 1) & 7)) << 4);
    }
    // my 16 rows of block b; past the end of a short last block: its last 16 rows again (nobody reads that copy)
    auto row0 = [&](int b) { int64_t r = (int64_t)BR * b + 16 * w; return r > nrows - 16 ? nrows - 16 : r; };
    int i_kt = 0, i_blk = 0, i_slot = 0, i_left = T;
    int64_t i_boff = row0(0) * row_bytes;
    // BIAS (squared distances: score = q.x - |x|^2 / 2): the FIRST stage of a block carries a third piece, ahead of its two row pieces — the
    // 16 bias values of the wave's rows, into the block's slot of sbias.  Loads retire in order, so the piece is in LDS when the stage's
    // rows are, for everybody after that step's barrier, long before the block's epilogue reads it.  The counted waits see it as one
    // more newer piece while a NEXT block's first stage is among the D - 2 stages in flight behind the awaited one: the block's last
    // D - 2 steps, unless it is the chunk's last block (a piece per STAGE keeps the literal single and was measured first: 6.67-6.72 ms
    // against 6.53-6.56 at nlist 2048 / nprobe 128, squared distances; the 64-query form: 7.1-7.5).
    const float* bias_l = BIAS ? a.bias + base : nullptr;
    auto issue = [&]() {
        const char* gsrc = xc + i_boff + (int64_t)i_kt * 128;
        unsigned char* dst = ring + i_slot * STAGE + w * 2048;
        if (BIAS && i_kt == 0) {
            if (lane < 16) {
                int64_t r = c0 + row0(i_blk) + lane; r = r > len - 1 ? len - 1 : r;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bias_l + r),
                                                 (__attribute__((address_space(3))) void*)(sbias + (i_blk & 1) * BR + 16 * w), 4, 0, 0);
            }
        }
        fg2_dma16<RSX_NT_LIST>(gsrc + off0, dst);
        fg2_dma16<RSX_NT_LIST>(gsrc + off1, dst + 1024);
        const bool adv = i_left > 1;           // past the last step: the last piece again, into the slot of a finished stage
        i_left -= adv ? 1 : 0;
        i_kt += adv ? 1 : 0;
        if (i_kt == KT) { i_kt = 0; i_blk++; i_boff = row0(i_blk) * row_bytes; }
        i_slot = (i_slot + 1 == D) ? 0 : i_slot + 1;
    };
#pragma unroll
    for (int d = 0; d < D - 1; d++) issue();

    const int swz = (lr >> 1) & 7;
    const int boff0 = lr * 128 + (((0 + kg) ^ swz) << 4), boff1 = lr * 128 + (((4 + kg) ^ swz) << 4);
    int slot = 0;
    for (int b = 0; b < nblk; b++) {
        floatx4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        const int64_t rb = (int64_t)BR * b;
        const int jn = nrows - rb >= BR ? 8 : (int)((nrows - rb) >> 4);
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
            // my pieces of this stage are in LDS ...
            if (BIAS && kt >= KT - (D - 2) && b + 1 < nblk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (D - 2) + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (D - 2)) : "memory");
            __builtin_amdgcn_s_barrier();                           // ... and so are everybody's; everybody has left the previous stage
            issue();                                                // -> into the previous stage's slot
            if (active) {
                const unsigned char* bs = ring + slot * STAGE;
                // all eight 16-row pieces, also of a short last block (what the slot holds beyond its rows is never stored); the sixteen
                // fragment reads go out first, the MFMAs follow them in with counted waits
                half8 xs[16];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    xs[2 * j] = *reinterpret_cast<const half8*>(bs + j * 2048 + boff0);
                    xs[2 * j + 1] = *reinterpret_cast<const half8*>(bs + j * 2048 + boff1);
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[2 * kt], xs[2 * j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[2 * kt + 1], xs[2 * j + 1], acc[j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);      // eight reads ahead, then two MFMAs per two reads
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            slot = (slot + 1 == D) ? 0 : slot + 1;
        }
        if (!active) continue;
        // C/D layout of the 16x16 MFMA: reg r of lane l holds (row i = 4 (l >> 4) + r = query, col j = l & 15 = db row)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j >= jn) break;
            const int64_t rloc = c0 + rb + 16 * j + lr;
            const float bv = BIAS ? sbias[(b & 1) * BR + 16 * j + lr] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int qi = 16 * w + kg * 4 + r;
                if (!FILTER) {
                    if (qi < np) a.temp[segoff[qi] + rloc] = (rloc < len) ? acc[j][r] + bv : -__builtin_inff();
                } else {
                    const uint64_t key = (rloc < len && qi < np) ? make_key(acc[j][r] + bv, (uint32_t)(segoff[qi] + rloc)) : 0ull;
                    const bool pass = key > stau[qi];
                    const uint64_t mask = __ballot(pass);
                    const uint64_t mine = (mask >> (16 * kg)) & 0xffffull;
                    if (mine) {   // one atomic per query per 16 rows
                        const int64_t q = sq[qi];
                        unsigned long long slot0 = 0;
                        const int leader = (__ffsll((unsigned long long)mine) - 1) + 16 * kg;
                        if (lane == leader) slot0 = atomicAdd(&a.cand_cnt[(int64_t)q * CCS], (unsigned long long)__popcll(mine));
                        slot0 = __shfl(slot0, leader);
                        const unsigned long long sl = slot0 + __popcll(mine & ((1ull << lr) - 1ull));
                        if (pass && sl < (unsigned long long)a.cand_cap) a.cand[q * a.cand_cap + sl] = key;
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// the query-stationary form applies to fp16 rows of d = 384 / 512 / 768 / 1024 (KT = 6 / 8 / 12 / 16: the fragments are register arrays); 0 otherwise
int list_scan3_applies(int x_f16, int ld, int has_bias) {
    static const int off = measure_env("RSX_LIST_SCAN3_OFF", 0);
    (void)has_bias;                              // (round 5, second half: the bias rides in the stages)
    return !off && list_scan2_chunk_rows(x_f16, ld) > 0 && (ld == 384 || ld == 512 || ld == 768 || ld == 1024);
}

// 16-query tiles per group the LDS holds beside the four DMA rings (queries 16 qt x (ld + 8) halfs)
int list_scan2_max_qtiles(int ld) {
    for (int qt = 4; qt > 1; qt >>= 1)
        if ((size_t)16 * qt * (ld + 8) * 2 + 28 * 16 * qt + (size_t)(4 * LS2_D) * 2048 <= 158 * 1024) return qt;      // (8 x 3 stages = the same 48 KiB)
    return 1;
}

void launch_list_scan(const ListScanArgs& a, hipStream_t st) {
    if (a.max_groups <= 0 || a.max_chunks <= 0) return;
    dim3 grid((unsigned)a.max_groups, (unsigned)a.max_chunks);
    const int base_rows = list_scan2_chunk_rows(a.x_f16, a.ld);
    if (base_rows > 0 && (a.chunk_rows == base_rows || ((a.qtiles == 4 || a.qtiles == 8) && a.chunk_rows == 2 * base_rows))) {
        if (a.item_off && !a.flat_mode && a.max_items > 0) grid = dim3((unsigned)((a.max_items + 7) & ~7), 1);   // XCD-aware item order
        if (a.qtiles == 8) {            // 128 queries per group: the query-stationary form (the caller asked list_scan3_applies)
            const size_t shm3 = (size_t)LS3_D * 16384 + 28 * 128 + 2 * 128 * 4;
#ifdef RSX_MEASURE
            static const int d_env = measure_env("RSX_LS3_D", 0);
#define LS3_VARIANT(DD) if (d_env == DD && a.ld == 768 && !a.bias) { \
                (void)hipFuncSetAttribute((const void*)k_list_scan3<false, 12, false, DD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                (void)hipFuncSetAttribute((const void*)k_list_scan3<true, 12, false, DD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                const size_t shm = (size_t)DD * 16384 + 28 * 128 + 2 * 128 * 4; \
                if (a.tau_key) hipLaunchKernelGGL((k_list_scan3<true, 12, false, DD>), grid, dim3(512), shm, st, a); \
                else hipLaunchKernelGGL((k_list_scan3<false, 12, false, DD>), grid, dim3(512), shm, st, a); \
                return; }
            LS3_VARIANT(3) LS3_VARIANT(6) LS3_VARIANT(8)
#undef LS3_VARIANT
#endif
#define LS3_LAUNCH(KT_, B_) { \
                static DevOnce once3; \
                once3.once([&] { \
                    (void)hipFuncSetAttribute((const void*)k_list_scan3<false, KT_, B_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                    (void)hipFuncSetAttribute((const void*)k_list_scan3<true, KT_, B_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                }); \
                if (a.tau_key) hipLaunchKernelGGL((k_list_scan3<true, KT_, B_>), grid, dim3(512), shm3, st, a); \
                else hipLaunchKernelGGL((k_list_scan3<false, KT_, B_>), grid, dim3(512), shm3, st, a); \
                return; }
#define LS3_BY_BIAS(KT_) { if (a.bias) LS3_LAUNCH(KT_, true) else LS3_LAUNCH(KT_, false) }
            switch (a.ld) {
                case 384: LS3_BY_BIAS(6)
                case 512: LS3_BY_BIAS(8)
                case 768: LS3_BY_BIAS(12)
                case 1024: LS3_BY_BIAS(16)
                default: fprintf(stderr, "rsx: k_list_scan3 asked for d = %d (internal error)\n", a.ld); abort();
            }
#undef LS3_BY_BIAS
#undef LS3_LAUNCH
        }
        // the grouping was made for 16 x qtiles queries per group: qtiles is binding (a smaller kernel would misread the groups)
        const int qt = (a.qtiles == 2 || a.qtiles == 4) && !a.flat_mode ? a.qtiles : 1;
        const bool wide = qt == 4 && a.chunk_rows == 2 * base_rows;        // 8 waves x 3 stages, 1024 rows per work item
        const int nw = wide ? 8 : 4, dd = wide ? 3 : LS2_D;
        size_t shm2 = (size_t)16 * qt * (a.ld + 8) * 2 + 28 * 16 * qt + (size_t)nw * dd * 2048;
        static DevOnce once;
        once.once([&] {
            (void)hipFuncSetAttribute((const void*)k_list_scan2<false, 1, 4, LS2_D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_list_scan2<true, 1, 4, LS2_D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_list_scan2<false, 2, 4, LS2_D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_list_scan2<true, 2, 4, LS2_D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_list_scan2<false, 4, 4, LS2_D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_list_scan2<true, 4, 4, LS2_D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_list_scan2<false, 4, 8, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_list_scan2<true, 4, 8, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        if (wide) {
            if (a.tau_key) hipLaunchKernelGGL((k_list_scan2<true, 4, 8, 3>), grid, dim3(512), shm2, st, a);
            else hipLaunchKernelGGL((k_list_scan2<false, 4, 8, 3>), grid, dim3(512), shm2, st, a);
        } else if (qt == 4) {
            if (a.tau_key) hipLaunchKernelGGL((k_list_scan2<true, 4, 4, LS2_D>), grid, dim3(256), shm2, st, a);
            else hipLaunchKernelGGL((k_list_scan2<false, 4, 4, LS2_D>), grid, dim3(256), shm2, st, a);
        } else if (qt == 2) {
            if (a.tau_key) hipLaunchKernelGGL((k_list_scan2<true, 2, 4, LS2_D>), grid, dim3(256), shm2, st, a);
            else hipLaunchKernelGGL((k_list_scan2<false, 2, 4, LS2_D>), grid, dim3(256), shm2, st, a);
        } else {
            if (a.tau_key) hipLaunchKernelGGL((k_list_scan2<true, 1, 4, LS2_D>), grid, dim3(256), shm2, st, a);
            else hipLaunchKernelGGL((k_list_scan2<false, 1, 4, LS2_D>), grid, dim3(256), shm2, st, a);
        }
        return;
    }
    size_t shm = (size_t)16 * (a.ld + 8) * 2 + 16 * sizeof(int64_t);
    if (a.x_f16) {
        if (shm > 48 * 1024) hipFuncSetAttribute((const void*)k_list_scan<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(k_list_scan<true>, grid, dim3(256), shm, st, a);
    } else {
        if (shm > 48 * 1024) hipFuncSetAttribute((const void*)k_list_scan<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(k_list_scan<false>, grid, dim3(256), shm, st, a);
    }
}

}  // namespace rsx
