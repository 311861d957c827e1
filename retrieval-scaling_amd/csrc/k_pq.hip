// k_pq.hip — product-quantiser kernels of IndexIVFPQ (reference src/indicies/ivf_pq.py:147-153,
// search :230, add :185).  METRIC_INNER_PRODUCT, by_residual: the look-up table
// T[q][m][c] = <q_m, codebook[m][c]> does not depend on the list, and
// score(q, v) = <q, centroid(list(v))> + sum_m T[q][m][code_v[m]].
//
//  k_pq_lut    : T for a batch of queries (fp32 fmaf chains, bit-equal to the oracle).
//  k_pq_scan   : one workgroup = (query, probed list, chunk of slabs).  The query's table is staged
//                in LDS (M KiB); codes stream from HBM in the slab layout (lane v of a wave reads 16
//                contiguous bytes, the wave 1 KiB per instruction); each lane gathers its vector's
//                M table entries from LDS and accumulates them in m order, exactly the sequential
//                fp32 sum of FAISS's generic IVFPQ scanner; the score goes to the query's row of the
//                score buffer (coalesced fp32 stores).  Integer/byte work bounded by HBM and the LDS
//                gather rate — deliberately NOT reshaped into a GEMM.
//  k_pq_scan2  : exact list-major scan (two queries per ds_read_b64 of an interleaved float2 table): the
//                certified fast path's fallback and the A/B reference.
//  k_pq_lut8 / k_pq_lut8f / k_pq_lut_tiled + k_pq_qparam : the fast path's 8-bit affine-quantised tables with a
//                rigorous per-query error bound (from an fp32 table in HBM / built in LDS per query / built by
//                codebook-slice tiles shared by 32 queries — all three give the same bits).
//  k_pq_scan8  : THE hot kernel: list-major fast scan, four queries' 8-bit tables interleaved as one dword per
//                (code, m) in 97 KiB of LDS, one ds_read_b32 per (vector, m) serves four queries, exact integer
//                sums in 16-bit fields, candidates beating the query's threshold key appended from the kernel.
//                (k_pq_prepass, the one-launch threshold pre-pass, lives in k_select.hip next to the radix select.)
//  k_pq_encode : ProductQuantizer::compute_code on residuals: nearest codeword per subspace by
//                squared L2 (fmaf chain, first minimum), written straight into the slab layout.
#include <cstdlib>

#include "rsx_internal.h"

namespace rsx {

// number of fp32 roundings separating the certificate's two scores (see k_pq_qparam), as a relative slack on B
__device__ __forceinline__ float pq_round_slack(int M) { return (float)(3 * M + 12) * 5.9604645e-8f * 1.05f; }


// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pq_lut(const float* Q32, int ldq, int d, int M, int dsub,
                                                const float* codebooks, float* lut, int Mpad) {
    const int64_t q = blockIdx.x;
    const int m = blockIdx.y;
    const int c = threadIdx.x;
    float s = 0.0f;
    if (m < M) {
        const float* qs = Q32 + q * ldq + m * dsub;
        const float* cw = codebooks + ((int64_t)m * 256 + c) * dsub;
        for (int t = 0; t < dsub; t++) s = __fmaf_rn(qs[t], cw[t], s);
    }
    lut[(q * Mpad + m) * 256 + c] = s;
}
void launch_pq_lut(const float* Q32, int ldq, int64_t nq, int d, int M, int Mpad, const float* codebooks,
                   float* lut, hipStream_t st) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_pq_lut, dim3((unsigned)nq, Mpad), dim3(256), 0, st, Q32, ldq, d, M, d / M, codebooks, lut, Mpad);
}

// ---------------------------------------------------------------------------------------
// NCH = Mpad/16 for the 16-byte-granule layout (CB = 16); NCH = 0: generic 4-byte granules.
template <int NCH>
__global__ __launch_bounds__(1024) void k_pq_scan(PQScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float pq_lut_s[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nw = blockDim.x >> 6;
    const int64_t pair = blockIdx.x;
    const int chunk = blockIdx.y;
    const int32_t l = a.probe_list[pair];
    if (l < 0) return;
    const int64_t len = a.list_len[l];
    const int64_t nslab = (len + 63) >> 6;
    const int64_t s0 = (int64_t)chunk * a.slabs_per_chunk;
    if (s0 >= nslab) return;
    int64_t s1 = s0 + a.slabs_per_chunk; if (s1 > nslab) s1 = nslab;
    const int64_t q = pair / a.nprobe;
    const int j = (int)(pair - q * a.nprobe);
    const float dis0 = a.probe_dis0[pair];
    float* out = a.temp + q * a.tstride + a.seg_start[q * (a.nprobe + 1) + j];

    // stage the query's table: Mpad*256 floats
    {
        const float4* src = reinterpret_cast<const float4*>(a.lut + q * a.Mpad * 256);
        float4* dst = reinterpret_cast<float4*>(pq_lut_s);
        const int n4 = a.Mpad * 64;
        for (int i = tid; i < n4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    const int64_t slab_base = a.list_base[l] >> 6;
    const int64_t slab_bytes = (int64_t)64 * a.Mpad;
    for (int64_t s = s0 + w; s < s1; s += nw) {
        const uint8_t* sp = a.codes + (slab_base + s) * slab_bytes;
        float sum = 0.0f;
        if (NCH > 0) {
            uint4 c[NCH > 0 ? NCH : 1];
#pragma unroll
            for (int g = 0; g < NCH; g++) c[g] = *reinterpret_cast<const uint4*>(sp + g * 1024 + lane * 16);
#pragma unroll
            for (int g = 0; g < NCH; g++) {
                const uint32_t wds[4] = {c[g].x, c[g].y, c[g].z, c[g].w};
#pragma unroll
                for (int b = 0; b < 16; b++) {
                    uint32_t code = (wds[b >> 2] >> (8 * (b & 3))) & 0xffu;
                    sum += pq_lut_s[(g * 16 + b) * 256 + code];
                }
            }
        } else {
            const int ng = a.Mpad >> 2;
            for (int g = 0; g < ng; g++) {
                uint32_t wd = *reinterpret_cast<const uint32_t*>(sp + g * 256 + lane * 4);
#pragma unroll
                for (int b = 0; b < 4; b++) sum += pq_lut_s[(g * 4 + b) * 256 + ((wd >> (8 * b)) & 0xffu)];
            }
        }
        const int64_t pos = s * 64 + lane;
        out[pos] = (pos < len) ? dis0 + sum : -__builtin_inff();
    }
}

template <int NCH>
static int launch_pq_scan_t(const PQScanArgs& a, dim3 grid, size_t shm, hipStream_t st) {
    if (hipFuncSetAttribute((const void*)k_pq_scan<NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
        return -1;
    hipLaunchKernelGGL(k_pq_scan<NCH>, grid, dim3(1024), shm, st, a);
    return 0;
}

int launch_pq_scan(const PQScanArgs& a, hipStream_t st) {
    int64_t pairs = a.nq * a.nprobe;
    if (pairs <= 0 || a.max_chunks <= 0) return 0;
    dim3 grid((unsigned)pairs, (unsigned)a.max_chunks);
    size_t shm = (size_t)a.Mpad * 1024;
    if (shm > 160 * 1024) return -1;
    if (a.CB == 16) {
        switch (a.Mpad / 16) {
            case 1: return launch_pq_scan_t<1>(a, grid, shm, st);
            case 2: return launch_pq_scan_t<2>(a, grid, shm, st);
            case 3: return launch_pq_scan_t<3>(a, grid, shm, st);
            case 4: return launch_pq_scan_t<4>(a, grid, shm, st);
            case 6: return launch_pq_scan_t<6>(a, grid, shm, st);
            case 8: return launch_pq_scan_t<8>(a, grid, shm, st);
            default: return -1;
        }
    }
    return launch_pq_scan_t<0>(a, grid, shm, st);
}

// ---------------------------------------------------------------------------------------
// k_pq_scan2: list-major scan, TWO queries per LDS read.
//
// The v1 kernel above is bound by the LDS gather rate: one ds_read_b32 per (query, vector, m) and
// ~3.5-way bank conflicts on random codes (measured 8.5 look-ups/clk/CU).  Here the (query, probe)
// pairs of a batch are grouped by inverted list (k_select.hip: launch_group_pairs, groups of 2); a
// workgroup takes (list, pair of queries, tile of slabs), stages BOTH queries' tables interleaved
// as float2 [m][code] and every lane issues ONE ds_read_b64 per (vector, m) that serves both
// queries: half the LDS instructions at the same conflict rate, and each list's codes are pulled
// from HBM/L2 once per query pair instead of once per query.  160 KiB of LDS holds 48
// sub-quantisers of a float2 table (96 KiB), so M = 96 runs in two passes with the partial sums
// kept in registers; the additions still happen in m = 0..M-1 order, so scores stay bit-identical
// to v1 / the oracle.
// NCH = Mpad/16 granules per vector, VPL = slabs per wave per tile (register accumulators).
struct PQScan2Args {
    PQScanArgs b;
    const int32_t* pairs_sorted; const int32_t* pair_off; const int32_t* group_off; const int32_t* total_groups;
    const int32_t* item_off; const int32_t* total_items;   // work items = (list, tile, group), list-major
    int nlist; int max_items;
};

template <int NCH, int VPL>
__global__ __launch_bounds__(1024) void k_pq_scan2(PQScan2Args A) {
    extern __shared__ __attribute__((aligned(16))) float2 pq_lut2_s[];
    constexpr int GPC = NCH < 3 ? NCH : 3;  // granules (16 sub-quantisers each) per pass
    constexpr int NP = (NCH + GPC - 1) / GPC;
    const PQScanArgs& a = A.b;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int l, gi, tile;
    if (!pq_decode_item(A.item_off, A.group_off, *A.total_items, A.nlist, l, gi, tile)) return;
    const int cnt = A.pair_off[l + 1] - A.pair_off[l];
    const int np = (cnt - 2 * gi) > 1 ? 2 : 1;
    const int pair0 = A.pair_off[l] + 2 * gi;
    const int64_t len = a.list_len[l];
    const int64_t nslab = (len + 63) >> 6;
    const int64_t s0 = (int64_t)tile * (16 * VPL);
    if (s0 >= nslab) return;
    const int p0 = A.pairs_sorted[pair0];
    const int p1 = (np > 1) ? A.pairs_sorted[pair0 + 1] : p0;
    const int64_t q0 = p0 / a.nprobe, q1 = p1 / a.nprobe;
    const float* lut0 = a.lut + q0 * a.Mpad * 256;
    const float* lut1 = a.lut + q1 * a.Mpad * 256;

    float2 acc[VPL];
#pragma unroll
    for (int u = 0; u < VPL; u++) acc[u] = make_float2(0.0f, 0.0f);

    const int64_t slab_base = a.list_base[l] >> 6;
    const int64_t slab_bytes = (int64_t)64 * a.Mpad;
#pragma unroll
    for (int pass = 0; pass < NP; pass++) {
        constexpr int dummy = 0; (void)dummy;
        const int g0 = pass * GPC;
        const int ng = (NCH - g0) < GPC ? (NCH - g0) : GPC;
        if (pass > 0) __syncthreads();
        uint4 cur[GPC], nxt[GPC];
        if (s0 + w < nslab) {   // first slab of the pass requested before the table staging
            const uint8_t* sp = a.codes + (slab_base + s0 + w) * slab_bytes;
#pragma unroll
            for (int gg = 0; gg < GPC; gg++)
                if (gg < ng) cur[gg] = *reinterpret_cast<const uint4*>(sp + (g0 + gg) * 1024 + lane * 16);
        }
        for (int i = tid; i < ng * 16 * 256; i += 1024)
            pq_lut2_s[i] = make_float2(lut0[g0 * 16 * 256 + i], lut1[g0 * 16 * 256 + i]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < VPL; u++) {
            const int64_t s = s0 + w + 16 * u;
            const int64_t sn = s + 16;
            if (u + 1 < VPL && sn < nslab) {   // next slab's codes in flight during this slab's gathers
                const uint8_t* sp = a.codes + (slab_base + sn) * slab_bytes;
#pragma unroll
                for (int gg = 0; gg < GPC; gg++)
                    if (gg < ng) nxt[gg] = *reinterpret_cast<const uint4*>(sp + (g0 + gg) * 1024 + lane * 16);
            }
            if (s < nslab) {
                float2 sum = acc[u];
#pragma unroll
                for (int gg = 0; gg < GPC; gg++) {
                    if (gg < ng) {
                        const uint32_t wds[4] = {cur[gg].x, cur[gg].y, cur[gg].z, cur[gg].w};
#pragma unroll
                        for (int b = 0; b < 16; b++) {
                            uint32_t code = (wds[b >> 2] >> (8 * (b & 3))) & 0xffu;
                            float2 t = pq_lut2_s[(gg * 16 + b) * 256 + code];
                            sum.x += t.x; sum.y += t.y;
                        }
                    }
                }
                acc[u] = sum;
            }
#pragma unroll
            for (int gg = 0; gg < GPC; gg++) cur[gg] = nxt[gg];
        }
    }
    const float d0 = a.probe_dis0[p0], d1 = a.probe_dis0[p1];
    float* out0 = a.temp + q0 * a.tstride + a.seg_start[q0 * (a.nprobe + 1) + (p0 - (int)q0 * a.nprobe)];
    float* out1 = a.temp + q1 * a.tstride + a.seg_start[q1 * (a.nprobe + 1) + (p1 - (int)q1 * a.nprobe)];
#pragma unroll
    for (int u = 0; u < VPL; u++) {
        const int64_t s = s0 + w + 16 * u;
        if (s < nslab) {
            const int64_t pos = s * 64 + lane;
            out0[pos] = (pos < len) ? d0 + acc[u].x : -__builtin_inff();
            if (np > 1) out1[pos] = (pos < len) ? d1 + acc[u].y : -__builtin_inff();
        }
    }
}

template <int NCH, int VPL>
static int launch_pq_scan2_t(const PQScan2Args& A, hipStream_t st) {
    constexpr int GPC = NCH < 3 ? NCH : 3;
    size_t shm = (size_t)GPC * 16 * 256 * sizeof(float2);
    if (hipFuncSetAttribute((const void*)k_pq_scan2<NCH, VPL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
        return -1;
    dim3 grid((unsigned)((A.max_items + 7) & ~7));
    hipLaunchKernelGGL((k_pq_scan2<NCH, VPL>), grid, dim3(1024), shm, st, A);
    return 0;
}
template <int NCH>
static int launch_pq_scan2_v(const PQScan2Args& A, int vpl, hipStream_t st) {
    switch (vpl) {
        case 8: return launch_pq_scan2_t<NCH, 8>(A, st);
        case 4: return launch_pq_scan2_t<NCH, 4>(A, st);
        case 2: return launch_pq_scan2_t<NCH, 2>(A, st);
        default: return launch_pq_scan2_t<NCH, 1>(A, st);
    }
}
// returns 0 on launch, -1 if this (M, layout) has no v2 kernel (caller falls back to k_pq_scan)
int launch_pq_scan2(const PQScanArgs& a, const int32_t* pairs_sorted, const int32_t* pair_off, const int32_t* group_off,
                    const int32_t* total_groups, const int32_t* item_off, const int32_t* total_items, int nlist,
                    int64_t max_items, int vpl, hipStream_t st) {
    if (a.CB != 16 || max_items <= 0 || max_items > 0x7fffff00) return -1;
    PQScan2Args A;
    A.b = a; A.pairs_sorted = pairs_sorted; A.pair_off = pair_off; A.group_off = group_off; A.total_groups = total_groups;
    A.item_off = item_off; A.total_items = total_items;
    A.nlist = nlist; A.max_items = (int)max_items;
    switch (a.Mpad / 16) {
        case 1: return launch_pq_scan2_v<1>(A, vpl, st);
        case 2: return launch_pq_scan2_v<2>(A, vpl, st);
        case 3: return launch_pq_scan2_v<3>(A, vpl, st);
        case 4: return launch_pq_scan2_v<4>(A, vpl, st);
        case 6: return launch_pq_scan2_v<6>(A, vpl, st);
        case 8: return launch_pq_scan2_v<8>(A, vpl, st);
        default: return -1;
    }
}

// ---------------------------------------------------------------------------------------
// Fast scan with an 8-bit table + certified exact re-rank.
//
// k_pq_scan2 is bound by the LDS gather INSTRUCTION rate (measured ~10 clk per random ds_read_b64,
// ~7 per ds_read_b32), so the lever is queries served per gathered dword.  k_pq_lut8 quantises each
// query's table affinely to 8 bits (one scale per query, one offset per sub-quantiser):
//     T[m][c] ~= mn[m] + scale * u8[m][c],   |error| <= e_m,   e_quant = sum_m e_m
// k_pq_scan8 interleaves FOUR queries' u8 tables as one dword per (m, code) — M KiB of LDS, a single
// pass for M = 96 — so one ds_read_b32 per (vector, m) serves four queries; the four sums are exact
// integers (two 16-bit fields per accumulator register).  The approximate score
//     a(v) = dis0 + bias + scale * A(v)          (bias = sum_m mn[m])
// is within eps_q = e_quant + fp32 slack of the canonical fp32 score s(v).  k_select keeps the top K'
// by a(.), k_finalize re-scores them EXACTLY (sequential fp32 table sum, = oracle) and certifies:
// if a(K'-th candidate) + eps_q < s(k-th best candidate) no excluded vector can reach the top k, so
// the result equals the exact search; otherwise the query is flagged and re-run with k_pq_scan2.

// Unfused form (tables too large for LDS): quantises an fp32 table k_pq_lut already wrote to HBM.
__global__ __launch_bounds__(256) void k_pq_lut8(const float* lut32, int M, int Mpad, const float* probe_dis0, int nprobe,
                                                 uint8_t* lut8, PQQParam* qp, int transposed) {
    __shared__ float s_mn[256], s_red[8];
    const int64_t q = blockIdx.x;
    const int c = threadIdx.x, lane = c & 63, w = c >> 6;
    const float* T = lut32 + q * Mpad * 256;
    float absmax_sum = 0.0f, maxrange = 0.0f;
    for (int m = 0; m < Mpad; m++) {
        float v = T[m * 256 + c];
        float mn = v, mx = v;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
        if (lane == 0) { s_red[w] = mn; s_red[4 + w] = mx; }
        __syncthreads();
        mn = fminf(fminf(s_red[0], s_red[1]), fminf(s_red[2], s_red[3]));
        mx = fmaxf(fmaxf(s_red[4], s_red[5]), fmaxf(s_red[6], s_red[7]));
        __syncthreads();
        if (c == 0) s_mn[m] = mn;
        absmax_sum += fmaxf(fabsf(mn), fabsf(mx));
        maxrange = fmaxf(maxrange, mx - mn);
    }
    __syncthreads();
    const float scale = maxrange > 0.0f ? maxrange / 255.0f : 1.0f;
    const float inv = 1.0f / scale;
    float e_quant = 0.0f, bias = 0.0f;
    for (int m = 0; m < Mpad; m++) {
        float v = T[m * 256 + c];
        float mn = s_mn[m];
        float u = rintf((v - mn) * inv);
        u = fminf(fmaxf(u, 0.0f), 255.0f);
        lut8[pq_lut8_index(q, c, m, Mpad, transposed)] = (uint8_t)u;   // 1: [q][code][m], the rotated-layout scans; 2: sliced
        float err = fabsf(v - (mn + scale * u));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) err = fmaxf(err, __shfl_xor(err, off));
        if (lane == 0) s_red[w] = err;
        __syncthreads();
        err = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        __syncthreads();
        e_quant += err;
        bias += mn;
    }
    if (c == 0) {
        float d0 = 0.0f;
        for (int j = 0; j < nprobe; j++) { float d = probe_dis0[q * nprobe + j]; if (d > -__builtin_inff()) d0 = fmaxf(d0, fabsf(d)); }
        // fp32 slack: every fp32 operation between the stored quantities and the two scores compared by the certificate rounds
        // by at most 2^-24 of a magnitude bounded by B — M adds in the exact score's table sum (+1 for dis0), M adds in bias,
        // M (mul, add, sub) in the per-entry error terms, and the fma + add of the approximate score: fewer than 3 M + 12
        float B = absmax_sum + d0 + fabsf(bias) + scale * 255.0f * (float)M + 1.0f;
        PQQParam r; r.scale = scale; r.bias = bias; r.eps = e_quant * 1.0001f + pq_round_slack(M) * B; r.pad = 255.0f * (float)Mpad;   // no tighter bound computed on this path
        qp[q] = r;
    }
}

// Fused form: one workgroup builds the query's fp32 table T[m][c] = <q_m, cb[m][c]> (k_pq_lut's fmaf chain,
// hence the same bits) straight into LDS, derives the affine 8-bit quantisation from it and writes only
// the u8 table + parameters.  The fp32 table (98 KB/query) never touches HBM; k_finalize recomputes the
// few entries the exact re-score needs.  LDS: Mpad*1 KiB table + the query + 3*Mpad floats.
template <int DSUB>   // 8: two float4 loads per codeword; 0: generic dsub
__global__ __launch_bounds__(256) void k_pq_lut8f(const float* Q32, int ldq, const float* codebooks, int dsub, int M,
                                                  int Mpad, const float* probe_dis0, int nprobe, uint8_t* lut8,
                                                  PQQParam* qp, int transposed, float* lut32_out) {
    extern __shared__ float sm_lut8f[];
    float* T = sm_lut8f;                 // [Mpad][256]
    float* s_q = T + Mpad * 256;         // [M*dsub]
    float* s_mn = s_q + M * dsub;        // [Mpad]
    float* s_mx = s_mn + Mpad;           // [Mpad]
    float* s_err = s_mx + Mpad;          // [Mpad]
    const int64_t q = blockIdx.x;
    const int c = threadIdx.x, lane = c & 63, w = c >> 6;
    for (int i = c; i < M * dsub; i += 256) s_q[i] = Q32[q * ldq + i];
    __syncthreads();
    // A: thread c owns codeword c of every sub-quantiser; the loads of different m are independent
#pragma unroll 4
    for (int m = 0; m < M; m++) {
        float s = 0.0f;
        if (DSUB == 8) {
            const float4* cw = (const float4*)(codebooks + ((int64_t)m * 256 + c) * 8);
            float4 x = cw[0], y = cw[1];
            const float* qs = s_q + m * 8;
            s = __fmaf_rn(qs[0], x.x, s); s = __fmaf_rn(qs[1], x.y, s); s = __fmaf_rn(qs[2], x.z, s); s = __fmaf_rn(qs[3], x.w, s);
            s = __fmaf_rn(qs[4], y.x, s); s = __fmaf_rn(qs[5], y.y, s); s = __fmaf_rn(qs[6], y.z, s); s = __fmaf_rn(qs[7], y.w, s);
        } else {
            const float* cw = codebooks + ((int64_t)m * 256 + c) * dsub;
            const float* qs = s_q + m * dsub;
            for (int t = 0; t < dsub; t++) s = __fmaf_rn(qs[t], cw[t], s);
        }
        T[m * 256 + c] = s;
        if (lut32_out) lut32_out[(q * Mpad + m) * 256 + c] = s;      // the fp32 table for k_pq_final_tab (round 4): M KiB per query
    }
    for (int m = M; m < Mpad; m++) T[m * 256 + c] = 0.0f;
    __syncthreads();
    // B: wave w reduces min / max of sub-quantisers w, w+4, ... (4 entries per lane, shuffles only)
    for (int m = w; m < Mpad; m += 4) {
        float4 v = *(const float4*)&T[m * 256 + lane * 4];
        float mn = fminf(fminf(v.x, v.y), fminf(v.z, v.w)), mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
        if (lane == 0) { s_mn[m] = mn; s_mx[m] = mx; }
    }
    __syncthreads();
    float maxrange = 0.0f;
    for (int m = 0; m < Mpad; m++) maxrange = fmaxf(maxrange, s_mx[m] - s_mn[m]);
    const float scale = maxrange > 0.0f ? maxrange / 255.0f : 1.0f;
    const float inv = 1.0f / scale;
    // C: quantise; 4 consecutive codewords per lane -> one dword store
    for (int m = w; m < Mpad; m += 4) {
        float4 v = *(const float4*)&T[m * 256 + lane * 4];
        const float mn = s_mn[m];
        float vv[4] = {v.x, v.y, v.z, v.w};
        uint32_t pk = 0; float err = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float u = rintf((vv[j] - mn) * inv);
            u = fminf(fmaxf(u, 0.0f), 255.0f);
            pk |= (uint32_t)u << (8 * j);
            err = fmaxf(err, fabsf(vv[j] - (mn + scale * u)));
        }
        if (transposed) {
#pragma unroll
            for (int j = 0; j < 4; j++) lut8[pq_lut8_index(q, lane * 4 + j, m, Mpad, transposed)] = (uint8_t)(pk >> (8 * j));
        } else *(uint32_t*)&lut8[(q * Mpad + m) * 256 + lane * 4] = pk;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) err = fmaxf(err, __shfl_xor(err, off));
        if (lane == 0) s_err[m] = err;
    }
    __syncthreads();
    if (c == 0) {
        float absmax_sum = 0.0f, e_quant = 0.0f, bias = 0.0f;
        for (int m = 0; m < Mpad; m++) {
            absmax_sum += fmaxf(fabsf(s_mn[m]), fabsf(s_mx[m]));
            e_quant += s_err[m];
            bias += s_mn[m];
        }
        float d0 = 0.0f;
        for (int j = 0; j < nprobe; j++) { float d = probe_dis0[q * nprobe + j]; if (d > -__builtin_inff()) d0 = fmaxf(d0, fabsf(d)); }
        float B = absmax_sum + d0 + fabsf(bias) + scale * 255.0f * (float)M + 1.0f;
        PQQParam r; r.scale = scale; r.bias = bias; r.eps = e_quant * 1.0001f + pq_round_slack(M) * B; r.pad = 255.0f * (float)Mpad;   // no tighter bound computed on this path
        qp[q] = r;
    }
}

// Tiled form of the same table build for dsub = 8 (what k_pq_lut8f computes, bit for bit).  k_pq_lut8f reads the whole
// 786 KB codebook per query through one CU's L2 path (~47 GB/s): 4 rounds of ~30 us.  Here a workgroup takes the
// codebook slice of LT_MB sub-quantisers (each wave two of them, 4 codewords per lane in registers) and streams
// LT_QC queries past it, so the codebook is read LT_QC x less often; the global maximum range a query's quantisation
// scale needs is taken between two passes:
//   pass 0: per (query, m) min / max of the entries        -> mnmx [nq][Mpad][2]
//   pass 1: entries again, quantised with the query's scale -> lut8, per (query, m) max error -> err [nq][Mpad]
//   k_pq_qparam: per query, the sums over m in m order      -> {scale, bias, eps}
#ifdef RSX_MEASURE
// tools/ builds only (RSX_LUT_STEP, read once by launch_pq_lut8): table entries restricted to multiples of `step` (17 = 16 levels =
// what a 4-bit table could hold) so that the survivor count of a coarser table can be MEASURED on the real index before a kernel
// is written for it (profiles/r05_lut_bits_precheck.md).  eps follows by itself: it is the measured per-(query, m) maximum error.
__device__ int g_lut_step = 1;
__device__ __forceinline__ float lut_coarsen(float u) {
    const int st = g_lut_step;
    if (st <= 1) return u;
    const float fs = (float)st;
    return fs * fminf(rintf(u / fs), floorf(255.0f / fs));
}
#endif
#define LT_MB 8
#ifndef LT_QC
#define LT_QC 8      // queries per tile: 32 -> 8 quadruples the workgroups (1536 at batch 1024), measured 0.135 -> 0.100 ms for the build
#endif
template <int PASS>
__global__ __launch_bounds__(256) void k_pq_lut_tiled(const float* Q32, int ldq, const float* codebooks, int M, int Mpad,
                                                      int64_t nq, float* mnmx, float* errb, uint8_t* lut8, int transposed, float* lut32_out) {
    __shared__ float s_q[LT_QC * LT_MB * 8];     // the tile's query slices
    __shared__ float s_scale[2 * LT_QC];         // scale, 1 / scale
    __shared__ float s_mn[LT_QC * LT_MB];
    extern __shared__ __attribute__((aligned(16))) uint8_t lt_obuf[];   // transposed output: [LT_QC][256][LT_MB] bytes (64 KiB)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // 1-D grid, XCD-aware (round 3): workgroup b runs on XCD b % 8; the sub-quantiser blocks of ONE query tile get ids that differ by
    // multiples of 8, so they run on one XCD one after the other and their 8-byte pieces of the tile's 96-byte table rows merge in
    // that XCD's L2 into whole lines (with the 2-D grid they came from all 8 XCDs: 12 masked partial write-backs per row)
    const int nmb = (Mpad + LT_MB - 1) / LT_MB, nqt = (int)((nq + LT_QC - 1) / LT_QC);
    const int rest = (int)(blockIdx.x >> 3);
    const int qt = (rest / nmb) * 8 + (int)(blockIdx.x & 7);
    if (qt >= nqt) return;
    const int m0 = (rest % nmb) * LT_MB;
    const int64_t q0 = (int64_t)qt * LT_QC;
    const int nqc = (int)((nq - q0) < LT_QC ? (nq - q0) : LT_QC);
    for (int i = tid; i < LT_QC * LT_MB * 8; i += 256) {
        const int qi = i / (LT_MB * 8), t = i % (LT_MB * 8);
        const int m = m0 + t / 8;
        s_q[i] = (qi < nqc && m < M) ? Q32[(q0 + qi) * ldq + m0 * 8 + t] : 0.0f;
    }
    if (PASS == 1) {
        if (tid < nqc) {
            const float* mm = mnmx + (q0 + tid) * Mpad * 2;
            float maxrange = 0.0f;
            for (int m = 0; m < Mpad; m++) maxrange = fmaxf(maxrange, mm[2 * m + 1] - mm[2 * m]);
            const float scale = maxrange > 0.0f ? maxrange / 255.0f : 1.0f;
            s_scale[tid] = scale; s_scale[LT_QC + tid] = 1.0f / scale;
        }
        for (int i = tid; i < LT_QC * LT_MB; i += 256) {
            const int qi = i / LT_MB, m = m0 + i % LT_MB;
            s_mn[i] = (qi < nqc && m < Mpad) ? mnmx[((q0 + qi) * Mpad + m) * 2] : 0.0f;
        }
    }
    __syncthreads();
    // a wave owns sub-quantisers mi = w, w + 4 of the tile: its 4 codewords per lane stay in registers while the tile's
    // queries stream past (query values are LDS broadcasts); reductions over the 256 codewords are shuffles only
    for (int mi = w; mi < LT_MB; mi += 4) {
        const int m = m0 + mi;
        if (m >= Mpad) break;
        float4 cx[4], cy[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            cx[j] = make_float4(0.f, 0.f, 0.f, 0.f); cy[j] = cx[j];
            if (m < M) {
                const float4* cw = reinterpret_cast<const float4*>(codebooks + ((int64_t)m * 256 + lane + 64 * j) * 8);
                cx[j] = cw[0]; cy[j] = cw[1];
            }
        }
        for (int qi = 0; qi < nqc; qi++) {
            const int64_t q = q0 + qi;
            const float* qs = s_q + qi * (LT_MB * 8) + mi * 8;
            const float q0_ = qs[0], q1_ = qs[1], q2_ = qs[2], q3_ = qs[3], q4_ = qs[4], q5_ = qs[5], q6_ = qs[6], q7_ = qs[7];
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float t = 0.0f;
                t = __fmaf_rn(q0_, cx[j].x, t); t = __fmaf_rn(q1_, cx[j].y, t); t = __fmaf_rn(q2_, cx[j].z, t); t = __fmaf_rn(q3_, cx[j].w, t);
                t = __fmaf_rn(q4_, cy[j].x, t); t = __fmaf_rn(q5_, cy[j].y, t); t = __fmaf_rn(q6_, cy[j].z, t); t = __fmaf_rn(q7_, cy[j].w, t);
                v[j] = t;
            }
            if (PASS == 0) {
                float mn = fminf(fminf(v[0], v[1]), fminf(v[2], v[3])), mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
                if (lane == 0) { mnmx[(q * Mpad + m) * 2] = mn; mnmx[(q * Mpad + m) * 2 + 1] = mx; }
            } else {
                const float mn = s_mn[qi * LT_MB + mi], scale = s_scale[qi], inv = s_scale[LT_QC + qi];
                float err = 0.0f;
                if (lut32_out && m < M) {         // the fp32 table for k_pq_final_tab (round 4): 256-byte runs per (query, m, j)
#pragma unroll
                    for (int j = 0; j < 4; j++) lut32_out[(q * Mpad + m) * 256 + lane + 64 * j] = v[j];
                }
                // transposed ([q][code][m], the rotated-layout scans): bytes collect in LDS and leave as 8-byte runs of the
                // tile's 8 sub-quantisers (a byte store per entry at stride M costs 0.1 ms per batch)
                uint8_t* o = transposed ? lt_obuf + ((size_t)qi * 256 + lane) * LT_MB + mi : lut8 + (q * Mpad + m) * 256 + lane;
                const int ostep = transposed ? 64 * LT_MB : 64;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float u = rintf((v[j] - mn) * inv);
                    u = fminf(fmaxf(u, 0.0f), 255.0f);
#ifdef RSX_MEASURE
                    u = lut_coarsen(u);
#endif
                    o[ostep * j] = (uint8_t)u;
                    err = fmaxf(err, fabsf(v[j] - (mn + scale * u)));
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) err = fmaxf(err, __shfl_xor(err, off));
                if (lane == 0) errb[q * Mpad + m] = err;
            }
        }
    }
    if (PASS == 1 && transposed) {
        __syncthreads();
        for (int e = tid; e < nqc * 256; e += 256) {
            const int qi = e >> 8, c = e & 255;
            const uint2 v8 = *reinterpret_cast<const uint2*>(lt_obuf + (size_t)e * LT_MB);
            uint8_t* dst = lut8 + pq_lut8_index(q0 + qi, c, m0, Mpad, transposed);      // LT_MB = 8 consecutive m: one run in both transposed forms
            if (m0 + LT_MB <= Mpad) *reinterpret_cast<uint2*>(dst) = v8;
            else for (int t = 0; m0 + t < Mpad; t++) dst[t] = lt_obuf[(size_t)e * LT_MB + t];
        }
    }
}

// one wave per query: the per-query sums over m (lanes over m, tree order — any fixed order is fine: scale, bias and
// eps only have to be the values the scan and the certificate both use)
// AGENT: the (min, max) pairs and errors were stored by other workgroups of the SAME launch (k_pq_lut_once): agent-scope loads
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool AGENT>
__device__ __forceinline__ void pq_qparam_wave(int64_t q, int lane, int M, int Mpad, const float* mnmx, const float* errb,
                                               const float* probe_dis0, int nprobe, PQQParam* qp) {
    const float* mm = mnmx + q * Mpad * 2;
    float absmax_sum = 0.0f, maxrange = 0.0f, e_quant = 0.0f, bias = 0.0f, d0 = 0.0f;
    float mn_[2] = {0.0f, 0.0f}, mx_[2] = {0.0f, 0.0f};      // this lane's (min, max) pairs, m = lane and lane + 64 (Mpad <= 128 keeps both; else re-read)
    for (int m = lane; m < Mpad; m += 64) {
        const float mn = AGENT ? ld_agent(mm + 2 * m) : mm[2 * m], mx = AGENT ? ld_agent(mm + 2 * m + 1) : mm[2 * m + 1];
        if (m < 128) { mn_[m >> 6] = mn; mx_[m >> 6] = mx; }
        absmax_sum += fmaxf(fabsf(mn), fabsf(mx));
        maxrange = fmaxf(maxrange, mx - mn);
        e_quant += AGENT ? ld_agent(errb + q * Mpad + m) : errb[q * Mpad + m];
        bias += mn;
    }
    float mr = maxrange;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mr = fmaxf(mr, __shfl_xor(mr, off));
    // largest integer sum any code vector can reach with this query's table: per sub-quantiser the entry of its maximum,
    // quantised exactly as the table builder does (rint((mx - mn) / scale) clamped to 255) — the bound pair pruning uses
    const float inv_s = 1.0f / (mr > 0.0f ? mr / 255.0f : 1.0f);
    float smax = 0.0f;
    for (int m = lane; m < Mpad; m += 64) {
        const float mn = m < 128 ? mn_[m >> 6] : (AGENT ? ld_agent(mm + 2 * m) : mm[2 * m]), mx = m < 128 ? mx_[m >> 6] : (AGENT ? ld_agent(mm + 2 * m + 1) : mm[2 * m + 1]);
#ifdef RSX_MEASURE
        smax += lut_coarsen(fminf(fmaxf(rintf((mx - mn) * inv_s), 0.0f), 255.0f)) + (g_lut_step > 1 ? (float)g_lut_step : 0.0f);
#else
        smax += fminf(fmaxf(rintf((mx - mn) * inv_s), 0.0f), 255.0f);
#endif
    }
    for (int j = lane; j < nprobe; j += 64) { float d = probe_dis0[q * nprobe + j]; if (d > -__builtin_inff()) d0 = fmaxf(d0, fabsf(d)); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        absmax_sum += __shfl_xor(absmax_sum, off); e_quant += __shfl_xor(e_quant, off); bias += __shfl_xor(bias, off);
        smax += __shfl_xor(smax, off);
        maxrange = fmaxf(maxrange, __shfl_xor(maxrange, off)); d0 = fmaxf(d0, __shfl_xor(d0, off));
    }
    if (lane == 0) {
        const float scale = maxrange > 0.0f ? maxrange / 255.0f : 1.0f;
        // fp32 slack: every fp32 operation between the stored quantities and the two scores compared by the certificate rounds
        // by at most 2^-24 of a magnitude bounded by B — M adds in the exact score's table sum (+1 for dis0), M adds in bias,
        // M (mul, add, sub) in the per-entry error terms, and the fma + add of the approximate score: fewer than 3 M + 12
        const float B = absmax_sum + d0 + fabsf(bias) + scale * 255.0f * (float)M + 1.0f;
        PQQParam r; r.scale = scale; r.bias = bias; r.eps = e_quant * 1.0001f + pq_round_slack(M) * B; r.pad = smax;
        qp[q] = r;
    }
}
__global__ __launch_bounds__(64) void k_pq_qparam(int64_t nq, int M, int Mpad, const float* mnmx, const float* errb,
                                                  const float* probe_dis0, int nprobe, PQQParam* qp) {
    pq_qparam_wave<false>(blockIdx.x, threadIdx.x, M, Mpad, mnmx, errb, probe_dis0, nprobe, qp);
}

// ---------------------------------------------------------------------------------------
// The same tables on the matrix cores, in two launches instead of three (round 6; lut_tiled = 2).  The VALU form above costs 20 + 35 us
// (+ 5 for the parameters) per 1024 queries for 0.4 GFLOP: 60-odd VALU instructions and 12 dependent ds_bpermute shuffles per (query,
// sub-quantiser) and pass.  Here an entry is a column of a v_mfma_f32_32x32x2_f32 product — on gfx950 bit for bit the k-ordered fmaf
// chain (k_gemm.hip), i.e. exactly the value the VALU form computes:
//   * a wave owns ONE sub-quantiser m and the tile's 32 queries: A = 32 codewords x 2 dims per instruction (8 code tiles x 4 K steps,
//     32 registers for the whole codebook slice), B = the 32 queries' sub-vectors; lane (j, h) ends up with query j's entries of the 16
//     codes (r & 3) + 8 (r >> 2) + 4 h of every tile: min / max / error reductions are IN-LANE but for one exchange with lane ^ 32;
//   * PASS 0 leaves the (min, max) pairs, PASS 1 recomputes the entries (32 MFMAs) and quantises them with the query's scale; the bytes
//     leave through a padded LDS image [query][code] x 4 sub-quantisers as 4-byte runs of both transposed layouts;
//   * the per-query parameters come from the tile's first workgroup of PASS 1, with the quantisation error BOUNDED (scale / 2 per entry)
//     instead of measured — a measured sum needs every workgroup of the tile: a counter, agent-scope round trips and a tail of ~8 us
//     (built and measured: 48 us against 40 for the two launches);
// (One launch with the tile's workgroups meeting at a counter between the passes was built and measured first: 77 us — every hand-over
//  is a memory round trip of 2-3 us and the waiters hold their CU slots; profiles/r06_fixed_cost.md.)
// ---------------------------------------------------------------------------------------
template <int PASS>
__global__ __launch_bounds__(256) void k_pq_lut_mfma(const float* Q32, int ldq, const float* codebooks, int M, int Mpad, int64_t nq,
                                                     float* mnmx, uint8_t* lut8, int mode, float* lut32_out,
                                                     const float* probe_dis0, int nprobe, PQQParam* qp, PairGroupArgs pg) {
    __shared__ float s_scale[2 * LM_Q];
    extern __shared__ __attribute__((aligned(16))) uint8_t lt_obuf[];      // [LM_Q][257] dwords: byte mi of dword (query, code) = sub-quantiser m0 + mi
    unsigned bid = blockIdx.x;
    if (PASS == 1 && pg.nb > 0) {      // the first pg.nb workgroups group the (query, probe) pairs by list (rsx_internal.h: group_pairs_block)
        __shared__ int32_t pg_hist[PG_MAX_LPB + 16];
        if (bid < (unsigned)pg.nb) { group_pairs_block(pg, (int)bid, pg_hist); return; }
        bid -= (unsigned)pg.nb;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, hh = lane >> 5;
    const int nmb = (Mpad + LM_MB - 1) / LM_MB, nqt = (int)((nq + LM_Q - 1) / LM_Q);
    const int rest = (int)(bid >> 3);
    const int qt = (rest / nmb) * 8 + (int)(bid & 7);
    if (qt >= nqt) return;
    const int m0 = (rest % nmb) * LM_MB, m = m0 + w;
    const int64_t q0 = (int64_t)qt * LM_Q, q = q0 + j;
    const int nqc = (int)((nq - q0) < LM_Q ? (nq - q0) : LM_Q);
    const bool mreal = m < M, mpad = m < Mpad, qok = j < nqc;
    if (PASS == 0) {        // (min, max) per (query, m): rsx_internal.h (the same block can ride in the probe-pick launch)
        LutPass0Args p0{Q32, ldq, codebooks, M, Mpad, nq, mnmx, 0};
        pq_lut_pass0_block(p0, bid);
        return;
    }
    float bq[4], ac[8][4];
    lut_mfma_operands(Q32, ldq, codebooks, q, m, mreal, qok, j, hh, bq, ac);
    auto tile = [&](int t, float dep) { return lut_mfma_tile(ac, bq, t, dep); };
    // the queries' scales: 8 threads per query over m (max is order-free).  The tile's first workgroup also derives the queries' PQQParam
    // records here: everything they need is pass 0's (min, max) pairs and the coarse scores — the quantisation error is BOUNDED, not
    // measured: u = rint((v - mn) / scale) leaves |v - (mn + scale u)| <= scale (1/2 + 255 x 3 x 2^-24) for every entry, and the measured
    // maximum over a sub-quantiser's 256 entries is within a percent of that anyway.  (Summing measured errors needs every workgroup of the
    // tile: a counter, an agent-scope round trip and a tail of ~8 us in this launch.)
    {
        const int qi = tid >> 3, part = tid & 7;
        const int64_t qq = q0 + qi;
        const bool params = qp != nullptr && m0 == 0;
        float r = 0.0f, absmax_sum = 0.0f, bias = 0.0f, d0 = 0.0f;
        if (qi < nqc) {
            const float2* mm = reinterpret_cast<const float2*>(mnmx + qq * Mpad * 2);
#pragma unroll 4
            for (int mm_ = part; mm_ < Mpad; mm_ += 8) { const float2 p = mm[mm_]; r = fmaxf(r, p.y - p.x); absmax_sum += fmaxf(fabsf(p.x), fabsf(p.y)); bias += p.x; }
            if (params)
                for (int jj = part; jj < nprobe; jj += 8) { const float dv = probe_dis0[qq * nprobe + jj]; if (dv > -__builtin_inff()) d0 = fmaxf(d0, fabsf(dv)); }
        }
        r = fmaxf(r, __shfl_xor(r, 1)); r = fmaxf(r, __shfl_xor(r, 2)); r = fmaxf(r, __shfl_xor(r, 4));
        const float scale = r > 0.0f ? r / 255.0f : 1.0f, inv_s = 1.0f / scale;
        if (part == 0 && qi < nqc) { s_scale[qi] = scale; s_scale[LM_Q + qi] = inv_s; }
        if (params) {          // sums in a fixed order (m strided by 8 per thread, then a three-step tree): any fixed order is fine, see k_pq_qparam
            float smax = 0.0f;
            if (qi < nqc) {
                const float2* mm = reinterpret_cast<const float2*>(mnmx + qq * Mpad * 2);
#pragma unroll 4
                for (int mm_ = part; mm_ < Mpad; mm_ += 8) {
                    const float2 p = mm[mm_];
#ifdef RSX_MEASURE
                    smax += lut_coarsen(fminf(fmaxf(rintf((p.y - p.x) * inv_s), 0.0f), 255.0f)) + (g_lut_step > 1 ? (float)g_lut_step : 0.0f);
#else
                    smax += fminf(fmaxf(rintf((p.y - p.x) * inv_s), 0.0f), 255.0f);
#endif
                }
            }
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                absmax_sum += __shfl_xor(absmax_sum, off); bias += __shfl_xor(bias, off); smax += __shfl_xor(smax, off); d0 = fmaxf(d0, __shfl_xor(d0, off));
            }
            if (part == 0 && qi < nqc) {
                float e_quant = (float)M * scale * 0.50005f;
#ifdef RSX_MEASURE
                if (g_lut_step > 1) e_quant *= (float)g_lut_step;
#endif
                const float B = absmax_sum + d0 + fabsf(bias) + scale * 255.0f * (float)M + 1.0f;       // the fp32 slack: see pq_qparam_wave
                PQQParam pr; pr.scale = scale; pr.bias = bias; pr.eps = e_quant * 1.0001f + pq_round_slack(M) * B; pr.pad = smax;
                qp[qq] = pr;
            }
        }
    }
    const float mn = (qok && mpad) ? mnmx[(q * Mpad + m) * 2] : 0.0f;
    __syncthreads();
    const float inv = qok ? s_scale[LM_Q + j] : 1.0f;
    float dep = 0.0f;
    uint8_t* ob = lt_obuf + ((size_t)j * 257) * 4 + w;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const lm_f16 v = tile(t, dep);
        dep = v[15];
        if (lut32_out && mreal && qok) {
#pragma unroll
            for (int g = 0; g < 4; g++)
                *reinterpret_cast<float4*>(lut32_out + (q * Mpad + m) * 256 + 32 * t + 8 * g + 4 * hh) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int c = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hh;
            float u = rintf((v[r] - mn) * inv);
            u = fminf(fmaxf(u, 0.0f), 255.0f);
#ifdef RSX_MEASURE
            u = lut_coarsen(u);
#endif
            ob[(size_t)c * 4] = (uint8_t)u;
        }
    }
    __syncthreads();
    if (mode == 0) {        // [q][m][code]: 4 consecutive codes of one sub-quantiser per store
        for (int e = tid; e < nqc * LM_MB * 64; e += 256) {
            const int qi = e / (LM_MB * 64), mi = (e / 64) % LM_MB, c4 = (e % 64) * 4;
            if (m0 + mi >= Mpad) continue;
            const uint8_t* src = lt_obuf + ((size_t)qi * 257 + c4) * 4 + mi;
            const uint32_t v4 = (uint32_t)src[0] | ((uint32_t)src[4] << 8) | ((uint32_t)src[8] << 16) | ((uint32_t)src[12] << 24);
            *reinterpret_cast<uint32_t*>(lut8 + pq_lut8_index(q0 + qi, c4, m0 + mi, Mpad, 0)) = v4;
        }
    } else {                // [q][code][m] / [q][slice][code][m & 31]: the 4 sub-quantisers of one code per store
        for (int e = tid; e < nqc * 256; e += 256) {
            const int qi = e >> 8, c = e & 255;
            const uint32_t v4 = *reinterpret_cast<const uint32_t*>(lt_obuf + ((size_t)qi * 257 + c) * 4);
            uint8_t* dst = lut8 + pq_lut8_index(q0 + qi, c, m0, Mpad, mode);
            if (m0 + LM_MB <= Mpad) *reinterpret_cast<uint32_t*>(dst) = v4;
            else for (int t = 0; m0 + t < Mpad; t++) dst[t] = (uint8_t)(v4 >> (8 * t));
        }
    }
}

size_t pq_lut8_tiled_ws(int64_t nq, int Mpad) { return (size_t)nq * Mpad * 3 * 4; }   // mnmx + err
int pq_lut_pass0_blocks(int64_t nq, int Mpad) { return (int)(((Mpad + LM_MB - 1) / LM_MB) * ((((nq + LM_Q - 1) / LM_Q) + 7) / 8) * 8); }

size_t pq_lut8_fused_lds(int M, int Mpad, int dsub) { return ((size_t)Mpad * 256 + (size_t)M * dsub + 3 * (size_t)Mpad) * 4; }

void launch_pq_lut8(const float* lut32, const float* Q32, int ldq, const float* codebooks, int dsub, int64_t nq, int M,
                    int Mpad, const float* probe_dis0, int nprobe, uint8_t* lut8, void* qparam, void* ws, int transposed,
                    hipStream_t st, int phase, float* lut32_out, int mfma, const PairGroupArgs* pg) {
    if (nq <= 0) return;
    if (phase != 0 && !(ws && dsub == 8 && !lut32)) return;     // only the tiled build splits into tables (1) + per-query parameters (2)
    if (lut32) {
        hipLaunchKernelGGL(k_pq_lut8, dim3((unsigned)nq), dim3(256), 0, st, lut32, M, Mpad, probe_dis0, nprobe, lut8,
                           (PQQParam*)qparam, transposed);
        return;
    }
    if (ws && dsub == 8) {   // tiled: codebook slices shared by 32 queries
#ifdef RSX_MEASURE
        static int lut_step = -1;
        if (lut_step < 0) {
            const char* e = getenv("RSX_LUT_STEP");
            lut_step = e ? atoi(e) : 1;
            if (lut_step < 1 || lut_step > 255) lut_step = 1;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lut_step), &lut_step, sizeof(int));
        }
#endif
        float* mnmx = reinterpret_cast<float*>(ws);
        float* errb = mnmx + (size_t)nq * Mpad * 2;
        const int64_t nmb = (Mpad + LT_MB - 1) / LT_MB, nqt8 = (((nq + LT_QC - 1) / LT_QC) + 7) / 8;
        dim3 grid((unsigned)(nmb * nqt8 * 8));
        const size_t osm = transposed ? (size_t)LT_QC * 256 * LT_MB : 0;
        static DevOnce once;
        if (osm) once.once([&] { hipFuncSetAttribute((const void*)k_pq_lut_tiled<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)osm); });
        if (mfma && phase == 0) {      // matrix-core form: tables + per-query parameters in two launches (needs the coarse scores: phase 0 only)
            const int64_t nmb4 = (Mpad + LM_MB - 1) / LM_MB, nqt4 = (((nq + LM_Q - 1) / LM_Q) + 7) / 8;
            const size_t osm4 = (size_t)LM_Q * 257 * 4;
            const dim3 grid4((unsigned)(nmb4 * nqt4 * 8));
            PairGroupArgs pg0{};
            if (pg) pg0 = *pg;
            PairGroupArgs pgn{};
            if (mfma != 2)      // 2: pass 0 already ran as extra workgroups of the probe-pick launch (pq_lut_pass0_blocks)
                hipLaunchKernelGGL(k_pq_lut_mfma<0>, grid4, dim3(256), 0, st, Q32, ldq, codebooks, M, Mpad, nq, mnmx, lut8, transposed,
                                   lut32_out, probe_dis0, nprobe, (PQQParam*)nullptr, pgn);
            hipLaunchKernelGGL(k_pq_lut_mfma<1>, dim3(grid4.x + (unsigned)pg0.nb), dim3(256), osm4, st, Q32, ldq, codebooks, M, Mpad, nq, mnmx, lut8, transposed,
                               lut32_out, probe_dis0, nprobe, (PQQParam*)qparam, pg0);
            return;
        }
        if (phase != 2) {
            hipLaunchKernelGGL(k_pq_lut_tiled<0>, grid, dim3(256), 0, st, Q32, ldq, codebooks, M, Mpad, nq, mnmx, errb, lut8, transposed, (float*)nullptr);
            hipLaunchKernelGGL(k_pq_lut_tiled<1>, grid, dim3(256), osm, st, Q32, ldq, codebooks, M, Mpad, nq, mnmx, errb, lut8, transposed, lut32_out);
        }
        if (phase != 1)
            hipLaunchKernelGGL(k_pq_qparam, dim3((unsigned)nq), dim3(64), 0, st, nq, M, Mpad, mnmx, errb, probe_dis0, nprobe,
                               (PQQParam*)qparam);
        return;
    }
    const size_t lds = pq_lut8_fused_lds(M, Mpad, dsub);
    auto kern = dsub == 8 ? k_pq_lut8f<8> : k_pq_lut8f<0>;
    static DevOnce once8, once0;
    (dsub == 8 ? once8 : once0).once([&] { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL(kern, dim3((unsigned)nq), dim3(256), lds, st, Q32, ldq, codebooks, dsub, M, Mpad, probe_dis0, nprobe,
                       lut8, (PQQParam*)qparam, transposed, lut32_out);
}

// LDS byte offset (code * 4) of byte K of w in ONE instruction: the SDWA form of v_lshlrev selects the
// source byte for free (hipcc emits v_bfe + v_lshl_add for the same thing).
template <int K>
__device__ __forceinline__ uint32_t code_x4(uint32_t w, uint32_t two) {
    uint32_t r;
    if (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(two), "v"(w));
    if (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(two), "v"(w));
    if (K == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(two), "v"(w));
    if (K == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(two), "v"(w));
    return r;
}

// LDS byte offset (code * stride) of byte K of w in one instruction (24-bit multiply with SDWA byte select)
template <int K>
__device__ __forceinline__ uint32_t code_mul(uint32_t w, uint32_t stride) {
    uint32_t r;
    if (K == 0) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(stride), "v"(w));
    if (K == 1) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(stride), "v"(w));
    if (K == 2) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(stride), "v"(w));
    if (K == 3) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(stride), "v"(w));
    return r;
}

// VAR != 0 are MEASUREMENT-ONLY variants (wrong results; selected with RSX_SCAN8_VARIANT for the cost split
// in DESIGN.md): 1 = gathers kept, mask/shift accumulate replaced by one add; 2 = no LDS gather at all.
template <int NCH, int VPL, int VAR = 0, bool FILTER = false>
__global__ __launch_bounds__(1024) void k_pq_scan8(PQScan8Args A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t pq_lut4_s[];  // [256][TS] : byte i = query i
    constexpr int TS = NCH * 16 + 1;   // table row stride in dwords
    const PQScanArgs& a = A.b;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int l, gi, tile;
    if (!pq_decode_item(A.item_off, A.group_off, *A.total_items, A.nlist, l, gi, tile)) return;
    const int cnt = A.pair_off[l + 1] - A.pair_off[l];
    int np = cnt - 4 * gi; if (np > 4) np = 4;
    const int pair0 = A.pair_off[l] + 4 * gi;
    const int64_t len = a.list_len[l];
    const int64_t nslab = (len + 63) >> 6;
    const int64_t s0 = (int64_t)tile * (16 * VPL);
    if (s0 >= nslab) return;

    int pidx[4]; int64_t qq[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { pidx[i] = A.pairs_sorted[pair0 + (i < np ? i : 0)]; qq[i] = pidx[i] / a.nprobe; }
    // stage the four tables interleaved: thread handles 4 consecutive codes of one m
    {
        const int n4 = a.Mpad * 64;
        for (int i = tid; i < n4; i += 1024) {
            uint32_t in[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                in[k] = (k < np) ? *reinterpret_cast<const uint32_t*>(A.lut8 + qq[k] * a.Mpad * 256 + (int64_t)i * 4) : 0u;
            uint4 o;
            o.x = (in[0] & 0xffu) | ((in[1] & 0xffu) << 8) | ((in[2] & 0xffu) << 16) | ((in[3] & 0xffu) << 24);
            o.y = ((in[0] >> 8) & 0xffu) | (((in[1] >> 8) & 0xffu) << 8) | (((in[2] >> 8) & 0xffu) << 16) | (((in[3] >> 8) & 0xffu) << 24);
            o.z = ((in[0] >> 16) & 0xffu) | (((in[1] >> 16) & 0xffu) << 8) | (((in[2] >> 16) & 0xffu) << 16) | (((in[3] >> 16) & 0xffu) << 24);
            o.w = (in[0] >> 24) | ((in[1] >> 24) << 8) | ((in[2] >> 24) << 16) | ((in[3] >> 24) << 24);
            // table layout [code][m], row stride TS = Mpad + 1 dwords: the gather address is code * (4*TS) plus an
            // immediate m * 4, and bank = (code * TS + m) % 32 = (code + m) % 32 for Mpad % 32 == 0 -> as
            // conflict-random as the [m][code] layout.  i enumerates (m, 4 consecutive codes).
            const int m_ = i >> 6, c_ = (i & 63) * 4;
            pq_lut4_s[(c_ + 0) * TS + m_] = o.x;
            pq_lut4_s[(c_ + 1) * TS + m_] = o.y;
            pq_lut4_s[(c_ + 2) * TS + m_] = o.z;
            pq_lut4_s[(c_ + 3) * TS + m_] = o.w;
        }
    }

    const int64_t slab_base = a.list_base[l] >> 6;
    const int64_t slab_bytes = (int64_t)64 * a.Mpad;
    uint32_t two = 4 * TS;          // bytes per table row
    asm volatile("" : "+v"(two));   // keep the multiplier in a VGPR for the SDWA operand
    // per-query output parameters live in LDS behind the table: they are touched once per slab (768 gathers)
    // and would otherwise pin ~48 VGPRs through the gather loop
    float* prm_f = reinterpret_cast<float*>(pq_lut4_s + 256 * TS);             // [4][4]: dis0, scale, bias, -
    int64_t* prm_o = reinterpret_cast<int64_t*>(prm_f + 16);                    // [4][2]: temp offset | row column, query
    uint64_t* prm_t = reinterpret_cast<uint64_t*>(prm_o + 8);                   // [4]: threshold key
    if (tid < 4) {
        const int i = tid;
        PQQParam p = A.qp[qq[i]];
        const int64_t col = a.seg_start[qq[i] * (a.nprobe + 1) + (pidx[i] - (int)qq[i] * a.nprobe)];
        prm_f[i * 4 + 0] = a.probe_dis0[pidx[i]]; prm_f[i * 4 + 1] = p.scale; prm_f[i * 4 + 2] = p.bias;
        prm_f[i * 4 + 3] = 0.0f;
        prm_o[i * 2 + 0] = FILTER ? col : qq[i] * a.tstride + col;
        prm_o[i * 2 + 1] = qq[i];
        prm_t[i] = FILTER ? A.tau_key[qq[i] * A.tau_stride] : 0ull;
    }
    __syncthreads();
    // one slab: 64 vectors x NCH x 16 look-ups for the group's <= 4 queries, then the score / candidate output
    auto slab = [&](const uint4 (&c)[NCH], const int64_t s) {
        uint32_t acc02 = 0, acc13 = 0;   // 16-bit fields: queries (0,2) and (1,3); 96*255 < 65536
#pragma unroll
        for (int gg = 0; gg < NCH; gg++) {
            const uint32_t wds[4] = {c[gg].x, c[gg].y, c[gg].z, c[gg].w};
#pragma unroll
            for (int b = 0; b < 16; b++) {
                uint32_t off4;   // code * row bytes
                switch (b & 3) {
                    case 0: off4 = code_mul<0>(wds[b >> 2], two); break;
                    case 1: off4 = code_mul<1>(wds[b >> 2], two); break;
                    case 2: off4 = code_mul<2>(wds[b >> 2], two); break;
                    default: off4 = code_mul<3>(wds[b >> 2], two); break;
                }
                if (VAR == 2) { acc02 += off4; continue; }
                // raw LDS address (this kernel declares no static LDS, so the dynamic segment starts at 0):
                // (m * 4) goes into the ds_read offset field, no address add at all
                uint32_t e = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(
                    (uintptr_t)(off4 + (uint32_t)((gg * 16 + b) * 4)));
                if (VAR == 1) { acc02 += e; continue; }
                acc02 += e & 0x00ff00ffu;                                  // queries 0 and 2
                acc13 += __builtin_amdgcn_perm(e, e, 0x0c030c01u);         // [b1, 0, b3, 0]: queries 1 and 3
            }
        }
        const int64_t pos = s * 64 + lane;
        const uint32_t A4[4] = {acc02 & 0xffffu, acc13 & 0xffffu, acc02 >> 16, acc13 >> 16};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i >= np) continue;
            const float sc = prm_f[i * 4] + __fmaf_rn(prm_f[i * 4 + 1], (float)A4[i], prm_f[i * 4 + 2]);
            if (!FILTER) {
                a.temp[prm_o[i * 2] + pos] = (pos < len) ? sc : -__builtin_inff();
            } else {
                // candidate key in the same index space the score buffer would use (column of the query's row)
                const uint64_t key = (pos < len) ? make_key(sc, (uint32_t)(prm_o[i * 2] + pos)) : 0ull;
                const bool pass = key > prm_t[i];
                const uint64_t mask = __ballot(pass);
                if (mask) {   // wave-aggregated append: one atomic per wave per query
                    const int64_t q = prm_o[i * 2 + 1];
                    unsigned long long base = 0;
                    const int leader = __ffsll((unsigned long long)mask) - 1;
                    if (lane == leader) base = atomicAdd(&A.cand_cnt[q * CCS], (unsigned long long)__popcll(mask));
                    base = __shfl(base, leader);
                    const unsigned long long slot = base + __popcll(mask & ((1ull << lane) - 1ull));
                    if (pass && slot < (unsigned long long)A.cand_cap) A.cand[q * A.cand_cap + slot] = key;
                }
            }
        }
    };
    // Slabs in flight per wave (round 3).  The loop used to load a slab and use it at once: every iteration paid a full HBM
    // round trip (~2 us against ~0.3 us of gathers), which made the small-M scans latency-bound (M = 16, nprobe 512: 8.9 ms
    // where the LDS gathers need ~5).  Now PD slabs travel while one is summed.  Two register rings alternate (a slot's
    // refill lands in the OTHER ring: the words being summed stay live through the body, so an in-place refill would be a
    // copy behind a vmcnt(0)); the loop is unrolled 2 PD times so the slots are plain registers, and every wait is
    // `vmcnt((PD - 1) NCH)` on the entry and on the back edge alike (the prologue issues in slot order).
    // M >= 64 (NCH >= 4) keeps the plain form: those sizes run on the rotated layout, and 2 PD NCH x 4 registers would spill.
    constexpr int PD = NCH == 1 ? 4 : (NCH <= 3 ? 2 : 0);
    const int wu = __builtin_amdgcn_readfirstlane(w);      // scalar slab numbers: the loop's branches stay uniform
    if constexpr (PD == 0) {
        // rolled on purpose: one slab body is ~1000 instructions; unrolling VPL of them overflows the 64 KB instruction cache
#pragma unroll 1
        for (int u = 0; u < VPL; u++) {
            const int64_t s = s0 + wu + 16 * u;
            if (s >= nslab) break;
            const uint8_t* sp = a.codes + (slab_base + s) * slab_bytes;
            uint4 c[NCH];
#pragma unroll
            for (int gg = 0; gg < NCH; gg++) c[gg] = *reinterpret_cast<const uint4*>(sp + gg * 1024 + lane * 16);
            slab(c, s);
        }
    } else {
        uint4 ring[2][PD][NCH];
        auto fetch = [&](uint4 (&dst)[NCH], const int64_t sl) {
            const uint8_t* sp = a.codes + (slab_base + sl) * slab_bytes;
#pragma unroll
            for (int gg = 0; gg < NCH; gg++) dst[gg] = *reinterpret_cast<const uint4*>(sp + gg * 1024 + lane * 16);
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int p = 0; p < PD; p++) {
            const int64_t sl = s0 + wu + 16 * p;
            fetch(ring[0][p], (p < VPL && sl < nslab) ? sl : s0);     // s0 < nslab: a valid slab either way
        }
        bool done = false;
#pragma unroll 1
        for (int u0 = 0; u0 < VPL && !done; u0 += 2 * PD) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int p = 0; p < PD; p++) {
                    const int u = u0 + h * PD + p;
                    const int64_t s = s0 + wu + 16 * u;
                    if (done || u >= VPL || s >= nslab) { done = true; continue; }
                    const int64_t sn = s + 16 * PD;         // the slot's next slab goes to the other ring
                    if (u + PD < VPL && sn < nslab) fetch(ring[h ^ 1][p], sn);
                    slab(ring[h][p], s);
                }
            }
        }
    }
}

template <int NCH, int VPL, int VAR = 0, bool FILTER = false>
static int launch_pq_scan8_t(const PQScan8Args& A, hipStream_t st) {
    size_t shm = (size_t)(NCH * 16 + 1) * 256 * 4 + 256;   // [256][Mpad+1] table + per-query output parameters
    if (hipFuncSetAttribute((const void*)k_pq_scan8<NCH, VPL, VAR, FILTER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
        return -1;
    dim3 grid((unsigned)((A.max_items + 7) & ~7));
    hipLaunchKernelGGL((k_pq_scan8<NCH, VPL, VAR, FILTER>), grid, dim3(1024), shm, st, A);
    return 0;
}
template <int NCH, bool FILTER = false>
static int launch_pq_scan8_v(const PQScan8Args& A, int vpl, hipStream_t st) {
    switch (vpl) {
        case 16: return launch_pq_scan8_t<NCH, 16, 0, FILTER>(A, st);
        case 8: return launch_pq_scan8_t<NCH, 8, 0, FILTER>(A, st);
        case 4: return launch_pq_scan8_t<NCH, 4, 0, FILTER>(A, st);
        case 2: return launch_pq_scan8_t<NCH, 2, 0, FILTER>(A, st);
        default: return launch_pq_scan8_t<NCH, 1, 0, FILTER>(A, st);
    }
}
// returns 0 on launch, -1 if this (M, layout) has no fast-scan kernel
int launch_pq_scan8(const PQScanArgs& a, const uint8_t* lut8, const void* qparam, const int32_t* pairs_sorted,
                    const int32_t* pair_off, const int32_t* group_off, const int32_t* total_groups,
                    const int32_t* item_off, const int32_t* total_items, int nlist, int64_t max_items, int vpl,
                    hipStream_t st) {
    if (a.CB != 16 || max_items <= 0 || max_items > 0x7fffff00 || a.M * 255 >= 65536) return -1;
    PQScan8Args A;
    A.b = a; A.lut8 = lut8; A.qp = (const PQQParam*)qparam;
    A.pairs_sorted = pairs_sorted; A.pair_off = pair_off; A.group_off = group_off; A.total_groups = total_groups;
    A.item_off = item_off; A.total_items = total_items;
    A.nlist = nlist; A.max_items = (int)max_items;
    A.tau_key = nullptr; A.tau_stride = 0; A.cand = nullptr; A.cand_cnt = nullptr; A.cand_cap = 0; A.prune = 0; A.excl = nullptr; A.pace = 0; A.qitems = nullptr; A.qitems_tmax = 0;
#ifdef RSX_MEASURE     // cost-split variants (wrong results on purpose): tools/ builds only, not even instantiated in librsx.so
    if (a.Mpad == 96 && vpl == 8) {
        static const int var = measure_env("RSX_SCAN8_VARIANT", 0);
        if (var == 1) return launch_pq_scan8_t<6, 8, 1>(A, st);
        if (var == 2) return launch_pq_scan8_t<6, 8, 2>(A, st);
    }
#endif
    switch (a.Mpad / 16) {
        case 1: return launch_pq_scan8_v<1>(A, vpl, st);
        case 2: return launch_pq_scan8_v<2>(A, vpl, st);
        case 3: return launch_pq_scan8_v<3>(A, vpl, st);
        case 4: return launch_pq_scan8_v<4>(A, vpl, st);
        case 6: return launch_pq_scan8_v<6>(A, vpl, st);
        case 8: return launch_pq_scan8_v<8>(A, vpl, st);
        default: return -1;
    }
}

int launch_pq_scan8_filter(const PQScanArgs& a, const uint8_t* lut8, const void* qparam, const int32_t* pairs_sorted,
                           const int32_t* pair_off, const int32_t* group_off, const int32_t* total_groups,
                           const int32_t* item_off, const int32_t* total_items, int nlist, int64_t max_items, int vpl,
                           const uint64_t* tau_key, int64_t tau_stride, uint64_t* cand, unsigned long long* cand_cnt,
                           int cand_cap, hipStream_t st) {
    if (a.CB != 16 || max_items <= 0 || max_items > 0x7fffff00 || a.M * 255 >= 65536) return -1;
    PQScan8Args A;
    A.b = a; A.lut8 = lut8; A.qp = (const PQQParam*)qparam;
    A.pairs_sorted = pairs_sorted; A.pair_off = pair_off; A.group_off = group_off; A.total_groups = total_groups;
    A.item_off = item_off; A.total_items = total_items;
    A.nlist = nlist; A.max_items = (int)max_items;
    A.tau_key = tau_key; A.tau_stride = tau_stride; A.cand = cand; A.cand_cnt = cand_cnt; A.cand_cap = cand_cap; A.prune = 0; A.excl = nullptr; A.pace = 0; A.qitems = nullptr; A.qitems_tmax = 0;
    switch (a.Mpad / 16) {
        case 1: return launch_pq_scan8_v<1, true>(A, vpl, st);
        case 2: return launch_pq_scan8_v<2, true>(A, vpl, st);
        case 3: return launch_pq_scan8_v<3, true>(A, vpl, st);
        case 4: return launch_pq_scan8_v<4, true>(A, vpl, st);
        case 6: return launch_pq_scan8_v<6, true>(A, vpl, st);
        case 8: return launch_pq_scan8_v<8, true>(A, vpl, st);
        default: return -1;
    }
}

// ---------------------------------------------------------------------------------------
// Encode: 256 vectors per workgroup (one per thread); the codebook of subspace m is staged in LDS
// and read with broadcast ds_reads (all lanes the same address: conflict-free).
template <int DSUB>
__global__ __launch_bounds__(256) void k_pq_encode(const void* x, int x_f16, int64_t n, int ldx, int d, int M, int Mpad,
                                                   int CB, int dsub_rt, const float* centroids, const int32_t* assign,
                                                   const float* codebooks, const int64_t* dest_row, uint8_t* codes,
                                                   uint8_t* plain_out) {
    extern __shared__ __attribute__((aligned(16))) float enc_cb[];  // [256][dsub]
    const int dsub = DSUB > 0 ? DSUB : dsub_rt;
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 256 + tid;
    const int64_t ii = i < n ? i : n - 1;
    const int64_t drow = (dest_row && i < n) ? dest_row[ii] : 0;
    const bool valid = i < n && drow >= 0;       // dest < 0: the vector's list belongs to another shard (rsx_set_param "add_list_mod")
    const float* cen = centroids ? centroids + (int64_t)assign[ii] * d : nullptr;
    uint32_t packed = 0;
    for (int m = 0; m < Mpad; m++) {
        uint32_t code = 0;
        if (m < M) {
            __syncthreads();
            for (int e = tid; e < 256 * dsub; e += 256) enc_cb[e] = codebooks[(int64_t)m * 256 * dsub + e];
            __syncthreads();
            float r[DSUB > 0 ? DSUB : 1];
            if (DSUB > 0) {
#pragma unroll
                for (int t = 0; t < DSUB; t++) {
                    float xv = x_f16 ? __half2float(((const __half*)x)[ii * ldx + m * dsub + t])
                                     : ((const float*)x)[ii * ldx + m * dsub + t];
                    r[t] = cen ? __fsub_rn(xv, cen[m * dsub + t]) : xv;
                }
            }
            float best = __builtin_inff();
            for (int c = 0; c < 256; c++) {
                float acc = 0.0f;
                if (DSUB > 0) {
#pragma unroll
                    for (int t = 0; t < DSUB; t++) { float df = __fsub_rn(r[t], enc_cb[c * DSUB + t]); acc = __fmaf_rn(df, df, acc); }
                } else {
                    for (int t = 0; t < dsub; t++) {
                        float xv = x_f16 ? __half2float(((const __half*)x)[ii * ldx + m * dsub + t])
                                         : ((const float*)x)[ii * ldx + m * dsub + t];
                        float rv = cen ? __fsub_rn(xv, cen[m * dsub + t]) : xv;
                        float df = __fsub_rn(rv, enc_cb[c * dsub + t]);
                        acc = __fmaf_rn(df, df, acc);
                    }
                }
                if (acc < best) { best = acc; code = (uint32_t)c; }
            }
        }
        if (plain_out) {
            if (valid) plain_out[i * Mpad + m] = (uint8_t)code;
        } else if (pq_rot_family(CB)) {      // rotated / sliced layout: consecutive m of a vector are not contiguous
            if (valid) codes[pq_code_addr(drow, m, Mpad, CB)] = (uint8_t)code;
        } else {
            packed |= code << (8 * (m & 3));
            if ((m & 3) == 3) {
                if (valid) {
                    int64_t slab = drow >> 6; int v = (int)(drow & 63);
                    int m0 = m - 3;
                    int g = m0 / CB, b = m0 - g * CB;
                    int64_t addr = (slab * (Mpad / CB) + g) * (int64_t)(64 * CB) + v * CB + b;
                    *reinterpret_cast<uint32_t*>(codes + addr) = packed;
                }
                packed = 0;
            }
        }
    }
}

void launch_pq_encode(const void* x, int x_f16, int64_t n, int ldx, int d, int M, int Mpad, int CB,
                      const float* centroids, const int32_t* assign, const float* codebooks,
                      const int64_t* dest_row, uint8_t* codes, uint8_t* plain_out, hipStream_t st) {
    if (n <= 0) return;
    int dsub = d / M;
    dim3 grid((unsigned)((n + 255) / 256));
    size_t shm = (size_t)256 * dsub * sizeof(float);
#define RSX_ENC(DS)                                                                                         \
    case DS:                                                                                                \
        if (shm > 48 * 1024) hipFuncSetAttribute((const void*)k_pq_encode<DS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
        hipLaunchKernelGGL(k_pq_encode<DS>, grid, dim3(256), shm, st, x, x_f16, n, ldx, d, M, Mpad, CB, dsub, centroids, assign, \
                           codebooks, dest_row, codes, plain_out);                                          \
        break;
    switch (dsub) {
        RSX_ENC(2) RSX_ENC(4) RSX_ENC(8) RSX_ENC(12) RSX_ENC(16) RSX_ENC(24) RSX_ENC(32) RSX_ENC(48) RSX_ENC(64)
        default:
            if (shm > 48 * 1024) hipFuncSetAttribute((const void*)k_pq_encode<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            hipLaunchKernelGGL(k_pq_encode<0>, grid, dim3(256), shm, st, x, x_f16, n, ldx, d, M, Mpad, CB, dsub, centroids,
                               assign, codebooks, dest_row, codes, plain_out);
    }
#undef RSX_ENC
}

}  // namespace rsx
