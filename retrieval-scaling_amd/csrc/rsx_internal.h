// rsx_internal.h — internal declarations shared by the HIP kernels and the C-ABI host code.
// gfx950 (MI355X / CDNA4) only: wave64, MFMA, 160 KiB LDS.  No portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

namespace rsx {

// ---------------------------------------------------------------------------------------
// Candidate key: 64-bit, larger = better.  high 32 = order-preserving map of the fp32 score,
// low 32 = ~idx so that among equal scores the smaller idx wins.  0 = empty slot.
// ---------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t f2ord(float f) {
    union { float f; uint32_t u; } c; c.f = f;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t o) {
    union { float f; uint32_t u; } c;
    c.u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return c.f;
}
// Only scores > -inf are admitted (FAISS: strict comparison against a -inf heap top); NaN never.
__device__ inline uint64_t make_key(float s, uint32_t idx) {
    s = s + 0.0f;  // -0 -> +0
    if (!(s > -__builtin_inff())) return 0ull;
    return ((uint64_t)f2ord(s) << 32) | (uint64_t)(0xFFFFFFFFu - idx);
}
__host__ __device__ inline uint32_t key_idx(uint64_t k) { return 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull); }
__host__ __device__ inline float key_score(uint64_t k) { return ord2f((uint32_t)(k >> 32)); }

enum { KIND_FLAT = 0, KIND_IVFFLAT = 1, KIND_IVFPQ = 2 };

// Environment switches between kernel generations / cost-split variants exist ONLY in the -DRSX_MEASURE build that tools/ load
// through RSX_LIB (csrc/Makefile: librsx_measure.so).  In the shipped librsx.so this is the constant `dflt`: no environment
// variable can change which kernel serves a search, let alone make one skip part of its work.
#ifdef RSX_MEASURE
#include <cstdlib>
inline int measure_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
constexpr int measure_env(const char*, int dflt) { return dflt; }
#endif

// hipFuncSetAttribute (dynamic LDS above 64 KiB) is a per-DEVICE property of a kernel: with several GPUs driven from one
// process (rsx_sharded_create) every device needs its own call.  first() is true once per device; need(bytes) is true when
// this device has not yet been granted that many bytes.
inline int cur_device() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
// Both are used from the host threads of a sharded handle concurrently: the check, the attribute call and the update happen
// under one process-wide mutex, so no thread can launch a kernel between another thread's "first" and its attribute call.
inline std::mutex& attr_mutex() { static std::mutex m; return m; }
struct DevOnce {
    uint64_t m = 0;
    template <typename F> void once(F&& f) {          // f(): the per-device set-up; runs once per device
        std::lock_guard<std::mutex> g(attr_mutex());
        const uint64_t b = 1ull << cur_device();
        if (m & b) return;
        f();
        m |= b;
    }
};
struct DevSize {
    size_t v[64] = {};
    template <typename F> void grow(size_t bytes, F&& f) {   // f(): grant this device `bytes`; runs when it has fewer
        std::lock_guard<std::mutex> g(attr_mutex());
        size_t& s = v[cur_device()];
        if (bytes <= s) return;
        f();
        s = bytes;
    }
};

// Per-query candidate counters are atomically bumped from every CU: one counter per 128-byte line (8 B used), so that
// the device-scope atomics of different queries never serialise on a shared line (measured: ~0.7 ms of a 3.8 ms scan).
constexpr int CCS = 16;   // counter stride in 8-byte words: cand_cnt[q * CCS]

// ---------------------------------------------------------------------------------------
// IVF-PQ code layouts (CB = bytes of one vector that sit together):
//   CB = 16 / 4 ("granule"): slab of 64 vectors, byte(slab, granule g, lane v, b) = (slab*Mpad/CB + g)*64*CB + v*CB + b,
//             m = g*CB + b — every lane of a wave holds the SAME sub-quantiser at each step.
//   CB = 0 ("rotated", M % 32 == 0, M <= 128): block of 16 vectors = 16*M bytes; a wave lane (g, i) = (lane >> 4, lane & 15)
//             owns vector i of the block and reads, per 64-sub-quantiser phase p, 16 contiguous bytes at
//             p*1024 + lane*16 (byte s: m = 64p + 16g + ((i + s) & 15)) and, for a trailing 32-sub-quantiser phase, 8
//             contiguous bytes at NF*1024 + lane*8 (byte s: m = 64NF + 16(g & 1) + ((i + s + 8(g >> 1)) & 15)).
//             At every step the 32 lanes of a half-wave hold 32 different m % 32, so a table laid out [code][m]
//             (bank = m % 32) is gathered without a single LDS bank conflict whatever the codes are (k_pq_scan_rot).
//             M = 16: block of 64 vectors = 1 KiB; lane (g, i) owns vector 16 g + i and reads its 16 contiguous bytes at lane*16
//             (byte s: m = (i + s) & 15); the table row of a code holds the 16 entries twice (bytes 0-63 and 64-127), lane groups
//             of even / odd g use one copy each, so the 32 lanes of a half-wave again hit 32 different banks.
//   CB = PQ_SLICED (-1; "sliced", round 6, M % 32 == 0): block of 32 vectors, cut into M/32 SLICES of 1 KiB — slice s holds
//             sub-quantisers 32 s .. 32 s + 31 of all 32 vectors.  Within a slice a wave lane (g, i) = (lane >> 4, lane & 15) owns vector
//             16 (g >> 1) + i and reads 16 contiguous bytes at lane*16 (byte b: m = 32 s + 16 (g & 1) + ((i + b) & 15)).  The 32 lanes of
//             a half-wave hold 32 different m % 32 at every step, as in the rotated layout — but a pass over ONE slice needs only that
//             slice's table (64 KiB with 8-byte entries = eight queries per ds_read_b64), so a scan can keep two slices' tables in
//             LDS and restage the third behind a pass (k_pq_scan_sl8).  A 16-sub-quantiser run of a vector is one 16-byte piece.
//             PQ_SLICED_GB = 16 consecutive blocks (512 vectors) form a GROUP stored slice-major: [group][slice][block in group][1 KiB] —
//             the 16 waves of a scan workgroup read the same slice of 16 consecutive blocks at a time, one contiguous 16 KiB run instead of
//             sixteen 1 KiB pieces 3 KiB apart.  Lists of this layout start on 512-vector boundaries (rsx_index::row_align).
// Lists start on 64-vector boundaries in the other layouts and a 64-vector slab is 64*Mpad bytes in them.
// ---------------------------------------------------------------------------------------
constexpr int PQ_SLICED = -1;
#ifndef PQ_SLICED_GB_V
#define PQ_SLICED_GB_V 16
#endif
constexpr int PQ_SLICED_GB = PQ_SLICED_GB_V;          // blocks per slice-major group (A/B builds override it: the layout of every kernel follows)
// byte offset of slice `sl` of the 32-vector block `blk` (absolute, or relative to a group-aligned list start) in the sliced layout
__host__ __device__ inline int64_t pq_sliced_off(int64_t blk, int sl, int Mpad) {
    return (blk / PQ_SLICED_GB) * (int64_t)(PQ_SLICED_GB * 32 * Mpad) + (int64_t)sl * (PQ_SLICED_GB * 1024) + (blk % PQ_SLICED_GB) * 1024;
}
__host__ __device__ inline int64_t pq_code_addr(int64_t row, int m, int Mpad, int CB) {
    if (CB == PQ_SLICED) {
        const int v = (int)(row & 31), i = v & 15, mm = m & 31;
        const int lane = 16 * (2 * (v >> 4) + (mm >> 4)) + i;
        return pq_sliced_off(row >> 5, m >> 5, Mpad) + lane * 16 + (((mm & 15) - i) & 15);
    }
    if (CB != 0) {
        const int64_t slab = row >> 6; const int v = (int)(row & 63);
        const int g = m / CB, b = m - g * CB;
        return (slab * (Mpad / CB) + g) * (int64_t)(64 * CB) + v * CB + b;
    }
    if (Mpad == 16) {     // rotated, M = 16 (round 3): a block is 64 vectors = 1 KiB; vector v's 16 bytes sit together, byte s holds m = (v + s) & 15
        const int v = (int)(row & 63);
        return (row >> 6) * 1024 + v * 16 + ((m - v) & 15);
    }
    const int64_t base = (row >> 4) * (int64_t)(16 * Mpad);
    const int i = (int)(row & 15), NF = Mpad >> 6;
    if (m < 64 * NF) {
        const int p = m >> 6, mm = m & 63, g = mm >> 4, s = ((mm & 15) - i) & 15;
        return base + p * 1024 + (16 * g + i) * 16 + s;
    }
    const int mm = m - 64 * NF, t = ((mm & 15) - i) & 15, g = (mm >> 4) + 2 * (t >> 3), s = t & 7;
    return base + NF * 1024 + (16 * g + i) * 8 + s;
}
__host__ __device__ inline bool pq_rot_applies(int M) { return M == 16 || (M % 32 == 0 && M >= 32 && M <= 128); }
__host__ __device__ inline bool pq_sliced_applies(int M) { return M == 96; }      // the M whose 8-query table (M * 2 KiB) exceeds the LDS
__host__ __device__ inline bool pq_rot_family(int CB) { return CB == 0 || CB == PQ_SLICED; }
// 8-bit table layouts (launch_pq_lut8 `transposed`): 0 = [q][m][code] (granule scans), 1 = [q][code][m] (rotated layout: a code's row feeds
// the table image), 2 = [q][slice m >> 5][code][m & 31] (sliced layout: the 8 KiB table of one (query, slice) is contiguous — the scan
// re-stages single slices, and a [code][m] table would hand it a third of every cache line it touches)
__host__ __device__ inline int64_t pq_lut8_index(int64_t q, int c, int m, int Mpad, int mode) {
    return mode == 0 ? (q * Mpad + m) * 256 + c : mode == 1 ? (q * 256 + c) * Mpad + m : ((q * (Mpad >> 5) + (m >> 5)) * 256 + c) * 32 + (m & 31);
}     // block layouts: transposed tables, work-item scans
// The two 8-byte halves of the piece that holds sub-quantisers 16 run .. 16 run + 15 of vector `row` (M >= 32, rotated or sliced layout);
// rotated left by row & 15 bytes (rot16_bytes, k_select.hip) the 16 bytes are the codes in m order.
__device__ inline void pq_piece_ptrs(const uint8_t* codes, int64_t row, int M, int CB, int run, const uint8_t*& p0, const uint8_t*& p1) {
    const int i = (int)(row & 15);
    if (CB == PQ_SLICED) {
        p0 = codes + pq_sliced_off(row >> 5, run >> 1, M) + (16 * (2 * (int)((row >> 4) & 1) + (run & 1)) + i) * 16; p1 = p0 + 8;
        return;
    }
    const uint8_t* base = codes + (row >> 4) * (int64_t)(16 * M);
    const int NF = M >> 6;
    // run < 4 NF: the 16-byte piece of lane group run & 3 in phase run >> 2; later runs: the two 8-byte pieces of lane groups h and h + 2
    // of the half phase (h = run & 1)
    if (run < 4 * NF) { p0 = base + (run >> 2) * 1024 + ((run & 3) * 16 + i) * 16; p1 = p0 + 8; }
    else { const int h = run & 1; p0 = base + NF * 1024 + (h * 16 + i) * 8; p1 = base + NF * 1024 + ((h + 2) * 16 + i) * 8; }
}

// Inverted-list directory on the device (one entry per list; Flat uses a single list 0).
//   base : first storage row of the list (PQ: multiple of 64 = slab aligned; flat rows: of 16)
//   len  : vectors in the list
//   cap  : rows reserved
struct ListDir {
    const int64_t* base;
    const int64_t* len;
};

// ---------------------------------------------------------------------------------------
// Launch wrappers (implemented in the .hip files).  All asynchronous on `st`.
// ---------------------------------------------------------------------------------------

// k_misc.hip
void launch_convert_to_f32(const void* src, int src_f16, int64_t src_ld, int64_t n_rows, int d, float* dst, int ld, hipStream_t st);
void launch_convert_to_f32_f16(const void* src, int src_f16, int64_t src_ld, int64_t n_rows, int d, float* dst32, __half* dst16, int ld,
                               int64_t pad_rows_to, hipStream_t st);
void launch_convert_to_f16(const void* src, int src_f16, int64_t n_rows, int d, __half* dst, int ld,
                           int64_t pad_rows_to, int* not_representable_flag, hipStream_t st);
void launch_check_f16(const float* x, int64_t count, int* flag, hipStream_t st);
void launch_write_ids(const int64_t* dest_row, const int64_t* ids_in, int64_t id0, int64_t n, int64_t* ids_storage, hipStream_t st);
void launch_widen_storage(const __half* src, float* dst, int64_t count, hipStream_t st);
void launch_residuals(const float* x, int64_t n, int d, const float* centroids, const int32_t* assign, float* out, hipStream_t st);
void launch_fill_u64(uint64_t* p, int64_t n, uint64_t v, hipStream_t st);
void launch_keep_last_u64(uint64_t* p, int64_t rows, int len, hipStream_t st);   // zero all but the last key of each row
void launch_fill_f32(float* p, int64_t n, float v, hipStream_t st);
void launch_synth_vectors(int d, int ncentres, uint32_t seed_c, uint32_t seed_x, float sigma, int64_t i0,
                          int64_t n, __half* out, hipStream_t st);
void launch_synth_queries(int d, int ncentres, uint32_t seed_c, uint32_t seed_x, float sigma, int64_t nbase,
                          uint32_t seed_q, float sigma_q, int64_t r0, int64_t n, __half* out, hipStream_t st);
// scatter rows of a batch into list storage (IVF-Flat / Flat): dst row = dest_row[i]
void launch_kmeans_accumulate(const float* x, int64_t ldx, int col_stride, int d, int k, int nsets, int64_t n, const int32_t* order,
                              const int32_t* seg_off, float* sums, hipStream_t st);
void launch_max_norm2(const void* x, int x_f16, int64_t n, int d, unsigned int* out_bits, hipStream_t st);
void launch_scatter_rows(const void* x, int x_f16, int64_t n, int d, const int64_t* dest_row, void* storage,
                         int storage_f16, int ld, float* norms, const int64_t* ids_in, int64_t id0,
                         int64_t* ids_storage, hipStream_t st);
// destination rows of an add batch on the device (k_misc.hip): per-list totals first (the host grows the lists), then the rows
int64_t add_dest_segments(int64_t n);
void launch_add_destinations(const int32_t* assign, int64_t n, int nlist, int lmod, int lrem, int32_t* seg_cnt, int32_t* total, hipStream_t st);
void launch_add_place(const int32_t* assign, int64_t n, int nlist, int lmod, int lrem, const int32_t* seg_off, const int64_t* start,
                      int64_t* dest, hipStream_t st);
// copy lists between two layouts (re-layout on growth). row_bytes = bytes per row group unit.
void launch_copy_lists(int nlist, const int64_t* old_base, const int64_t* new_base, const int64_t* len,
                       const uint8_t* old_data, uint8_t* new_data, int64_t unit_rows, int64_t unit_bytes,
                       const int64_t* old_ids, int64_t* new_ids, const float* old_norms, float* new_norms,
                       hipStream_t st);
// export one PQ list from the interleaved slab layout to plain [n, M]
void launch_pq_export_list(const uint8_t* codes, int64_t base_row, int64_t n, int M, int Mpad, int CB,
                           uint8_t* out, hipStream_t st);
// import plain codes [n, M] into the slab layout at rows base_row + pos0 ...
void launch_pq_import_list(const uint8_t* plain, int64_t base_row, int64_t pos0, int64_t n, int M, int Mpad,
                           int CB, uint8_t* codes, hipStream_t st);

// k_gemm.hip
// S[i, c] = <X_i, C_c> as one sequential fp32 fmaf chain over t = 0..d-1 (f32-input MFMA).
// X: [n, ldx] f32 or f16;  C: [nc, d] f32;  S: [n, lds_] f32.
void launch_gemm_exact_scores(const void* X, int x_f16, int64_t n, int ldx, const float* C, int nc, int d,
                              float* S, int64_t lds_, hipStream_t st);
// assign[i] = argmax_c <X_i, C_c> (first maximum).  partial: workspace [n, 2*ceil(nc/128)] u64.
void launch_gemm_exact_argmax(const void* X, int x_f16, int64_t n, int ldx, const float* C, int nc, int d,
                              uint64_t* partial, int32_t* assign, float* best, hipStream_t st);
// Flat scan: temp[q, v] = <Q_q, X_v> (fp16 MFMA, fp32 accumulate) for v in [v0, v0+nv).
// Q16: [nq_pad(128), ld] f16; X: [., ld] f16|f32 storage (ld multiple of 64); bias: per-row or null.
void launch_flat_gemm(const __half* Q16, int nq_pad, const void* X, int x_f16, int64_t v0, int64_t nv,
                      int ld, const float* bias, float* temp, int64_t tstride, hipStream_t st);
void launch_flat_gemm_filter(const __half* Q16, int nq_pad, int nq, const void* X, int x_f16, int64_t v0, int64_t nv, int ld,
                             const float* bias, const uint64_t* tau, int64_t tau_stride, uint64_t* cand,
                             unsigned long long* cand_cnt, int cand_cap, hipStream_t st);
// List scan (IVF-Flat and small-batch Flat): groups of <=16 (query, probe) pairs per list.
struct ListScanArgs {
    const __half* Q16; int ld;             // queries [nq, ld]
    const void* X; int x_f16;              // storage rows
    const float* bias;                     // per storage row (L2: -0.5|x|^2) or null
    const int64_t* list_base; const int64_t* list_len;
    // grouping (IVF): pairs sorted by list
    const int32_t* pairs_sorted;           // [P] pair index = q*nprobe + j
    const int32_t* pair_off;               // [nlist+1]
    const int32_t* group_off;              // [nlist+1]
    const int32_t* total_groups;           // device scalar
    const int32_t* probe_list;             // [nq*nprobe]
    const int64_t* seg_start;              // [nq, nprobe+1]
    int nlist; int nprobe;
    int flat_mode; int64_t flat_n; int nq;  // flat_mode: single list, groups = blocks of 16 queries
    float* temp; int64_t tstride;
    int chunk_rows;                         // rows per work item (multiple of 64)
    int qtiles;                             // k_list_scan2: 16-query tiles per group (1, 2, 4; 8 = k_list_scan3): the grouping must have used 16 x qtiles
    int max_groups; int max_chunks;
    // filtered output (k_list_scan2 only; tau_key != null): keys > tau_key[q * tau_stride] are appended to
    // cand[q][0..cand_cap) (count in cand_cnt[q]) instead of storing every score
    const uint64_t* tau_key; int64_t tau_stride; uint64_t* cand; unsigned long long* cand_cnt; int cand_cap;
    // k_list_scan2, optional (round 4): a 1-D grid over the (list, chunk, group) work items in list-major order, decoded XCD-aware
    // (pq_decode_item below) — the query groups of one list chunk then run on ONE XCD at the same time and the chunk's rows are
    // fetched from HBM once (with the 2-D grid the groups of a list landed on different XCDs: one fetch per group).  item_off /
    // total_items from launch_group_pairs(tile_rows = chunk_rows); max_items = the grid.
    const int32_t* item_off; const int32_t* total_items; int max_items;
    // unfiltered k_list_scan2, optional: the scores of (query q, probe rank j) go to temp[q * tstride + j * pre_stride + row] instead of
    // the query's concatenated row (threshold pre-pass over the first rows of several lists: every list gets its own slice)
    int64_t pre_stride;
};
int list_scan2_chunk_rows(int x_f16, int ld);   // work-item rows of the LDS-DMA list scan, 0 if it does not apply
int list_scan2_max_qtiles(int ld);              // ... and the 16-query tiles per group its LDS holds
int list_scan3_applies(int x_f16, int ld, int has_bias);          // the query-stationary form (qtiles = 8: 128 queries per group, 1024 rows per work item) applies: fp16 rows, d = 384 / 512 / 768 / 1024
void launch_list_scan(const ListScanArgs& a, hipStream_t st);

// k_pq.hip
void launch_pq_lut(const float* Q32, int ldq, int64_t nq, int d, int M, int Mpad, const float* codebooks,
                   float* lut, hipStream_t st);
struct PQScanArgs {
    const uint8_t* codes; int M; int Mpad; int CB;
    const int64_t* list_base; const int64_t* list_len;
    const float* lut;                       // [nq, Mpad, 256]
    const int32_t* probe_list; const float* probe_dis0; const int64_t* seg_start;
    int64_t nq; int nprobe;
    float* temp; int64_t tstride;
    int slabs_per_chunk; int max_chunks;
};
int launch_pq_scan(const PQScanArgs& a, hipStream_t st);  // returns 0, or -1 if the LDS request cannot be met
// IVF-PQ, METRIC_L2: per-(query, list) table of squared sub-vector distances built in LDS, exact scan, scores = -distance (k_pq_rot.hip)
int launch_pq_scan_l2(const PQScanArgs& a, const float* Q32, int ldq, const float* centroids, const float* codebooks, int d, int dsub, hipStream_t st);
// list-major two-queries-per-LDS-read scan; pairs grouped by list in groups of 2 (launch_group_pairs)
int launch_pq_scan2(const PQScanArgs& a, const int32_t* pairs_sorted, const int32_t* pair_off, const int32_t* group_off,
                    const int32_t* total_groups, const int32_t* item_off, const int32_t* total_items, int nlist,
                    int64_t max_items, int vpl, hipStream_t st);
// 8-bit table fast scan (4 queries per LDS read) — approximate scores, certified by k_finalize
// lut32 null -> fused form: the fp32 table is built in LDS from Q32 [nq, ldq] and the codebooks (needs
// pq_lut8_fused_lds(M, Mpad, dsub) <= 160 KiB); otherwise quantises the given fp32 tables.
struct PairGroupArgs;
size_t pq_lut8_fused_lds(int M, int Mpad, int dsub);
void launch_pq_lut8(const float* lut32, const float* Q32, int ldq, const float* codebooks, int dsub, int64_t nq, int M,
                    int Mpad, const float* probe_dis0, int nprobe, uint8_t* lut8,
                    void* qparam /* [nq] {scale, bias, eps, pad} */,
                    void* ws /* pq_lut8_tiled_ws(nq, Mpad) bytes -> tiled build (dsub 8), or null */,
                    int transposed /* pq_lut8_index: 0: lut8 [nq][Mpad][256]; 1: [nq][256][Mpad] (rotated-layout scans); 2: [nq][Mpad/32][256][32] (sliced) */, hipStream_t st,
                    int phase = 0 /* tiled build only: 1 = the tables (independent of the probe selection), 2 = the per-query parameters */,
                    float* lut32_out = nullptr /* fused forms only: also store the fp32 tables [nq][Mpad][256] (k_pq_final_tab reads them) */,
                    int mfma = 0 /* tiled build, phase 0: the matrix-core form (k_pq_lut_mfma: tables and per-query parameters in two launches; 2: its
                                    first pass already ran in the probe-pick launch) */,
                    const struct PairGroupArgs* pg = nullptr /* matrix-core form: also group the (query, probe) pairs by list (extra workgroups) */);
size_t pq_lut8_tiled_ws(int64_t nq, int Mpad);
int pq_lut_pass0_blocks(int64_t nq, int Mpad);      // workgroups of pass 0 of the matrix-core form (mfma = 2: they already ran elsewhere)
int launch_pq_scan8(const PQScanArgs& a, const uint8_t* lut8, const void* qparam, const int32_t* pairs_sorted,
                    const int32_t* pair_off, const int32_t* group_off, const int32_t* total_groups,
                    const int32_t* item_off, const int32_t* total_items, int nlist, int64_t max_items, int vpl,
                    hipStream_t st);
// filtered variant: instead of storing every score, append keys > tau_key[q] to cand[q][..cap) (count in cand_cnt[q])
int launch_pq_scan8_filter(const PQScanArgs& a, const uint8_t* lut8, const void* qparam, const int32_t* pairs_sorted,
                           const int32_t* pair_off, const int32_t* group_off, const int32_t* total_groups,
                           const int32_t* item_off, const int32_t* total_items, int nlist, int64_t max_items, int vpl,
                           const uint64_t* tau_key, int64_t tau_stride, uint64_t* cand, unsigned long long* cand_cnt,
                           int cand_cap, hipStream_t st);
// shared by the list-major IVF-PQ scans (k_pq.hip, k_pq_rot.hip)
struct PQQParam { float scale, bias, eps, pad; };   // pad: the largest integer table sum this query's u8 table can produce
struct PQScan8Args {
    PQScanArgs b;
    const uint8_t* lut8; const PQQParam* qp;
    const int32_t* pairs_sorted; const int32_t* pair_off; const int32_t* group_off; const int32_t* total_groups;
    const int32_t* item_off; const int32_t* total_items;
    int nlist; int max_items;
    // filtered output (FILTER = true): keys > tau_key[q] are appended to cand[q][0..cap)
    const uint64_t* tau_key; int64_t tau_stride; uint64_t* cand; unsigned long long* cand_cnt; int cand_cap;
    int prune;   // k_pq_scan_rot: skip work items none of whose queries can beat its threshold in this list (exact bound)
    const uint16_t* excl;   // k_pq_scan_rot: per query, 0x8000 | probe rank whose tile 0 the pre-pass already emitted (null: none)
    int pace;    // k_pq_scan_rot: sibling query groups of a list tile stay within `pace` loop iterations of each other (0 = free-running)
    // k_pq_rot_items (optional): qitems[(q * nprobe + probe rank) * qitems_tmax + tile] = item * 4 + slot of that (query, list tile) —
    // the inverse of the item order, which lets k_pq_gather_select find a query's survivor segments without atomics
    int32_t* qitems; int qitems_tmax;
};

// Work-item decode shared by the list-major scans.  Items are ordered (list, tile, group) so that the
// query groups of one list-tile (same codes) are adjacent; XCD c takes the contiguous item range
// [c*TI/8, (c+1)*TI/8) (workgroups are dispatched round-robin over the 8 XCDs, block b -> XCD b % 8, each
// with a private 4 MiB L2), so those groups run on ONE XCD close together in time and the tile is
// fetched from HBM once.  Placement only affects speed, never results.
__device__ inline bool pq_decode_item(const int32_t* item_off, const int32_t* group_off, int total_items, int nlist,
                                      int& l, int& gi, int& tile) {
    const int per_xcd = (total_items + 7) >> 3;
    const int ix = (int)(blockIdx.x >> 3);
    if (ix >= per_xcd) return false;
    const int item = (int)(blockIdx.x & 7) * per_xcd + ix;
    if (item >= total_items) return false;
    int lo = 0, hi = nlist;  // largest l with item_off[l] <= item
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (item_off[mid] <= item) lo = mid; else hi = mid; }
    l = lo;
    const int ng = group_off[l + 1] - group_off[l];
    const int r = item - item_off[l];
    tile = r / ng;
    gi = r - tile * ng;
    return true;
}

// rotated-layout (CB = 0) scans: k_pq_rot.hip.  lut8 is TRANSPOSED there: [nq][256][M] (see launch_pq_lut8's `transposed`)
int launch_pq_scan_rot(const PQScanArgs& a, const uint8_t* lut8t, const void* qparam, const int32_t* pairs_sorted,
                       const int32_t* pair_off, const int32_t* group_off, const int32_t* total_groups,
                       const int32_t* item_off, const int32_t* total_items, int nlist, int64_t max_items, int vpl,
                       const uint64_t* tau_key, int64_t tau_stride, uint64_t* cand, unsigned long long* cand_cnt,
                       int cand_cap, void* item_ws /* pq_scan_rot_ws(max_items, log_cap, pq_scan_rot_max_wgs(M)) bytes */, int log_cap, int prune, int pace,
                       const uint16_t* excl, int32_t* qitems /* non-null: fill it and leave the runs for k_pq_gather_select */,
                       int qitems_tmax, hipStream_t st, int q8 = 0 /* filtered M = 64: eight queries per work item (pq_scan_rot_ngq) */);
// Survivors of the filtered scan (round 4): every (persistent workgroup, wave, query slot) appends to its own LOG of log_cap keys;
// an 8-byte descriptor per (item, wave, slot) = {index of the run's first key in the log pool, keys stored | bit 31: keys were
// dropped because the log was full} lets the gather / compaction kernels find an item's runs.
int pq_scan_rot_max_wgs(int M);     // persistent workgroups of the scan on the current device
int pq_scan_rot_ngq(int M, bool filtered, int q8);   // 4-query records per work item: 1, 2 (M = 64, q8) or 4 for the filtered M = 16 scan (group the pairs by 4 x this;
                                             // size the workspace for max_items x this records and workgroups x this x 64 logs)
inline size_t pq_scan_rot_ws(int64_t max_items, int log_cap, int nwg) {   // item records + run descriptors + logs + per-XCD counters + progress words
    return (size_t)(max_items + 8) * (176 + 512 + 4) + (size_t)nwg * 64 * (size_t)log_cap * 8 + 1024;
}
inline uint2* pq_scan_rot_ws_desc(void* ws, int64_t max_items) { return reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(ws) + (size_t)(max_items + 8) * 176); }
inline uint64_t* pq_scan_rot_ws_keys(void* ws, int64_t max_items) { return reinterpret_cast<uint64_t*>(reinterpret_cast<uint8_t*>(ws) + (size_t)(max_items + 8) * (176 + 512)); }
inline uint32_t* pq_scan_rot_ws_ctr(void* ws, int64_t max_items, int log_cap, int nwg) {
    return reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(pq_scan_rot_ws_keys(ws, max_items)) + (size_t)nwg * 64 * (size_t)log_cap * 8);
}
// Candidate gather + selection in one launch (round 3; replaces k_pq_rot_compact + the candidate merge when nprobe x tiles is small):
// one workgroup per query walks the query's own survivor runs (through qitems), lays their keys end to end in the query's
// candidate row behind what the pre-pass emitted, sets the row count (past the capacity when a log overflowed, as the
// compaction does) and writes the K' largest keys to the state row — no atomics, no second pass over the row.
struct PQGatherArgs {
    const int32_t* probe_list; const int64_t* list_len; int nprobe; int tile_rows; int tmax;
    const int32_t* qitems; const uint2* seg_desc; const uint64_t* log_keys;
    uint64_t* cand; unsigned long long* cand_cnt; int cand_cap;
    uint64_t* state; int KP;
};
bool pq_gather_select_applies(int nprobe, int tmax, int KP);
void launch_pq_gather_select(const PQGatherArgs& a, int64_t nq, hipStream_t st);
// exact per-(query, list) scan of the rotated layout (fp32 table, sequential sums = oracle bits): fallback / A-B path
int launch_pq_scan_rot_exact(const PQScanArgs& a, hipStream_t st);
// codes for rows of a batch: residual vs centroid[assign] (centroids may be null -> no residual).
// plain_out != null: write [n, Mpad] row-major instead of the slab layout (training / export).
void launch_pq_encode(const void* x, int x_f16, int64_t n, int ldx, int d, int M, int Mpad, int CB,
                      const float* centroids, const int32_t* assign, const float* codebooks,
                      const int64_t* dest_row, uint8_t* codes, uint8_t* plain_out, hipStream_t st);

// k_select.hip
struct SelectArgs {
    const void* in; int in_is_keys;         // float scores (idx = idx_base + column) or u64 keys
    int64_t row_stride;                     // elements between rows
    const int64_t* row_n; int64_t row_n_stride; int64_t n_uniform;  // valid elements per row
    int64_t seg_len; int nseg;
    uint32_t idx_base;
    const uint64_t* init;                   // optional [nrows, KP] running state merged in (nseg must be 1)
    const uint64_t* tau_ptr; int64_t tau_stride;  // optional per-row starting threshold (a key known to be
                                                  // <= the row's final k-th best): keys <= it are dropped
    int seg_base;                           // first segment index handled by this launch
    uint64_t* out; int64_t out_row_stride;  // out[row*out_row_stride + seg*KP + i]
    int64_t nrows; int KP; int BUF; int k;
    // threshold pre-pass form (nseg == 1): write only the row's KP-th key (the other KP-1 slots zero) and reset
    // the row's candidate counter, so no extra launches sit between the pre-pass and the filtered scan
    int keep_last; unsigned long long* zero_cnt;
    const int32_t* row_filter;              // optional (radix form): only rows with row_filter[row] == 1 are processed
};
void launch_select(const SelectArgs& a, hipStream_t st);
// IVF-PQ threshold pre-pass in one launch (k_pq_prepass): needs the 16-byte-granule code layout
struct PQPrepassArgs {
    const uint8_t* codes; const int64_t* list_base; const int64_t* list_len;
    const int32_t* probe_list; const float* probe_dis0; const int64_t* seg_start;
    const uint8_t* lut8; const float* qparam;   // [nq][Mpad][256] u8 tables; [nq] {scale, bias, eps, pad}
    int nprobe; int Mpad; int pre_rows; int KP; int CB;   // CB = 0: rotated layout, lut8 transposed
    int k;                                                 // the search's k: the threshold is derived from the sample's k-th best score
    uint64_t* state; unsigned long long* cand_cnt;
    uint64_t* tau;                                         // [nq] out: threshold key (score a_k - 2 eps, low word 0), 0 = none
    // emission (null cand = off): the sample's keys above the threshold go straight to cand[q][..] when the sample covers the
    // first scan tile (tile_rows) of the list; excl[q] = 0x8000 | probe rank tells the scan to drop that (query, list, tile 0)
    uint64_t* cand; int cand_cap; int tile_rows; uint16_t* excl;
};
void launch_pq_prepass(const PQPrepassArgs& a, int64_t nq, hipStream_t st);
// the same threshold from four queries per workgroup on the scan's table format (k_pq_rot.hip: k_pq_prepass4; no emission);
// returns -1 when it does not apply (layout, M, or a sample larger than pq_prepass4_max_rows(M))
int pq_prepass4_max_rows(int M);
int launch_pq_prepass4(const PQPrepassArgs& a, int64_t nq, hipStream_t st);
int launch_pq_prepass4_big(const PQPrepassArgs& a, int64_t nq, hipStream_t st);   // large samples (<= 32768 rows, several lists): histogram form
// Pass 0 of the matrix-core table build (k_pq.hip: k_pq_lut_mfma) as a device function: the (min, max) pairs of the table entries of one
// (32-query tile, 4 sub-quantisers) block — it needs the queries only, so it rides as extra workgroups of the probe-pick launch
// (k_coarse_pick: a latency chain that leaves the CUs' issue slots idle) instead of a launch of its own.  256 threads; bid = linear
// block number in the table build's XCD-aware order.
struct LutPass0Args { const float* Q32; int ldq; const float* codebooks; int M, Mpad; int64_t nq; float* mnmx; int nblocks; };
#define LM_Q 32
#define LM_MB 4
#ifdef __HIPCC__
typedef float lm_f16 __attribute__((ext_vector_type(16)));
// operands of the wave's sub-quantiser m for the tile's 32 queries: dims 2 kk + h of the lane's query / codeword (K step kk: lanes 0-31 feed
// dim 2 kk, lanes 32-63 dim 2 kk + 1 — chain order)
__device__ inline void lut_mfma_operands(const float* Q32, int ldq, const float* codebooks, int64_t q, int m, bool mreal, bool qok, int j, int hh,
                                         float (&bq)[4], float (&ac)[8][4]) {
    bq[0] = bq[1] = bq[2] = bq[3] = 0.f;
    if (mreal && qok) {
        const float4* p = reinterpret_cast<const float4*>(Q32 + q * ldq + m * 8);
        const float4 x = p[0], y = p[1];
        bq[0] = hh ? x.y : x.x; bq[1] = hh ? x.w : x.z; bq[2] = hh ? y.y : y.x; bq[3] = hh ? y.w : y.z;
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
        ac[t][0] = ac[t][1] = ac[t][2] = ac[t][3] = 0.f;
        if (mreal) {
            const float4* p = reinterpret_cast<const float4*>(codebooks + ((int64_t)m * 256 + 32 * t + j) * 8);
            const float4 x = p[0], y = p[1];
            ac[t][0] = hh ? x.y : x.x; ac[t][1] = hh ? x.w : x.z; ac[t][2] = hh ? y.y : y.x; ac[t][3] = hh ? y.w : y.z;
        }
    }
}
// (dep: a value of the previous tile's reductions — through an opaque asm it becomes this tile's accumulator zero, so the tiles are computed
//  one after the other: left alone the compiler computes all 8 first and spills 128 accumulators)
__device__ __forceinline__ lm_f16 lut_mfma_tile(const float (&ac)[8][4], const float (&bq)[4], int t, float dep) {
    lm_f16 acc;
    float z = 0.0f;
    asm volatile("" : "+v"(z) : "v"(dep));
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = z;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t][kk], bq[kk], acc, 0, 0, 0);
    return acc;
}
__device__ inline void pq_lut_pass0_block(const LutPass0Args& a, unsigned bid) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, hh = lane >> 5;
    const int nmb = (a.Mpad + LM_MB - 1) / LM_MB, nqt = (int)((a.nq + LM_Q - 1) / LM_Q);
    const int rest = (int)(bid >> 3);
    const int qt = (rest / nmb) * 8 + (int)(bid & 7);
    if (qt >= nqt) return;
    const int m = (rest % nmb) * LM_MB + w;
    const int64_t q0 = (int64_t)qt * LM_Q, q = q0 + j;
    const int nqc = (int)((a.nq - q0) < LM_Q ? (a.nq - q0) : LM_Q);
    const bool mreal = m < a.M, mpad = m < a.Mpad, qok = j < nqc;
    float bq[4], ac[8][4];
    lut_mfma_operands(a.Q32, a.ldq, a.codebooks, q, m, mreal, qok, j, hh, bq, ac);
    float mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const lm_f16 v = lut_mfma_tile(ac, bq, t, mn);
#pragma unroll
        for (int r = 0; r < 16; r++) { mn = fminf(mn, v[r]); mx = fmaxf(mx, v[r]); }
    }
    mn = fminf(mn, __shfl_xor(mn, 32)); mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (hh == 0 && qok && mpad) *reinterpret_cast<float2*>(a.mnmx + (q * a.Mpad + m) * 2) = make_float2(mn, mx);
}
#endif
// fast coarse quantiser (k_gemm.hip: launch_coarse_approx; k_select.hip: k_coarse_pick)
#define CP_CMAX 64             // most candidates a query may have (its nprobe best by approximate score + everything within 2 e of the last)
#define CP_CH 128              // dimensions per staged chunk of the candidates' centroid rows
#define CP_RS (CP_CH + 4)      // ... whose LDS row stride keeps ds_read_b128 of 16 consecutive lanes on 64 different banks
#define CP_MAXPROBE 48         // (more probes = more candidate rows than two chunks in flight fit in 128 registers; the exact GEMM serves them)
#define CP_MAXLIST 16384       // the row's keys live in LDS
struct CoarsePickArgs {
    const float* approx; int64_t nlp; int nlist;      // [nq][nlp] approximate scores
    const float* Q32; int ld; int d; const float* C;  // exact operands: queries [nq][ld], centroids [nlist][d]
    float cmax;                                       // >= the largest centroid L2 norm
    float ef;                                         // relative rounding of the approximate operands: 2^-11 per fp16-rounded side
    int nprobe; const int64_t* list_len; int pad_to;
    int32_t* probe_list; float* dis0; int64_t* seg_start; int32_t* bad;
    int cmax_rt;                                      // set by the launcher: candidate rows of this launch (coarse_pick_cmax)
    int64_t nq;                                       // set by the launcher
    LutPass0Args lp0;                                 // nblocks > 0: that many extra workgroups run pass 0 of the IVF-PQ table build (grid = nq + nblocks)
};
size_t coarse_pick_lds(int nlist, int d, int nprobe);
void launch_coarse_approx(const __half* Q16, int64_t nq_pad, const __half* C16, int nlist, int ld, float* S, int64_t lds_, hipStream_t st,
                          float* Q32 = nullptr, int64_t widen_n = 0);
void launch_coarse_pick(const CoarsePickArgs& a, int64_t nq, hipStream_t st);
void launch_probe_setup(const uint64_t* probe_keys, int KPp, int64_t nq, int nprobe, const int64_t* list_len,
                        int pad_to, int32_t* probe_list, float* probe_dis0, int64_t* seg_start, hipStream_t st);
// tile_rows > 0: also build the (list, tile, group) work-item table: item_off[nlist+1], total_items
void launch_group_pairs(const int32_t* probe_list, int64_t npairs, int nlist, int group_size, int32_t* cnt,
                        int32_t* cursor, int32_t* pair_off, int32_t* group_off, int32_t* total_groups,
                        int32_t* pairs_sorted, const int64_t* list_len, int tile_rows, int32_t* item_off,
                        int32_t* total_items, int nprobe, int jmin, int jmax, int tile_cap, hipStream_t st);
// The same grouping as EXTRA WORKGROUPS of another launch (round 6): the four launches above are ~24 us of launch floors and round trips in
// front of the scan for a few microseconds of work.  Blocks 0 .. nb - 1 of the host kernel (k_pq_lut_mfma<1>: it follows the probe
// selection in stream order and has LDS to spare) each own nlist / nb consecutive lists: histogram of their lists over ALL pairs in LDS,
// local offsets, a chained hand-over of the block totals (block b waits for blocks < b: they are dispatched before it, so the wait is
// safe), then the scatter.  flags: [nb][4] words = {pairs, groups, items, epoch}: a block's totals are valid once its epoch word equals
// the launch's epoch (the host increments it per launch: no reset pass).  Full probe-rank range only (jmin = 0, jmax = nprobe).
struct PairGroupArgs {
    const int32_t* probe_list; const int64_t* list_len;
    int32_t *pair_off, *group_off, *item_off, *total_groups, *total_items, *pairs_sorted;
    uint32_t* flags;
    int npairs, nlist, G, tile_rows, tile_cap, nb;
    uint32_t epoch;
};
#define PG_BLOCKS 32
#define PG_MAX_LPB 2048                 // lists per block the LDS histogram holds (nlist <= 65536 with 32 blocks)
#define PG_MAX_PAIRS 131072             // every block reads all pairs twice
#ifdef __HIPCC__
__device__ inline void group_pairs_block(const PairGroupArgs& g, int b, int32_t* hist /* LDS: PG_MAX_LPB + 16 ints */) {
    const int t = (int)threadIdx.x, nt = (int)blockDim.x, lane = t & 63, w = t >> 6, nw = nt >> 6;
    int32_t* sc = hist + PG_MAX_LPB;           // [12] cross-wave scratch
    __builtin_amdgcn_s_setprio(3);             // a latency chain beside the host kernel's throughput work on the same CUs: first pick of the issue slots
    const int lpb = (g.nlist + g.nb - 1) / g.nb, lo = b * lpb;
    int hi = lo + lpb; if (hi > g.nlist) hi = g.nlist;
    const int nl = hi > lo ? hi - lo : 0;
    for (int i = t; i < nl; i += nt) hist[i] = 0;
    __syncthreads();
    // every block reads ALL pairs, twice: 16-byte loads, sixteen in flight per thread (one load per loop turn was 128 exposed L2 round trips)
    const int np4 = ((reinterpret_cast<uintptr_t>(g.probe_list) & 15) == 0) ? g.npairs >> 2 : 0;
    const int4* pl4 = reinterpret_cast<const int4*>(g.probe_list);
    auto for_pairs = [&](auto&& f) {
        for (int i0 = t; i0 < np4; i0 += 16 * nt) {
            int4 v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) { const int i = i0 + u * nt; v[u] = i < np4 ? pl4[i] : make_int4(-1, -1, -1, -1); }
#pragma unroll
            for (int u = 0; u < 16; u++) { const int i = 4 * (i0 + u * nt); f(v[u].x, i); f(v[u].y, i + 1); f(v[u].z, i + 2); f(v[u].w, i + 3); }
        }
        for (int i = 4 * np4 + t; i < g.npairs; i += nt) f(g.probe_list[i], i);
    };
    for_pairs([&](int32_t l, int) { if (l >= lo && l < hi) atomicAdd(&hist[l - lo], 1); });
    __syncthreads();
    auto ntiles = [&](int l) {
        int32_t n = (int32_t)((g.list_len[l] + g.tile_rows - 1) / g.tile_rows);
        return (g.tile_cap > 0 && n > g.tile_cap) ? g.tile_cap : n;
    };
    const int per = (lpb + nt - 1) / nt, l0 = t * per;
    int l1 = l0 + per; if (l1 > nl) l1 = nl;
    int32_t ap = 0, ag = 0, ai = 0;
    for (int i = l0; i < l1; i++) { const int c = hist[i], ng = (c + g.G - 1) / g.G; ap += c; ag += ng; if (g.tile_rows > 0 && ng) ai += ng * ntiles(lo + i); }
    // exclusive scan of (ap, ag, ai) over the block
    int32_t ia = ap, ib = ag, ic = ai;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t ya = __shfl_up(ia, off), yb = __shfl_up(ib, off), yc = __shfl_up(ic, off);
        if (lane >= off) { ia += ya; ib += yb; ic += yc; }
    }
    if (lane == 63) { sc[w] = ia; sc[4 + w] = ib; sc[8 + w] = ic; }
    __syncthreads();
    int32_t wa = 0, wb = 0, wc = 0, ta = 0, tb = 0, tc = 0;
    for (int k = 0; k < nw && k < 4; k++) { if (k < w) { wa += sc[k]; wb += sc[4 + k]; wc += sc[8 + k]; } ta += sc[k]; tb += sc[4 + k]; tc += sc[8 + k]; }
    ap = ia - ap + wa; ag = ib - ag + wb; ai = ic - ai + wc;          // exclusive, block-local
    // publish the block totals, then collect the predecessors'
    if (t == 0) {
        __hip_atomic_store(&g.flags[4 * b], (uint32_t)ta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&g.flags[4 * b + 1], (uint32_t)tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&g.flags[4 * b + 2], (uint32_t)tc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        __hip_atomic_store(&g.flags[4 * b + 3], g.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int32_t pa = 0, pb = 0, pc = 0;
    if (t < b) {
        while (__hip_atomic_load(&g.flags[4 * t + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.epoch) __builtin_amdgcn_s_sleep(1);
        pa = (int32_t)__hip_atomic_load(&g.flags[4 * t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pb = (int32_t)__hip_atomic_load(&g.flags[4 * t + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pc = (int32_t)__hip_atomic_load(&g.flags[4 * t + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                 // sc is free again
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { pa += __shfl_xor(pa, off); pb += __shfl_xor(pb, off); pc += __shfl_xor(pc, off); }
    if (lane == 0) { sc[w] = pa; sc[4 + w] = pb; sc[8 + w] = pc; }
    __syncthreads();
    int32_t ba = 0, bb = 0, bc = 0;
    for (int k = 0; k < nw && k < 4; k++) { ba += sc[k]; bb += sc[4 + k]; bc += sc[8 + k]; }
    if (b == g.nb - 1 && t == 0) {
        g.pair_off[g.nlist] = ba + ta; g.group_off[g.nlist] = bb + tb; *g.total_groups = bb + tb;
        if (g.tile_rows > 0) { g.item_off[g.nlist] = bc + tc; *g.total_items = bc + tc; }
    }
    ap += ba; ag += bb; ai += bc;
    for (int i = l0; i < l1; i++) {
        const int c = hist[i], ng = (c + g.G - 1) / g.G;
        g.pair_off[lo + i] = ap; g.group_off[lo + i] = ag;
        if (g.tile_rows > 0) { g.item_off[lo + i] = ai; if (ng) ai += ng * ntiles(lo + i); }
        hist[i] = ap;                // from here on: the list's write cursor
        ap += c; ag += ng;
    }
    __syncthreads();
    for_pairs([&](int32_t l, int i) { if (l >= lo && l < hi) g.pairs_sorted[atomicAdd(&hist[l - lo], 1)] = i; });
}
#endif
struct FinalizeArgs {
    int kind; int metric;
    const uint64_t* state; int KP; int k; int64_t nq;
    int KPv;                                // candidates CONSIDERED (0 = KP): the state rows are sorted by approximate key, so the first KPv are the
                                            // KPv best; the rest is ignored and the certificate runs against the KPv-th.  Flat / IVF-Flat pass
                                            // k + max(8, k / 16) instead of its power of two (k = 1000: 1062 rows of 1.5 KB re-read, not 2048)
    // location resolution
    const int32_t* probe_list; const int64_t* seg_start; int nprobe;
    const int64_t* list_base;
    const int64_t* ids;                     // per storage row, or null (id = row)
    // exact rerank (Flat / IVF-Flat)
    const float* Q32; int ldq; int d;
    const void* X; int x_f16; int ld;
    // exact re-score + certificate (IVF-PQ fast scan): candidates carry APPROXIMATE scores
    int pq_rescore; const uint8_t* codes; int M; int Mpad; int CB;
    int plain_stride;              // row stride of codes_plain in bytes (pq_plain_stride(M); 0 = M)
    const uint8_t* codes_plain;    // optional (large K'): a row-major copy of the codes, [storage row][M] — a candidate's M bytes are then two or
                                   // three 64-byte sectors instead of one per 16-byte piece of the scan layouts (launch_pq_plain_rows)
    const float* lut32;            // fp32 tables, or null: entries recomputed from Q32 and `codebooks`
    const float* codebooks; int dsub;
    const float* probe_dis0; const void* qparam; int32_t* uncertain;
    const unsigned long long* cand_cnt; int cand_cap;   // filtered scan: per-query candidate counts / capacity
    int cand_cnt_n; int64_t cand_cnt_stride;            // Flat, staged filter: counter sets (one per stage; 0 = 1) and the words between them — k_finalize flags a
                                                        // query any of whose stage rows overflowed
    // Flat / IVF-Flat certificate (uncertain != null): the scan ordered candidates by an APPROXIMATE score (fp16 MFMA, fp32
    // accumulation; queries or fp32 rows possibly rounded to fp16).  |approx - exact| <= cert_rel * |q| * cert_xmax + cert_abs
    // for every stored vector; a query whose K'-th approximate candidate could still beat its exact k-th is flagged.
    float cert_rel, cert_xmax, cert_abs;
    const int* cert_qflag; float cert_rel_qlossy;   // *cert_qflag != 0: the batch held an fp32 query value fp16 cannot represent
    float* D; int64_t* I;
    const int32_t* row_filter;     // optional: only queries with row_filter[q] == 1 are processed (second-chance pass)
    int no_cert;                   // the state keys are the best of a COMPLETE, exactly scored candidate row: nothing to certify
    unsigned long long* stat;      // profile >= 2 (else null): [0] += candidates k_pq_final_tab re-scored exactly (after its 2 eps cut)
    int probe_lds_off;             // set by launch_finalize: byte offset of the query's probe table in the dynamic LDS (0 = keep it in global memory)
    int par_entries;               // set by launch_finalize: parallel table-entry form of the IVF-PQ re-score (small batches)
    int rank_sort;                 // set by launch_finalize: order the candidates by counting (K' <= 1024) instead of a bitonic network
};
void launch_finalize(const FinalizeArgs& a, hipStream_t st);
// out[row * M + m] = the code of (storage row, sub-quantiser m) for every row < nrows of a rotated / sliced layout (M % 16 == 0, M >= 32)
int pq_plain_stride(int M);        // bytes between the rows of that copy (a power of two / a multiple of 128: no row straddles a 128-byte line)
void launch_pq_plain_rows(const uint8_t* codes, int64_t nrows, int M, int CB, uint8_t* out, hipStream_t st);
// exact scores for every candidate key of the queries with a.row_filter[q] == 1, in place (a.cand_cnt gives the row lengths)
void launch_pq_rescore_all(const FinalizeArgs& a, uint64_t* cand, int cand_cap, hipStream_t st);
// IVF-PQ finalize from the complete candidate row (k_select.hip: k_pq_final_tab): exact re-score of EVERY key of the row with the
// query's fp32 table in LDS, top k by (score, id) — no K', no certificate.  capacity 0: does not apply.  a.cand_cnt / a.uncertain
// / a.row_filter as in FinalizeArgs (row_filter: second chance of the K' path).
int pq_final_tab_capacity(int M, int CB, int k);
void launch_pq_final_tab(const FinalizeArgs& a, uint64_t* cand, int cand_cap, uint64_t* tie_ws /* [nq, cand_cap]: ids of score ties */, hipStream_t st);
// Exact scores of the uncertified queries (fallback of the certificate above): fp64 dot product of the query with EVERY row
// the query may see (Flat: all rows; IVF-Flat: the rows of its probed lists, laid out by seg_start), rounded once to fp32 —
// the canonical score — into temp[q * tstride + column]; invalid columns get -inf.
struct ExactScoreArgs {
    int kind; int metric; int64_t nq;
    const float* Q32; int ldq; int d;
    const void* X; int x_f16; int ld; int64_t flat_n;
    const int32_t* probe_list; const int64_t* seg_start; int nprobe; const int64_t* list_base; const int64_t* list_len;
    float* temp; int64_t tstride;
};
void launch_exact_scores(const ExactScoreArgs& a, hipStream_t st);
// any nshards x k with k <= 8192 (rounds of groups when nshards * k > 16384; stream-ordered round buffers); false: k too large or
// no memory for the round buffers
bool launch_merge_topk(int nshards, int64_t nq, int k, int metric, const float* D, const int64_t* I,
                       float* Do, int64_t* Io, hipStream_t st);
// single-index order (score desc, id asc) for the shards of ONE logical index (rsx_sharded_create); nshards * k <= 8192
void launch_merge_topk_byid(int nshards, int64_t nq, int k, int metric, const float* D, const int64_t* I, float* Do,
                            int64_t* Io, hipStream_t st);
bool launch_merge_packed(int nshards, int64_t nq, int k, int metric, const int64_t* packed /* [nshards,2,nq,k] */, float* Do,
                         int64_t* Io, hipStream_t st);
void launch_pack_topk(int64_t n, const float* D, const int64_t* I, int64_t id_offset, int64_t* out /* [2,n] */, hipStream_t st);

}  // namespace rsx
