// rsx_host.h — host side of librsx shared by the api_*.hip files: error plumbing, device / pinned buffers, the index object
// (struct rsx_index = the opaque rsx_index_t of include/rsx.h) and the prototypes of the host functions that cross files.
//   api_build.hip    HBM layout management, add / train, list import / export, persistence
//   api_search.hip   the search drivers (search_batch = one internal batch, search_impl = one rsx_search call)
//   api_sharded.hip  the single-process multi-GPU handle
//   rsx_api.hip      the extern "C" entry points
// Host control plane only; all search arithmetic is in the k_*.hip kernels.  There is no CPU search path.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/rsx.h"
#include "rsx_internal.h"

using namespace rsx;

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
extern thread_local std::string g_err;      // rsx_api.hip

struct RsxError : std::runtime_error {
    int code;
    RsxError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
#define RSX_THROW(code, ...)                                  \
    do {                                                      \
        char _b[512]; snprintf(_b, sizeof(_b), __VA_ARGS__);  \
        throw RsxError(code, _b);                             \
    } while (0)
#define HIPCHECK(expr)                                                                              \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            int code = (_e == hipErrorOutOfMemory) ? RSX_ERR_OOM : RSX_ERR_HIP;                     \
            RSX_THROW(code, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
        }                                                                                           \
    } while (0)

template <typename F>
static inline int guarded(F&& f) {
    try {
        f();
        return RSX_OK;
    } catch (const RsxError& e) {
        g_err = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        g_err = "host allocation failed";
        return RSX_ERR_OOM;
    } catch (const std::exception& e) {
        g_err = e.what();
        return RSX_ERR_INVALID;
    }
}

// ---------------------------------------------------------------------------------------
// device buffers
// ---------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool host_mapped = false;     // page-locked HOST memory mapped into the device (the per-query flag words: kernels write them across the
                                  // bus, the host reads them after the stream's synchronisation — no copy command at the end of a batch)
    void ensure(size_t n) {
        if (n <= bytes) return;
        release();
        size_t want = n + n / 8;
        if (host_mapped) { HIPCHECK(hipHostMalloc(&p, want, hipHostMallocMapped)); bytes = want; return; }
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            HIPCHECK(hipMalloc(&p, n));
            want = n;
        }
        bytes = want;
    }
    void release() {
        if (p) (void)(host_mapped ? hipHostFree(p) : hipFree(p));
        p = nullptr; bytes = 0;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// Page-locked host staging for the small transfers of the latency path (a single query in, k results and the
// certificate flags out): copies to and from pageable memory go through the runtime's own staging and block the host.
struct PinBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool ensure(size_t n) {          // false: no pinned memory to be had — the caller keeps the pageable path
        if (n <= bytes) return true;
        release();
        if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
        bytes = n;
        return true;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
    ~PinBuf() { release(); }
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
};

static inline bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

static inline int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static inline int pow2ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// ---------------------------------------------------------------------------------------
// deterministic host RNG shared with the training spec (splitmix64 Fisher-Yates)
// ---------------------------------------------------------------------------------------
static inline uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline void rand_perm(int64_t n, uint64_t seed, std::vector<int64_t>& perm) {
    perm.resize((size_t)n);
    uint64_t s = seed;
    for (int64_t i = 0; i < n; i++) perm[(size_t)i] = i;
    for (int64_t i = 0; i + 1 < n; i++) {
        int64_t j = i + (int64_t)(splitmix(s) % (uint64_t)(n - i));
        std::swap(perm[(size_t)i], perm[(size_t)j]);
    }
}

// ---------------------------------------------------------------------------------------
// the index object
// ---------------------------------------------------------------------------------------
struct rsx_index {
    int kind = 0, d = 0, metric = 0, device = 0;
    int nlist = 1, M = 0, nbits = 8, Mpad = 0, CB = 16, dsub = 0;   // CB: code layout (rsx_internal.h), 0 = rotated
    int CB_granule = 16;
#ifndef PQ_LAYOUT_DEFAULT
#define PQ_LAYOUT_DEFAULT 2      // RSX_PQ_LAYOUT unset: 1 = rotated wherever it applies, 2 = additionally the sliced layout for M = 96
#endif
    int nprobe = 1;
    bool trained = false;
    int64_t ntotal = 0;
    hipStream_t st = nullptr;
    // side stream of a search batch (round 4): stages that do not depend on each other run beside the main chain — the table build
    // beside the coarse quantiser + probe selection, the pair grouping beside the threshold pre-pass
    hipStream_t st2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_probe = nullptr, ev_lut = nullptr, ev_group = nullptr;
    int overlap = 0;          // 1 = IVF-PQ table build and pair grouping on a side stream beside the coarse quantiser / the pre-pass (rounds 4-5; round 6:
                              // three cross-stream joins of ~12 us each cost what the overlap buys — one stream, fewer and fused launches instead)
    ~rsx_index() {
        if (st2) { (void)hipSetDevice(device); (void)hipStreamDestroy(st2); }
        for (hipEvent_t e : {ev_fork, ev_probe, ev_lut, ev_group}) if (e) (void)hipEventDestroy(e);
    }
    PinBuf pin_q, pin_out, pin_flags;     // latency path: pinned staging of queries / results / certificate flags

    // trained parameters
    std::vector<float> h_centroids, h_codebooks;
    DevBuf d_centroids, d_codebooks;
    DevBuf cent16;            // fp16 copy of the centroids, [round_up(nlist, 128)][ld] zero padded: the fast coarse quantiser's operand (built on demand)
    uint64_t cent_gen = 0, cent16_gen = 0;   // set_centroids counts; the copy belongs to cent16_gen
    float cent_cmax = 0.0f;   // >= the largest centroid norm (with the copy)
    int coarse_fast = 1;      // coarse quantiser: fp16 MFMA scores + exact re-score of the candidates (k_coarse_pick), 0 = the exact GEMM for all lists
    bool coarse_flags_live = false;   // this batch's k_coarse_pick wrote per-query flags behind w_uncertain's

    // storage.  PQ: slab layout bytes.  Flat/IVFFlat: rows of ld elements (fp16 or fp32).
    int ld = 0;               // row stride (elements) of flat rows: d rounded up to 64
    int storage_f16 = 1;
    bool storage_decided = false;
    bool custom_ids = false;  // Flat: ids array only when the caller supplied ids
    DevBuf data, ids, norms;
    DevBuf codes_plain;               // IVF-PQ, block layouts: row-major copy of the codes for large-K' finalizes, built on demand (api_search.hip)
    uint64_t plain_gen = ~0ull; const void* plain_of = nullptr;
    std::vector<int64_t> h_base, h_len, h_cap;
    DevBuf d_base, d_len;
    int64_t total_cap = 0;

    // knobs
    int query_batch = 1024;
    int scan_chunk = 0;
    int scan_kernel = 0;  // 0 = auto (list-major v2 when the layout allows), 1 = force the per-pair v1 kernel
    int pq_fast = 1;      // IVFPQ: 8-bit-table fast scan + certified exact re-rank (results identical to exact)
    int pq_fast_kp = 0;   // candidates kept by the fast scan (0 = auto); tests shrink it to force fallbacks
    int pq_filter = 1;    // fast scan: filter candidates inside the scan kernel (0 = full score buffer + select)
    int pq_pace = 128 | (4 << 12);   // rotated fast scan schedule (never changes a result): bit 7 = a query group that starts while a
                          // sibling group of its list tile is under way JOINS it at its position and wraps around (L2 reuse of the code
                          // lines), bits 8-11 = join offset in tile rows (0 = 2), bits 12-15 = the last n rows of a tile are handed to
                          // the waves dynamically (default 4; 0 = static columns), bit 4 = every chunk dynamic, bit 6 = no issue-
                          // priority rotation
    int pq_q8 = 1;        // eight queries per table gather (8-byte entries, ds_read_b64): rotated layout M = 64 (k_pq_scan_rot<..., 2>; 0 = the 4-query form);
                          // sliced layout (M = 96): 1 = k_pq_scan_sl8 for batches with >= 3 probing queries per list on lists of >= 4096 vectors on average,
                          // else the four-query single-pass k_pq_scan_sl4; 2 = always eight, 0 = always four
    int pq_prune = 0;     // rotated fast scan: skip (list, query group) items that cannot hold a survivor (exact bound; opt-in)
    int pq_lut_early = 1;     // pass 0 of the matrix-core table build as extra workgroups of the probe-pick launch (0 = its own launch)
    int pq_group_fused = 1;   // the (query, probe) pairs grouped by list by extra workgroups of the table launch (0 = its own four launches)
    uint32_t pg_epoch = 0;    // ... whose hand-over words carry this launch counter
    int lut_tiled = 2;    // fast scan tables (dsub 8): tiled build sharing codebook slices across queries: 2 = one launch (k_pq_lut_once), 1 = two passes
                          // + parameters in three launches, 0 = one workgroup per query
    int pq_prepass_fused = 1;   // filtered fast scan: threshold pre-pass in one launch (0 = grouping + scan + selection)
    int ivf_qtiles = 1;      // IVF-Flat LDS-DMA scans: 1 = choose 16 / 64 / 128 probing queries per group from the queries per list, 2 / 4 / 8 = force 32 / 64 / 128, 0 = always 16
    int pq_prepass4 = 1;     // rotated fast scan, full batches: threshold pre-pass with four queries per workgroup on the scan's table format (1 = small and large k, 2 = small k only, 0 = never)
    int pq_plain_codes = 1;  // block layouts, K' >= 256: finalize from a row-major copy of the codes (built on demand, M bytes per vector)
    int pq_gather = 1;       // rotated fast scan: candidate gather + selection in one launch (k_pq_gather_select) instead of compaction + merge
    int pq_final_tab = 1;    // rotated fast scan: finalize from the complete candidate row with the fp32 table in LDS (1 = when K' >= 512 or dsub > 8 and as the second chance, 2 = always, 0 = never)
    int pq_log_cap = 0;      // rotated fast scan: keys per survivor log (0 = from the pool budget); tests shrink it to force the overflow path
    int pq_pre_mult = 80, pq_pre_max = 16384;    // ... and for larger k: pq_pre_mult x k rows, at most pq_pre_max (<= 32768: 64 KiB of 16-bit sums in LDS; 16384 measured best overall on the headline index at k = 100 / 1000 / 2000, profiles/r04u_pre_sweep.jsonl; round 6, after the four-query histogram pre-pass: 80 x k — k = 100 2.95 -> 2.88 ms per batch, k = 50 / 200 / 1000 unchanged, profiles/r06_fixed_cost.md 5)
    int pq_pre_rows = 4096;  // filtered fast scan: vectors of each query's closest list the threshold pre-pass scores (0 = one scan tile)
    int add_list_mod = 1, add_list_rem = 0;   // IVF add keeps only lists l with l % mod == rem (list-sharded multi-GPU index)
    int64_t ndropped = 0;                     // vectors seen by add but owned by other shards
    int flat_filter = 1;  // Flat: filtered GEMM launches after the threshold phase (0 = score buffer per chunk)
    int flat_pre_mult = 16;  // Flat: rows of the threshold phase per K' (through the score buffer), rounded up to units of flat_pre_unit rows
    int ivf_overflow_max = 0; // IVF-Flat filtered scan: overflowed candidate rows whose queries are re-run exactly on their own before the batch is rescanned (0 = max(2, nq / 128))
    int flat_pre_unit = 0;   // Flat: rows per unit of the threshold phase and the stage boundaries (0 = 16384 for K' <= 64, else 32768)
    int flat_stages = 0;     // Flat: filtered stages behind the threshold phase (0 = from K' and the row count; see search_batch)
    int ivf_filter = 1;   // IVF-Flat: candidates filtered inside the list scan (0 = full score rows + select)
    int ivf_pre_lists = 0;   // IVF-Flat threshold sample at large K': closest lists sampled (0 = 2)
    int ivf_pre_mult = 4;    // ... and rows of each per K'
    int profile = 0;
    int64_t temp_budget = (int64_t)16 << 30;

    // workspace
    DevBuf w_q32, w_q16, w_coarse, w_keys1, w_probekeys, w_probelist, w_dis0, w_segstart, w_temp, w_lut, w_lutws, w_pgflags, w_state,
        w_D, w_I, w_qin, w_pairs, w_flag, w_x, w_partial, w_assign, w_dest, w_idsin, w_misc, w_lut8, w_qparam, w_uncertain, w_fbq, w_fbD, w_fbI, w_cand, w_candcnt, w_itemdesc, w_tau, w_excl, w_state2, w_addcnt, w_addstart, w_qitems, w_tiews;
    std::map<std::string, double> timing;

    // Flat / IVF-Flat: largest |x|^2 ever added (certificate of the MFMA scan); device copy is the running atomic max
    float max_norm2 = 0.0f;
    DevBuf d_maxnorm;
    int flat_cert = 1;        // 1 = certify the fp16-MFMA scan and re-run uncertified queries exactly (0 = round-1 behaviour)

    // host-side bounds that only depend on the list lengths (top-nprobe sums of list / tile counts): computed once per
    // directory generation instead of a partial_sort over nlist two to four times per search batch
    uint64_t dir_gen = 0;
    std::map<std::tuple<int, int, int>, std::pair<uint64_t, std::pair<int64_t, int64_t>>> bound_cache;

    // two-call search (rsx_search_prepass / rsx_search_scan: the caller exchanges the per-query thresholds between the calls,
    // e.g. an all-reduce(MAX) across the ranks of a LIST-sharded index).  The search runs on a worker thread that parks right
    // after the threshold pre-pass (only the worker thread ever parks: TwoCall::worker); parked / go / done / tau are guarded by
    // TwoCall::mu, and every other entry point refuses the handle while `active` (refuse_while_two_call).
    struct TwoCall {
        std::thread th;
        std::thread::id worker;     // the thread running the parked search: the only one allowed to park
        std::mutex mu;
        std::condition_variable cv;
        bool active = false, parked = false, go = false, done = false;
        uint64_t* tau = nullptr; int64_t ntau = 0;
        int status = 0; std::string err;
    };
    std::unique_ptr<TwoCall> tc;

    // single-process multi-GPU handle (rsx_sharded_create): this object owns one child index per device and nothing else
    std::vector<rsx_index*> shards;
    DevBuf sh_D, sh_I, sh_q, sh_oD, sh_oI;      // parent-device gather / merge buffers
    int64_t sh_next_id = 0;                      // next sequential id of the logical index

    // rows a list's storage starts on and grows by: 64-vector slabs of PQ codes — 512 (a slice-major group of 16 blocks) in the sliced layout
    int row_align() const { return kind == KIND_IVFPQ ? (CB == PQ_SLICED ? 32 * PQ_SLICED_GB : 64) : (kind == KIND_FLAT ? 128 : 64); }
    size_t row_bytes() const { return kind == KIND_IVFPQ ? (size_t)Mpad : (size_t)ld * (storage_f16 ? 2 : 4); }
};


// ---------------------------------------------------------------------------------------
// RSX1 file records (api_build.hip: save_impl / load_impl; api_sharded.hip re-shards a plain file while loading)
// ---------------------------------------------------------------------------------------
struct FileHeader {
    char magic[4];
    int32_t version, kind, d, metric, nlist, M, nbits, trained, storage_f16, custom_ids, nprobe;
    int64_t ntotal;
};
// version >= 2 appends: what a LIST shard (rsx_set_param "add_list_mod") needs to keep assigning the logical index's
// sequential ids after a reload — the vectors it saw but did not keep, and its (mod, rem)
struct FileHeaderV2 { int64_t ndropped; int32_t add_list_mod, add_list_rem; };

static inline void wr(FILE* f, const void* p, size_t n) {
    if (n && fwrite(p, 1, n, f) != n) RSX_THROW(RSX_ERR_IO, "short write");
}
static inline void rd(FILE* f, void* p, size_t n) {
    if (n && fread(p, 1, n, f) != n) RSX_THROW(RSX_ERR_IO, "short read (truncated index file)");
}

// ---------------------------------------------------------------------------------------
// host functions that cross files (definitions: see the file list above)
// ---------------------------------------------------------------------------------------
void use_device(rsx_index* h);
void ensure_side_stream(rsx_index* h);
void refuse_while_two_call(const rsx_index* h, const char* what);
int64_t workspace_bytes(const rsx_index* h);
void upload_dir(rsx_index* h);
void ensure_capacity(rsx_index* h, const std::vector<int64_t>& need, bool exact);
const void* stage_rows(rsx_index* h, DevBuf& buf, const void* x, int64_t n, int d, int dtype);
void set_centroids(rsx_index* h, const float* c);
void set_codebooks(rsx_index* h, const float* c);
void update_trained(rsx_index* h);
rsx_index* create_common(int kind, int d, int nlist, int M, int nbits, int metric, int device);
void add_all(rsx_index* h, int64_t n, const void* x, int dtype, const int64_t* ids);
void train_impl(rsx_index* h, int64_t n, const void* x, int dtype);
void search_impl(rsx_index* h, int64_t nq, const void* q, int dtype, int k, float* D, int64_t* I);
void get_list_impl(rsx_index* h, int64_t l, int64_t* n_out, void* codes_out, int64_t* ids_out);
void add_list_impl(rsx_index* h, int64_t l, int64_t n, const void* codes, int dtype, const int64_t* ids);
void save_impl(rsx_index* h, const char* path);
rsx_index* load_impl(const char* path, int device);
rsx_index* sharded_create(int kind, int d, int nlist, int M, int nbits, int metric, int ndev, const int* devices);
void sharded_sync_trained(rsx_index* h);
void sharded_add(rsx_index* h, int64_t n, const void* x, int dtype, const int64_t* ids);
void sharded_search(rsx_index* h, int64_t nq, const void* q, int dtype, int k, float* D, int64_t* I);
void sharded_add_list(rsx_index* h, int64_t l, int64_t n, const void* codes, int dtype, const int64_t* ids);
void sharded_reserve(rsx_index* h, const int64_t* counts);
void sharded_save(rsx_index* h, const char* path);
void destroy_handle(rsx_index* h);
rsx_index* sharded_load(const char* path, int ndev, const int* devices);

static inline bool is_sharded(const rsx_index* h) { return !h->shards.empty(); }

template <typename F>
static inline void for_each_shard_parallel(rsx_index* h, F&& f) {
    const size_t n = h->shards.size();
    std::vector<std::string> errs(n);
    std::vector<int> codes(n, 0);
    std::vector<std::thread> th;
    for (size_t r = 0; r < n; r++)
        th.emplace_back([&, r] {
            try { (void)hipSetDevice(h->shards[r]->device); f((int)r, h->shards[r]); }
            catch (const RsxError& e) { errs[r] = e.what(); codes[r] = e.code; }
            catch (const std::exception& e) { errs[r] = e.what(); codes[r] = RSX_ERR_INVALID; }
        });
    for (auto& t : th) t.join();
    for (size_t r = 0; r < n; r++)
        if (codes[r]) RSX_THROW(codes[r], "shard %zu (device %d): %s", r, h->shards[r]->device, errs[r].c_str());
}

