/*
 * rsx.h — C ABI of librsx.so, the MI355X (gfx950) dense-retrieval search engine.
 *
 * This is the drop-in boundary for the search path of RulinShao/retrieval-scaling.
 * In the reference every entry point below is a call into the third-party FAISS 1.8.0
 * SWIG module (the reference holds no search arithmetic of its own); the citation on each
 * function names the reference call site(s) it replaces (paths relative to the reference
 * repository root).
 *
 * Conventions
 *   - every function returns an int status: RSX_OK (0) or a negative rsx_status;
 *     rsx_last_error() returns a thread-local message for the last failure on this thread
 *     (FAISS raises C++ exceptions that SWIG turns into Python RuntimeError; the Python shim
 *     rsx.py raises RuntimeError(rsx_last_error()) in the same places).
 *   - plain pointers + sizes only; no torch / numpy types cross this boundary.
 *   - "x"/"q" pointers may be HOST or DEVICE (HBM) pointers; the library detects which with
 *     hipPointerGetAttributes.  Output D/I follow the same rule.  The caller owns inputs and
 *     outputs; the handle owns all device memory of the index; no pointer is retained
 *     past the call.
 *   - one in-flight call per handle; callable from any host thread (hipSetDevice per call).
 *   - vectors are row-major, C-contiguous [n, d]; dtype is RSX_F32 or RSX_F16.
 *   - search returns D float32 [nq,k], I int64 [nq,k]; unfilled slots I=-1 and
 *     D=-inf (inner product) / +inf (L2), exactly as faiss.Index.search does.
 */
#ifndef RSX_H
#define RSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rsx_index rsx_index_t;

enum rsx_status {
    RSX_OK = 0,
    RSX_ERR_INVALID = -1,      /* bad argument (shape, dtype, k, null pointer) */
    RSX_ERR_NOT_TRAINED = -2,  /* add/search before train / set_centroids */
    RSX_ERR_HIP = -3,          /* HIP runtime failure (message has hipGetErrorString) */
    RSX_ERR_OOM = -4,          /* device or host allocation failed */
    RSX_ERR_IO = -5,           /* save/load failure */
    RSX_ERR_UNSUPPORTED = -6   /* valid FAISS request this build does not implement */
};

enum rsx_metric { RSX_METRIC_INNER_PRODUCT = 0, RSX_METRIC_L2 = 1 }; /* = faiss.METRIC_* */
enum rsx_dtype { RSX_F32 = 0, RSX_F16 = 1 };
enum rsx_kind { RSX_KIND_FLAT = 0, RSX_KIND_IVFFLAT = 1, RSX_KIND_IVFPQ = 2 };

/* Thread-local message of the last error raised on the calling thread ("" if none). */
const char* rsx_last_error(void);
/* ABI version (major*1000+minor). */
int rsx_version(void);
/* Number of visible HIP devices; fails with RSX_ERR_HIP when there is no usable GPU.
 * There is NO CPU fallback anywhere in this library. */
int rsx_device_count(int* n);

/* ---- construction ------------------------------------------------------------------ */

/* faiss.IndexFlatIP(d)                         — src/indicies/flat.py:42
 * (also the coarse quantiser object of ivf_flat.py:143 / ivf_pq.py:146, which this
 *  library folds into the IVF handles). Vectors are stored as fp16 when every added value
 *  is fp16-representable (the reference's embeddings are, src/embed.py:137-138), else fp32. */
int rsx_flat_create(int d, int metric, int device, rsx_index_t** out);

/* faiss.IndexIVFFlat(quantizer, d, nlist, METRIC_INNER_PRODUCT) — src/indicies/ivf_flat.py:144-148 */
int rsx_ivfflat_create(int d, int nlist, int metric, int device, rsx_index_t** out);

/* faiss.IndexIVFPQ(quantizer, d, nlist, M, nbits, METRIC_INNER_PRODUCT) — src/indicies/ivf_pq.py:147-153
 * by_residual = true (FAISS default). nbits must be 8.  The coarse quantiser is the reference's IndexFlatIP (ivf_pq.py:146) for
 * either metric.  RSX_METRIC_INNER_PRODUCT (what the reference passes) takes the certified 8-bit-table fast scans;
 * RSX_METRIC_L2 (round 6) ranks by the squared distance to the decoded vector c_l + r^ — a table per (query, list) pair, so it
 * is served by the exact scan (k_pq_scan_l2), every code layout, any k <= 4096 (tests/test_gpu_ivf.py::test_ivfpq_l2_metric_vs_oracle). */
int rsx_ivfpq_create(int d, int nlist, int M, int nbits, int metric, int device, rsx_index_t** out);

/* ONE handle over several GPUs of the node, single process — SURVEY.md 8(b)/(e).  The reference's offline driver makes one
 * index.search(all_queries, k) call (src/search.py:296) and its serving tier fans a query out to shard workers over HTTP and
 * re-sorts (api/serve_main_node.py:281-323); with this handle the same `Indexer(cfg).search` spans the node:
 *   kind: rsx_kind; nlist / M / nbits as for the single-device constructors (ignored where they do not apply);
 *   devices[ndev]: one shard per entry (an ordinal may repeat: several shards on one GPU).
 * train trains the first shard and copies the parameters; every add call is cut into ndev contiguous pieces (piece r ->
 * shard r) that keep the logical index's sequential ids, so the shards' lists are a partition of the single index's lists;
 * search runs every shard from its own host thread on its own device, copies the [nq, k] blocks to devices[0] and merges
 * them there by (score desc, id asc) — bit-identical to one index holding everything.  Every other entry point accepts the
 * handle (rsx_get "ntotal" sums the shards, knobs apply to all shards, "nshards" tells them apart) except the per-list
 * import / export calls.  rsx_save writes a manifest at `path` and one RSX1 file per shard beside it (path.shard<r>);
 * rsx_load_sharded reads them back onto the given devices (shard r -> devices[r % ndev]). */
int rsx_sharded_create(int kind, int d, int nlist, int M, int nbits, int metric, int ndev, const int* devices,
                       rsx_index_t** out);
int rsx_load_sharded(const char* path, int ndev, const int* devices, rsx_index_t** out);

/* Python garbage collection of the SWIG object (implicit in the reference). */
int rsx_destroy(rsx_index_t* h);

/* ---- training ---------------------------------------------------------------------- */

/* index.train(x)                               — src/indicies/ivf_flat.py:162,166; ivf_pq.py:166,170
 * Coarse k-means with an inner-product quantiser (assignment = argmax <x,c>, spherical
 * centroids, FAISS Level1Quantizer defaults: niter 10, <=256 points per centroid) and, for
 * IVFPQ, per-subspace k-means (256 codewords, niter 25) on residuals.
 * Assignment runs on the GPU; this also replaces the CUDA-only
 * index_cpu_to_gpu/GpuClonerOptions branch of ivf_flat.py:152-163. No-op for Flat. */
int rsx_train(rsx_index_t* h, int64_t n, const void* x, int dtype);

/* Import already-trained parameters (what faiss.read_index(<...>.trained) restores:
 * ivf_flat.py:170, ivf_pq.py:174).  centroids: [nlist, d] f32.  codebooks: [M, 256, d/M] f32. */
int rsx_set_centroids(rsx_index_t* h, const float* centroids);
int rsx_set_codebooks(rsx_index_t* h, const float* codebooks);
int rsx_get_centroids(rsx_index_t* h, float* centroids_out);
int rsx_get_codebooks(rsx_index_t* h, float* codebooks_out);

/* ---- population -------------------------------------------------------------------- */

/* index.add(x)                                 — flat.py:58; ivf_flat.py:180; ivf_pq.py:185
 * ids == NULL assigns sequential ids ntotal .. ntotal+n-1 (the only form the reference uses).
 * IVF: list = argmax-IP centroid; IVFPQ: residual = x - centroid, code = per-subspace nearest
 * (L2) codeword, first minimum on ties; in-list order = insertion order. */
int rsx_add(rsx_index_t* h, int64_t n, const void* x, int dtype, const int64_t* ids);

/* index.quantizer.assign(x) (FAISS Index::assign on the coarse quantiser): list number of each vector,
 * argmax <x, centroid> (first maximum) — the same exact fp32 kernel `add` uses.  labels: int64 [n].
 * Lets a caller count list sizes first and rsx_reserve_lists exactly (needed when the index cannot hold
 * two copies of itself in HBM, e.g. 100M x 768 fp16 IVF-Flat = 153.6 GB). */
int rsx_assign(rsx_index_t* h, int64_t n, const void* x, int dtype, int64_t* labels);

/* index.reset(): drop every stored vector, keep the trained parameters and the HBM reservation. */
int rsx_reset(rsx_index_t* h);

/* Optional: pre-size every inverted list (counts[nlist]) so add never re-lays-out HBM. */
int rsx_reserve_lists(rsx_index_t* h, const int64_t* counts);

/* Bulk import of one inverted list (FAISS on-disk reader; InvertedLists::add_entries).
 * IVFPQ: codes [n, M] u8.  IVFFlat: codes = raw vectors [n, d] in `dtype`. */
int rsx_add_list(rsx_index_t* h, int64_t list_no, int64_t n, const void* codes, int dtype,
                 const int64_t* ids);

/* Export one inverted list (inspection / parity tests / writer).  Pass NULL to skip an output.
 * n_out: list length.  codes_out: IVFPQ [n,M] u8; IVFFlat/Flat [n,d] f32 (list_no ignored
 * for Flat).  ids_out: [n]. */
int rsx_get_list(rsx_index_t* h, int64_t list_no, int64_t* n_out, void* codes_out,
                 int64_t* ids_out);

/* len(list l) for every inverted list: sizes int64 [nlist] (Flat: one entry = ntotal).  Host pointer.  Used for the
 * list-length histogram SURVEY.md 8(d) asks the bench to report (FAISS: index.invlists.list_size(l)). */
int rsx_get_list_sizes(rsx_index_t* h, int64_t* sizes);

/* ---- search ------------------------------------------------------------------------ */

/* index.nprobe = probe                         — ivf_flat.py:73,149; ivf_pq.py:77,154 */
int rsx_set_nprobe(rsx_index_t* h, int nprobe);

/* index.search(x, k) -> (D, I)   THE HOT PATH  — flat.py:139; ivf_flat.py:225; ivf_pq.py:230
 * q: [nq, d] dtype; D: float32 [nq, k]; I: int64 [nq, k].
 * Result order: score descending (IP) / distance ascending (L2); equal scores by id ascending. */
int rsx_search(rsx_index_t* h, int64_t nq, const void* q, int dtype, int k, float* D, int64_t* I);

/* The same search in TWO calls, for callers that improve the per-query thresholds between the threshold pre-pass and the scan —
 * the LIST-sharded multi-GPU form (rsx_set_param "add_list_mod"): a rank that does not own a query's closest lists derives
 * a weak threshold from its own lists; after an all-reduce(MAX) of the threshold keys every rank filters as hard as the single
 * index would (the reference has no counterpart: its shards are independent indexes, src/search.py:282-303).
 *   rsx_search_prepass: starts rsx_search(h, nq, q, dtype, k, D, I) and returns once the thresholds are final on the device:
 *       *tau_dev = device pointer to nq uint64 keys (high word = order-preserving score bits: compare as unsigned; 0 = no
 *       threshold), valid until rsx_search_scan; *tau_dev = NULL, *ntau = 0 when this search has no pre-pass (Flat, exact
 *       modes, nprobe 1).  Any key may be RAISED to another valid lower bound of the query's k-th best score; nq must not
 *       exceed "query_batch".
 *   rsx_search_scan: runs the rest; D / I (given to the first call) are complete when it returns its status. */
int rsx_search_prepass(rsx_index_t* h, int64_t nq, const void* q, int dtype, int k, float* D, int64_t* I, uint64_t** tau_dev,
                       int64_t* ntau);
int rsx_search_scan(rsx_index_t* h);

/* Multi-shard merge of per-shard top-k       — src/search.py:362-367 (post_hoc_merge_topk),
 *                                               api/serve_main_node.py:150-163 (rerank_elements)
 * D,I: [nshards, nq, k].  Output [nq, k]: best first; ties keep the earlier shard first, then the
 * original within-shard order (Python's stable sorted(..., reverse=True)).  ids < 0 are padding.
 * Pointers may be host or device (all four on the same side).  Any shard count for k <= 8192: up to 16384 keys per
 * query merge in one launch, more (the reference backends' default k = 4096, src/indicies/flat.py:138, on 8 ranks) in
 * rounds over groups of consecutive shards with the same result. */
int rsx_merge_topk(int nshards, int64_t nq, int k, int metric, const float* D, const int64_t* I,
                   float* D_out, int64_t* I_out, int device);

/* The same merge for the in-node multi-GPU exchange (replaces the HTTP fan-in of api/serve_main_node.py:281-323):
 * rsx_pack_topk turns one rank's (D, I) [nq, k] into its packed [2, nq, k] int64 block (plane 0 = the score's
 * bits in the low word, plane 1 = ids + id_offset, padding ids < 0 kept) — ONE buffer per rank, so ONE
 * all-gather; rsx_merge_packed merges the gathered [nshards, 2, nq, k] buffer with rsx_merge_topk's rule.
 * Device pointers; both calls run on `stream` (a hipStream_t, NULL = the default stream) and do NOT
 * synchronise, so they can sit between the search and the collective on the caller's stream. */
int rsx_pack_topk(int64_t nq, int k, const float* D, const int64_t* I, int64_t id_offset, int64_t* packed,
                  int device, void* stream);
int rsx_merge_packed(int nshards, int64_t nq, int k, int metric, const int64_t* packed, float* D_out,
                     int64_t* I_out, int device, void* stream);

/* ---- introspection / knobs --------------------------------------------------------- */

/* Integer properties: "ntotal", "nlist", "d", "is_trained", "nprobe", "M", "nbits", "kind",
 * "metric", "storage_dtype", "code_size", "device", "max_k", "pq_layout" (IVFPQ: 1 = rotated code layout), "nshards"
 * (0 = single-device handle), "hbm_bytes".
 * (index.ntotal / index.is_trained — ivf_flat.py:171; ivf_pq.py:175) */
int rsx_get(rsx_index_t* h, const char* key, int64_t* out);

/* Tuning knobs that do not change results: "query_batch" (max queries per internal pass),
 * "scan_chunk" (vectors per scan work item, 0 = auto), "scan_kernel" (IVFPQ: 0 = auto, 1 = per-pair
 * v1 kernel, 2 = exact list-major kernel; non-zero disables the fast scan), "pq_fast" (IVFPQ: 1 = 8-bit-table
 * fast scan with certified exact re-rank [default], 0 = exact scan only), "pq_fast_kp" (candidates kept by the
 * fast scan, 0 = auto), "pq_filter" (fast scan: 1 = candidates filtered inside the scan kernel [default], 0 = full
 * score buffer), "pq_pre_rows" (filtered fast scan: vectors of each query's closest list scored by the threshold
 * pre-pass, default 2048, 0 = one scan tile), "ivf_filter" (IVF-Flat: candidates filtered inside the list scan: 1 = when
 * the score rows of the batch would exceed ~2 GB [default], 2 = always, 0 = never = full score rows), "pq_prepass_fused" (filtered fast scan: 1 = threshold pre-pass in one launch [default], 0 = grouping +
 * scan + selection launches), "lut_tiled" (IVFPQ fast scan, dsub = 8: 1 = 8-bit tables built by
 * codebook-slice tiles shared by 32 queries [default], 0 = one workgroup per query), "flat_filter" (Flat: 1 = one filtered GEMM launch after the first chunk [default], 0 = score buffer
 * per chunk), "profile" (1 = record stage timings with HIP
 * events on the library's stream; 2 = additionally count the vectors each search scanned). */
/* Build-time knob (IVF indexes, before the first add): "add_list_mod" = N, "add_list_rem" = r make this handle a LIST
 * shard of an N-way multi-GPU index: rsx_add assigns every vector of the stream (sequential ids keep counting all
 * of them) but stores only those whose inverted list l has l % N == r.  N handles fed the same stream, searched with
 * the same queries and merged (rsx_merge_topk / rsx_merge_packed) return the single index's result; unlike vector
 * shards, every rank then scans whole lists for 1/N of the (query, probe) pairs. */
int rsx_set_param(rsx_index_t* h, const char* key, double value);

/* HIP-event timings (ms) of the stages of the last rsx_search on this handle when
 * "profile"=1: "coarse", "select_probe", "lut", "scan", "select", "finalize", "total",
 * "scan_launches", "fast_queries", "fallback_queries" (certificate failures re-run exactly) and (profile 2)
 * "scanned_vectors".  Used by bench.py for the roofline object. */
int rsx_get_timing(rsx_index_t* h, const char* key, double* ms);

/* ---- persistence ------------------------------------------------------------------- */

/* faiss.write_index(index, path) / faiss.read_index(path)
 *   — flat.py:39,63,69; ivf_flat.py:71,167,170,185; ivf_pq.py:75,171,174,190
 * Native container (magic "RSX1"); the FAISS .faiss reader lives in the Python host layer. */
int rsx_save(rsx_index_t* h, const char* path);
int rsx_load(const char* path, int device, rsx_index_t** out);

/* ---- synthetic data (bench / tests) ------------------------------------------------ */

/* Deterministic Gaussian-mixture generator, bit-identical to oracle/orc_synth (integer hash +
 * Irwin-Hall normal approximation, no transcendental functions):
 *   centre(j)[t]   = 1.0 * z(seed_c, j, t)
 *   vector(i)[t]   = fp16( centre(h(seed_x,i) % ncentres)[t] + sigma * z(seed_x, i, t) )
 * writes rows [i0, i0+n) as fp16 into `out` (host or device pointer, [n, d] fp16). */
int rsx_synth_vectors(int device, int d, int ncentres, uint32_t seed_c, uint32_t seed_x,
                      float sigma, int64_t i0, int64_t n, void* out_f16);
/* queries: q(r) = fp16( vector(h(seed_q, r) % nbase) + sigma_q * z(seed_q, r, t) ) */
int rsx_synth_queries(int device, int d, int ncentres, uint32_t seed_c, uint32_t seed_x,
                      float sigma, int64_t nbase, uint32_t seed_q, float sigma_q, int64_t r0,
                      int64_t n, void* out_f16);

#ifdef __cplusplus
}
#endif
#endif /* RSX_H */
