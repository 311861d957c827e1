/*
 * rsx_oracle.c — CPU ORACLE for the dense-retrieval search path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product (librsx.so, rsx.py, src/indicies/) never links, imports or
 * calls anything in oracle/.
 *
 * PARITY UNPINNED.  The arithmetic of the reference path lives in the third-party FAISS 1.8.0
 * wheel (environment.yml:11 faiss-gpu=1.8.0 / environment_cpu.yml:12 faiss-cpu=1.8.0), which
 * is NOT vendored in the reference repository and NOT installable here (no network); the
 * reference has no tests, golden vectors or fixtures for this path.
 * This file therefore restates the PUBLISHED FAISS-CPU algorithms that the reference's call
 * sites reach:
 *
 *   reference call site                         FAISS 1.8.0 routine restated here
 *   src/indicies/flat.py:139                    IndexFlatIP::search      -> orc_flat_search
 *   src/indicies/ivf_flat.py:225 (+:73 nprobe)  IndexIVFFlat::search     -> orc_ivfflat_search
 *   src/indicies/ivf_pq.py:230  (+:77 nprobe)   IndexIVFPQ::search       -> orc_ivfpq_search (inner product, what the reference builds)
 *                                                                         -> orc_ivfpq_search_l2 (METRIC_L2 over the same IP quantiser)
 *   src/indicies/ivf_flat.py:143, ivf_pq.py:146 IndexFlatIP quantizer    -> orc_coarse_probe / orc_assign_ip
 *   src/indicies/ivf_*.py train (:162/:166)     Clustering::train, ProductQuantizer::train
 *                                                                         -> orc_kmeans / orc_pq_train
 *   src/indicies/ivf_pq.py:185 add              ProductQuantizer::compute_code on residuals
 *                                                                         -> orc_pq_encode
 *   src/search.py:362-367 merge                 Python stable sorted(reverse=True)[:n_docs]
 *                                                                         -> orc_merge_topk
 *
 * Where FAISS leaves floating-point summation order to BLAS/SIMD (unobservable here) this
 * oracle fixes a canonical arithmetic, stated per function:
 *   - Flat / IVF-Flat scores: the exact dot product accumulated in fp64 and rounded once to
 *     fp32.  The reference's embeddings and queries are fp16-valued (src/embed.py:137-138,
 *     src/search.py:257-258) so every product is exact and the fp64 sum is (to 2^-53) the true
 *     value: an order-independent definition of the "right" answer FAISS approximates.
 *   - coarse quantiser and IVF-PQ: fp32, one sequential fmaf chain per dot product
 *     (t = 0..d-1, accumulator starts at +0), LUT entries likewise, and
 *     score = dis0 + (((0 + T[0][c0]) + T[1][c1]) + ... ) exactly as FAISS's generic
 *     IVFPQ scanner accumulates (dis0 + distance_single_code).
 *   - result order: score descending, equal scores by id ascending (a total order; FAISS's
 *     heap breaks exact ties by heap mechanics).  orc_ivfpq_search_heap keeps FAISS's CMin heap
 *     admission rule (strict >) and is the timed CPU baseline.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------ */
/* fp16 <-> fp32 (IEEE binary16, round-to-nearest-even), no compiler half type needed.  */

static inline float orc_h2f(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff, u;
    if (exp == 0) {
        if (man == 0) {
            u = sign;
        } else { /* subnormal */
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400));
            man &= 0x3ff;
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        u = sign | 0x7f800000u | (man << 13);
    } else {
        u = sign | ((exp + 112) << 23) | (man << 13);
    }
    float f; memcpy(&f, &u, 4); return f;
}

static inline uint16_t orc_f2h(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((a > 0x7f800000u) ? 0x200u : 0));
    if (a >= 0x477ff000u) { /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (a < 0x38800000u) { /* < 2^-14: subnormal half or zero */
        if (a < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 (2^-25 ties to even 0) */
        int e = (int)(a >> 23);
        uint32_t m = (a & 0x7fffffu) | 0x800000u;
        int shift = 126 - e; /* 14..24 */
        uint32_t r = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t e = (a >> 23) - 112, m = a & 0x7fffffu;
    uint32_t r = (e << 10) | (m >> 13);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
    return (uint16_t)(sign | r);
}

void orc_half_to_float(int64_t n, const uint16_t* in, float* out) {
    for (int64_t i = 0; i < n; i++) out[i] = orc_h2f(in[i]);
}
void orc_float_to_half(int64_t n, const float* in, uint16_t* out) {
    for (int64_t i = 0; i < n; i++) out[i] = orc_f2h(in[i]);
}

/* ------------------------------------------------------------------------------------ */
/* Synthetic Gaussian-mixture data, integer hash + Irwin-Hall(8) normal   */
/* approximation so that CPU and GPU generate bit-identical fp16 values.                   */

static inline uint32_t orc_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static inline uint32_t orc_hash4(uint32_t seed, uint64_t i, uint32_t t, uint32_t s) {
    uint32_t x = orc_mix32(seed + 0x9E3779B9u * (s + 1u));
    x = orc_mix32(x ^ (uint32_t)(i & 0xffffffffu));
    x = orc_mix32(x ^ ((uint32_t)(i >> 32) * 0x85EBCA6Bu) ^ (t * 0xC2B2AE35u));
    return x;
}
/* ~N(0,1): sum of eight 16-bit uniforms, centred and scaled by 1/(65536*sqrt(8/12)). */
static inline float orc_z(uint32_t seed, uint64_t i, uint32_t t) {
    uint32_t S = 0;
    for (uint32_t s = 0; s < 4; s++) {
        uint32_t h = orc_hash4(seed, i, t, s);
        S += (h & 0xffffu) + (h >> 16);
    }
    return ((float)S - 262140.0f) * 1.8688064e-5f;
}
static inline uint32_t orc_pick(uint32_t seed, uint64_t i, uint32_t n) {
    return orc_hash4(seed, i, 0xffffffffu, 7u) % n;
}
static inline uint16_t orc_synth_elem(int ncentres, uint32_t seed_c, uint32_t seed_x, float sigma,
                                      uint64_t i, uint32_t t) {
    uint32_t j = orc_pick(seed_x, i, (uint32_t)ncentres);
    float c = orc_z(seed_c, j, t);
    return orc_f2h(fmaf(sigma, orc_z(seed_x, i, t), c));
}
void orc_synth_vectors(int d, int ncentres, uint32_t seed_c, uint32_t seed_x, float sigma,
                       int64_t i0, int64_t n, uint16_t* out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; r++)
        for (int t = 0; t < d; t++)
            out[r * d + t] = orc_synth_elem(ncentres, seed_c, seed_x, sigma, (uint64_t)(i0 + r), (uint32_t)t);
}
void orc_synth_queries(int d, int ncentres, uint32_t seed_c, uint32_t seed_x, float sigma,
                       int64_t nbase, uint32_t seed_q, float sigma_q, int64_t r0, int64_t n,
                       uint16_t* out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; r++) {
        uint64_t rr = (uint64_t)(r0 + r);
        uint64_t b = orc_pick(seed_q, rr, (uint32_t)nbase);
        for (int t = 0; t < d; t++) {
            float base = orc_h2f(orc_synth_elem(ncentres, seed_c, seed_x, sigma, b, (uint32_t)t));
            out[r * d + t] = orc_f2h(fmaf(sigma_q, orc_z(seed_q, rr, (uint32_t)t), base));
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* Canonical top-k: (score desc, id asc) for IP; (dist asc, id asc) for L2.                */

typedef struct { float s; int64_t id; } orc_cand;

static inline int orc_better_ip(float sa, int64_t ia, float sb, int64_t ib) {
    return (sa > sb) || (sa == sb && ia < ib);
}
static int orc_cmp_ip(const void* a, const void* b) {
    const orc_cand *x = (const orc_cand*)a, *y = (const orc_cand*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->id < y->id) ? -1 : (x->id > y->id);
}
static int orc_cmp_l2(const void* a, const void* b) {
    const orc_cand *x = (const orc_cand*)a, *y = (const orc_cand*)b;
    if (x->s < y->s) return -1;
    if (x->s > y->s) return 1;
    return (x->id < y->id) ? -1 : (x->id > y->id);
}

/* bounded best-k set kept as a binary heap whose root is the WORST kept element under the
 * total order; exact (no ties lost). */
typedef struct { orc_cand* h; int k, n, l2; } orc_topk;
static inline int orc_worse(const orc_topk* t, const orc_cand* a, const orc_cand* b) {
    /* is a worse than b */
    if (t->l2) return (a->s > b->s) || (a->s == b->s && a->id > b->id);
    return (a->s < b->s) || (a->s == b->s && a->id > b->id);
}
static void orc_topk_push(orc_topk* t, float s, int64_t id) {
    orc_cand c = { s, id };
    if (s != s) return; /* NaN never admitted (FAISS comparisons are false) */
    if (t->n < t->k) {
        int i = t->n++;
        t->h[i] = c;
        while (i > 0) {
            int p = (i - 1) / 2;
            if (orc_worse(t, &t->h[i], &t->h[p])) { orc_cand x = t->h[i]; t->h[i] = t->h[p]; t->h[p] = x; i = p; }
            else break;
        }
        return;
    }
    if (t->k == 0 || !orc_worse(t, &t->h[0], &c)) return;
    t->h[0] = c;
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, w = i;
        if (l < t->n && orc_worse(t, &t->h[l], &t->h[w])) w = l;
        if (r < t->n && orc_worse(t, &t->h[r], &t->h[w])) w = r;
        if (w == i) break;
        orc_cand x = t->h[i]; t->h[i] = t->h[w]; t->h[w] = x; i = w;
    }
}
static void orc_topk_finish(orc_topk* t, float* D, int64_t* I) {
    qsort(t->h, (size_t)t->n, sizeof(orc_cand), t->l2 ? orc_cmp_l2 : orc_cmp_ip);
    for (int i = 0; i < t->k; i++) {
        if (i < t->n) { D[i] = t->h[i].s; I[i] = t->h[i].id; }
        else { D[i] = t->l2 ? INFINITY : -INFINITY; I[i] = -1; }
    }
}

/* ------------------------------------------------------------------------------------ */
/* Flat: IndexFlatIP / IndexFlatL2 search, canonical exact arithmetic.                     */
/* metric 0 = inner product, 1 = squared L2.  xb [nb,d] f32, xq [nq,d] f32.                */

static inline float orc_exact_ip(int d, const float* a, const float* b) {
    double s = 0.0;
    for (int t = 0; t < d; t++) s += (double)a[t] * (double)b[t];
    return (float)s;
}
static inline float orc_exact_l2(int d, const float* a, const float* b) {
    double s = 0.0;
    for (int t = 0; t < d; t++) { double df = (double)a[t] - (double)b[t]; s += df * df; }
    return (float)s;
}

void orc_flat_search(int metric, int d, int64_t nb, const float* xb, const int64_t* ids,
                     int64_t nq, const float* xq, int k, float* D, int64_t* I) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t q = 0; q < nq; q++) {
        orc_topk t = { (orc_cand*)malloc(sizeof(orc_cand) * (size_t)(k > 0 ? k : 1)), k, 0, metric };
        for (int64_t i = 0; i < nb; i++) {
            float s = metric ? orc_exact_l2(d, xq + q * d, xb + i * d) : orc_exact_ip(d, xq + q * d, xb + i * d);
            orc_topk_push(&t, s, ids ? ids[i] : i);
        }
        orc_topk_finish(&t, D + q * k, I + q * k);
        free(t.h);
    }
}

/* ------------------------------------------------------------------------------------ */
/* Coarse quantiser: IndexFlatIP over centroids, fp32 sequential fmaf chain.               */

static inline float orc_dot_f32(int d, const float* a, const float* b) {
    float s = 0.0f;
    for (int t = 0; t < d; t++) s = fmaf(a[t], b[t], s);
    return s;
}

/* assign[i] = argmax_c <x_i, centroid_c>, first maximum on ties (FAISS quantizer->assign). */
void orc_assign_ip(int d, int nlist, const float* centroids, int64_t n, const float* x,
                   int32_t* assign, float* best_out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        float best = -INFINITY; int bi = 0;
        for (int c = 0; c < nlist; c++) {
            float s = orc_dot_f32(d, x + i * d, centroids + (int64_t)c * d);
            if (s > best) { best = s; bi = c; }
        }
        assign[i] = bi;
        if (best_out) best_out[i] = best;
    }
}

/* probe_ids/probe_scores [nq,nprobe]: quantizer->search(nq, x, nprobe), best first,
 * ties by list number ascending; missing -> -1. */
void orc_coarse_probe(int d, int nlist, const float* centroids, int64_t nq, const float* xq,
                      int nprobe, int64_t* probe_ids, float* probe_scores) {
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < nq; q++) {
        orc_topk t = { (orc_cand*)malloc(sizeof(orc_cand) * (size_t)nprobe), nprobe, 0, 0 };
        for (int c = 0; c < nlist; c++)
            orc_topk_push(&t, orc_dot_f32(d, xq + q * d, centroids + (int64_t)c * d), c);
        orc_topk_finish(&t, probe_scores + q * nprobe, probe_ids + q * nprobe);
        free(t.h);
    }
}

/* ------------------------------------------------------------------------------------ */
/* IVF-Flat search.  Lists are given list-major: list l occupies rows                      */
/* [list_off[l], list_off[l+1]) of vecs [ntotal,d] f32 and ids [ntotal].                   */

void orc_ivfflat_search(int metric, int d, int nlist, const float* centroids,
                        const int64_t* list_off, const float* vecs, const int64_t* ids,
                        int64_t nq, const float* xq, int nprobe, int k, float* D, int64_t* I) {
    if (nprobe > nlist) nprobe = nlist;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t q = 0; q < nq; q++) {
        int64_t* pid = (int64_t*)malloc(sizeof(int64_t) * (size_t)nprobe);
        float* ps = (float*)malloc(sizeof(float) * (size_t)nprobe);
        orc_coarse_probe(d, nlist, centroids, 1, xq + q * d, nprobe, pid, ps);
        orc_topk t = { (orc_cand*)malloc(sizeof(orc_cand) * (size_t)(k > 0 ? k : 1)), k, 0, metric };
        for (int j = 0; j < nprobe; j++) {
            int64_t l = pid[j];
            if (l < 0) continue;
            for (int64_t r = list_off[l]; r < list_off[l + 1]; r++) {
                float s = metric ? orc_exact_l2(d, xq + q * d, vecs + r * d)
                                 : orc_exact_ip(d, xq + q * d, vecs + r * d);
                orc_topk_push(&t, s, ids[r]);
            }
        }
        orc_topk_finish(&t, D + q * k, I + q * k);
        free(t.h); free(pid); free(ps);
    }
}

/* ------------------------------------------------------------------------------------ */
/* Product quantiser.  codebooks [M,256,dsub] f32.                                         */

/* ProductQuantizer::compute_code: per subspace nearest codeword by squared L2 (sequential
 * fmaf chain of squared differences), first minimum on ties.  x [n,d] f32 -> codes [n,M]. */
void orc_pq_encode(int d, int M, const float* codebooks, int64_t n, const float* x, uint8_t* codes) {
    int dsub = d / M;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        for (int m = 0; m < M; m++) {
            const float* xs = x + i * d + m * dsub;
            float best = INFINITY; int bc = 0;
            for (int c = 0; c < 256; c++) {
                const float* cw = codebooks + ((int64_t)m * 256 + c) * dsub;
                float acc = 0.0f;
                for (int t = 0; t < dsub; t++) { float df = xs[t] - cw[t]; acc = fmaf(df, df, acc); }
                if (acc < best) { best = acc; bc = c; }
            }
            codes[i * M + m] = (uint8_t)bc;
        }
    }
}

/* residual of x against its assigned centroid (IndexIVF by_residual): r = x - c, fp32. */
void orc_residuals(int d, const float* centroids, int64_t n, const float* x, const int32_t* assign,
                   float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++)
        for (int t = 0; t < d; t++) out[i * d + t] = x[i * d + t] - centroids[(int64_t)assign[i] * d + t];
}

/* inner-product look-up table of one query: T[m][c] = <q_m, cb[m][c]> (fmaf chain). */
static void orc_pq_lut_one(int d, int M, const float* codebooks, const float* q, float* T) {
    int dsub = d / M;
    for (int m = 0; m < M; m++)
        for (int c = 0; c < 256; c++)
            T[m * 256 + c] = orc_dot_f32(dsub, q + m * dsub, codebooks + ((int64_t)m * 256 + c) * dsub);
}
void orc_pq_lut(int d, int M, const float* codebooks, int64_t nq, const float* xq, float* T) {
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < nq; q++) orc_pq_lut_one(d, M, codebooks, xq + q * d, T + q * M * 256);
}

static inline float orc_adc(int M, const float* T, const uint8_t* code, float dis0) {
    float r = 0.0f;
    for (int m = 0; m < M; m++) r += T[m * 256 + code[m]];
    return dis0 + r;
}
/* Eight consecutive vectors at once: eight INDEPENDENT sequential sums (each one exactly orc_adc's: same order, same bits), so the
 * fp32 add latency of one vector's chain is hidden behind the other seven — what FAISS's scanner gets from distance_four_codes
 * (faiss/impl/ProductQuantizer / IndexIVFPQ scan_list_with_table).  Only the timed CPU baseline uses it. */
static inline void orc_adc8(int M, const float* T, const uint8_t* code, float dis0, float* out) {
    float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f, r3 = 0.0f, r4 = 0.0f, r5 = 0.0f, r6 = 0.0f, r7 = 0.0f;
    const uint8_t *c0 = code, *c1 = code + M, *c2 = code + 2 * M, *c3 = code + 3 * M, *c4 = code + 4 * M, *c5 = code + 5 * M,
                  *c6 = code + 6 * M, *c7 = code + 7 * M;
    for (int m = 0; m < M; m++) {
        const float* t = T + m * 256;
        r0 += t[c0[m]]; r1 += t[c1[m]]; r2 += t[c2[m]]; r3 += t[c3[m]];
        r4 += t[c4[m]]; r5 += t[c5[m]]; r6 += t[c6[m]]; r7 += t[c7[m]];
    }
    out[0] = dis0 + r0; out[1] = dis0 + r1; out[2] = dis0 + r2; out[3] = dis0 + r3;
    out[4] = dis0 + r4; out[5] = dis0 + r5; out[6] = dis0 + r6; out[7] = dis0 + r7;
}

/* IndexIVFPQ::search, METRIC_INNER_PRODUCT, by_residual: canonical result order.
 * codes [ntotal,M] list-major (list l rows [list_off[l], list_off[l+1])). */
void orc_ivfpq_search(int d, int nlist, int M, const float* centroids, const float* codebooks,
                      const int64_t* list_off, const uint8_t* codes, const int64_t* ids,
                      int64_t nq, const float* xq, int nprobe, int k, float* D, int64_t* I) {
    if (nprobe > nlist) nprobe = nlist;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t q = 0; q < nq; q++) {
        int64_t* pid = (int64_t*)malloc(sizeof(int64_t) * (size_t)nprobe);
        float* ps = (float*)malloc(sizeof(float) * (size_t)nprobe);
        float* T = (float*)malloc(sizeof(float) * (size_t)M * 256);
        orc_coarse_probe(d, nlist, centroids, 1, xq + q * d, nprobe, pid, ps);
        orc_pq_lut_one(d, M, codebooks, xq + q * d, T);
        orc_topk t = { (orc_cand*)malloc(sizeof(orc_cand) * (size_t)(k > 0 ? k : 1)), k, 0, 0 };
        for (int j = 0; j < nprobe; j++) {
            int64_t l = pid[j];
            if (l < 0) continue;
            for (int64_t r = list_off[l]; r < list_off[l + 1]; r++)
                orc_topk_push(&t, orc_adc(M, T, codes + r * M, ps[j]), ids[r]);
        }
        orc_topk_finish(&t, D + q * k, I + q * k);
        free(t.h); free(pid); free(ps); free(T);
    }
}

/* IndexIVFPQ::search, METRIC_L2, by_residual (an IndexIVFPQ built over the reference's inner-product coarse quantiser,
 * src/indicies/ivf_pq.py:146, with metric_type L2 — the reference itself only ever passes METRIC_INNER_PRODUCT, :152; this is the
 * other metric `north_star` names).  FAISS's definition: the squared distance between the query and the DECODED vector
 * c_l + r^, i.e. ||(q - c_l) - r^||^2 = sum over m of ||(q - c_l)_m - codebook[m][code_m]||^2.  Canonical arithmetic here: the
 * residual query qr = q - c_l in fp32 (one rounding per component), each table entry one sequential fmaf chain over the sub-vector
 * (df = qr - cb, acc = fma(df, df, acc), accumulator from +0), distance = (((0 + T[0][c0]) + T[1][c1]) + ...) — the generic
 * scanner's order; FAISS's precomputed-table form (||q - c||^2 + ||r^||^2 + 2 <c, r^> - 2 <q, r^>) differs from it only in
 * rounding.  Lists are probed by INNER PRODUCT (the quantiser's own metric).  Result order: distance ascending, ties by id. */
void orc_ivfpq_search_l2(int d, int nlist, int M, const float* centroids, const float* codebooks,
                         const int64_t* list_off, const uint8_t* codes, const int64_t* ids,
                         int64_t nq, const float* xq, int nprobe, int k, float* D, int64_t* I) {
    if (nprobe > nlist) nprobe = nlist;
    const int dsub = d / M;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t q = 0; q < nq; q++) {
        int64_t* pid = (int64_t*)malloc(sizeof(int64_t) * (size_t)nprobe);
        float* ps = (float*)malloc(sizeof(float) * (size_t)nprobe);
        float* T = (float*)malloc(sizeof(float) * (size_t)M * 256);
        float* qr = (float*)malloc(sizeof(float) * (size_t)d);
        orc_coarse_probe(d, nlist, centroids, 1, xq + q * d, nprobe, pid, ps);
        orc_topk t = { (orc_cand*)malloc(sizeof(orc_cand) * (size_t)(k > 0 ? k : 1)), k, 0, 1 };
        for (int j = 0; j < nprobe; j++) {
            int64_t l = pid[j];
            if (l < 0 || list_off[l + 1] == list_off[l]) continue;
            for (int t2 = 0; t2 < d; t2++) qr[t2] = xq[q * d + t2] - centroids[l * d + t2];
            for (int m = 0; m < M; m++)
                for (int c = 0; c < 256; c++) {
                    const float* cw = codebooks + ((int64_t)m * 256 + c) * dsub;
                    float acc = 0.0f;
                    for (int t2 = 0; t2 < dsub; t2++) { float df = qr[m * dsub + t2] - cw[t2]; acc = fmaf(df, df, acc); }
                    T[m * 256 + c] = acc;
                }
            for (int64_t r = list_off[l]; r < list_off[l + 1]; r++) {
                float dis = 0.0f;
                for (int m = 0; m < M; m++) dis += T[m * 256 + codes[r * M + m]];
                orc_topk_push(&t, dis, ids[r]);
            }
        }
        orc_topk_finish(&t, D + q * k, I + q * k);
        free(t.h); free(pid); free(ps); free(T); free(qr);
    }
}

/* FAISS-structured variant (the timed CPU baseline): per-query CMin heap of size k,
 * admission `dis > heap_top` (strict), lists scanned in coarse-rank order, OpenMP over
 * queries (parallel_mode 0), heap reordered to descending at the end. */
static inline void orc_minheap_replace_top(int k, float* v, int64_t* id, float nv, int64_t nid) {
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, w;
        if (l >= k) break;
        w = (r < k && (v[r] < v[l] || (v[r] == v[l] && id[r] < id[l]))) ? r : l;
        if (nv < v[w] || (nv == v[w] && nid < id[w])) break;
        v[i] = v[w]; id[i] = id[w]; i = w;
    }
    v[i] = nv; id[i] = nid;
}
void orc_ivfpq_search_heap(int d, int nlist, int M, const float* centroids, const float* codebooks,
                           const int64_t* list_off, const uint8_t* codes, const int64_t* ids,
                           int64_t nq, const float* xq, int nprobe, int k, float* D, int64_t* I) {
    if (nprobe > nlist) nprobe = nlist;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t q = 0; q < nq; q++) {
        int64_t* pid = (int64_t*)malloc(sizeof(int64_t) * (size_t)nprobe);
        float* ps = (float*)malloc(sizeof(float) * (size_t)nprobe);
        float* T = (float*)malloc(sizeof(float) * (size_t)M * 256);
        orc_coarse_probe(d, nlist, centroids, 1, xq + q * d, nprobe, pid, ps);
        orc_pq_lut_one(d, M, codebooks, xq + q * d, T);
        float* hv = D + q * k; int64_t* hi = I + q * k;
        for (int i = 0; i < k; i++) { hv[i] = -INFINITY; hi[i] = -1; }
        for (int j = 0; j < nprobe; j++) {
            int64_t l = pid[j];
            if (l < 0) continue;
            float dis0 = ps[j];
            const uint8_t* cp = codes + list_off[l] * M;
            int64_t len = list_off[l + 1] - list_off[l];
            int64_t r = 0;
            for (; r + 8 <= len; r += 8, cp += 8 * (int64_t)M) {      /* the same vectors in the same order, eight sums in flight */
                float dis8[8];
                orc_adc8(M, T, cp, dis0, dis8);
                for (int u = 0; u < 8; u++)
                    if (dis8[u] > hv[0]) orc_minheap_replace_top(k, hv, hi, dis8[u], ids[list_off[l] + r + u]);
            }
            for (; r < len; r++, cp += M) {
                float dis = orc_adc(M, T, cp, dis0);
                if (dis > hv[0]) orc_minheap_replace_top(k, hv, hi, dis, ids[list_off[l] + r]);
            }
        }
        /* heap_reorder: repeatedly pop the minimum to the back -> descending */
        for (int n = k; n > 1; n--) {
            float tv = hv[0]; int64_t ti = hi[0];
            float lv = hv[n - 1]; int64_t li = hi[n - 1];
            orc_minheap_replace_top(n - 1, hv, hi, lv, li);
            hv[n - 1] = tv; hi[n - 1] = ti;
        }
        /* FAISS leaves unfilled slots (id -1) at the end after reorder */
        int w = 0;
        for (int i = 0; i < k; i++) if (hi[i] >= 0) { hv[w] = hv[i]; hi[w] = hi[i]; w++; }
        for (; w < k; w++) { hv[w] = -INFINITY; hi[w] = -1; }
        free(pid); free(ps); free(T);
    }
}

/* ------------------------------------------------------------------------------------ */
/* k-means (faiss::Clustering::train restated).                                            */
/* mode 0: inner-product assignment + spherical centroids (IVF coarse quantiser built on   */
/*         IndexFlatIP with METRIC_INNER_PRODUCT);  mode 1: L2 assignment (PQ codebooks).  */

static inline uint64_t orc_splitmix(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
/* Fisher-Yates permutation of 0..n-1 driven by splitmix64(seed). */
void orc_rand_perm(int64_t n, uint64_t seed, int64_t* perm) {
    uint64_t s = seed;
    for (int64_t i = 0; i < n; i++) perm[i] = i;
    for (int64_t i = 0; i + 1 < n; i++) {
        int64_t j = i + (int64_t)(orc_splitmix(&s) % (uint64_t)(n - i));
        int64_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
}

static void orc_renorm(int d, int k, float* c) {
    for (int i = 0; i < k; i++) {
        float nr = 0.0f;
        for (int t = 0; t < d; t++) nr = fmaf(c[(int64_t)i * d + t], c[(int64_t)i * d + t], nr);
        if (nr > 0.0f) {
            float inv = 1.0f / sqrtf(nr);
            for (int t = 0; t < d; t++) c[(int64_t)i * d + t] *= inv;
        }
    }
}

static void orc_assign_l2(int d, int k, const float* c, int64_t n, const float* x, int32_t* assign) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        float best = INFINITY; int bi = 0;
        for (int j = 0; j < k; j++) {
            float acc = 0.0f;
            for (int t = 0; t < d; t++) { float df = x[i * d + t] - c[(int64_t)j * d + t]; acc = fmaf(df, df, acc); }
            if (acc < best) { best = acc; bi = j; }
        }
        assign[i] = bi;
    }
}

/* x [n,d] f32 (already subsampled/ordered by the caller through orc_kmeans_sample);
 * centroids [k,d] out.  Deterministic: update sums in increasing point order, fp32. */
void orc_kmeans(int mode, int d, int k, int64_t n, const float* x, int niter, uint64_t seed,
                float* centroids) {
    int64_t* perm = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int32_t* assign = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int64_t* hassign = (int64_t*)malloc(sizeof(int64_t) * (size_t)k);
    orc_rand_perm(n, seed + 1, perm);
    for (int j = 0; j < k; j++) memcpy(centroids + (int64_t)j * d, x + perm[j % n] * d, sizeof(float) * (size_t)d);
    if (mode == 0) orc_renorm(d, k, centroids);
    for (int it = 0; it < niter; it++) {
        if (mode == 0) orc_assign_ip(d, k, centroids, n, x, assign, NULL);
        else orc_assign_l2(d, k, centroids, n, x, assign);
        memset(centroids, 0, sizeof(float) * (size_t)k * d);
        memset(hassign, 0, sizeof(int64_t) * (size_t)k);
        for (int64_t i = 0; i < n; i++) {
            int c = assign[i];
            hassign[c]++;
            float* cc = centroids + (int64_t)c * d;
            for (int t = 0; t < d; t++) cc[t] += x[i * d + t];
        }
        for (int j = 0; j < k; j++) {
            if (hassign[j] == 0) continue;
            float norm = 1.0f / (float)hassign[j];
            for (int t = 0; t < d; t++) centroids[(int64_t)j * d + t] *= norm;
        }
        /* split_clusters: every empty cluster steals from a big one (eps = 1/1024) */
        uint64_t rs = 1234;
        for (int ci = 0; ci < k; ci++) {
            if (hassign[ci] != 0) continue;
            int cj = 0;
            for (;;) {
                double p = ((double)hassign[cj] - 1.0) / (double)(n - k);
                double r = (double)(orc_splitmix(&rs) >> 11) * (1.0 / 9007199254740992.0);
                if (r < p) break;
                cj = (cj + 1) % k;
            }
            memcpy(centroids + (int64_t)ci * d, centroids + (int64_t)cj * d, sizeof(float) * (size_t)d);
            for (int t = 0; t < d; t++) {
                if (t % 2 == 0) { centroids[(int64_t)ci * d + t] *= 1.0f + 1.0f / 1024.0f; centroids[(int64_t)cj * d + t] *= 1.0f - 1.0f / 1024.0f; }
                else { centroids[(int64_t)ci * d + t] *= 1.0f - 1.0f / 1024.0f; centroids[(int64_t)cj * d + t] *= 1.0f + 1.0f / 1024.0f; }
            }
            hassign[ci] = hassign[cj] / 2;
            hassign[cj] -= hassign[ci];
        }
        if (mode == 0) orc_renorm(d, k, centroids);
    }
    free(perm); free(assign); free(hassign);
}

/* Clustering subsampling: if n > k*max_points_per_centroid keep the first k*mppc points of
 * a seeded permutation (returned in `sel`, caller gathers). Returns number kept. */
int64_t orc_kmeans_sample(int64_t n, int k, int mppc, uint64_t seed, int64_t* sel) {
    int64_t keep = (int64_t)k * mppc;
    if (n <= keep) { for (int64_t i = 0; i < n; i++) sel[i] = i; return n; }
    int64_t* perm = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    orc_rand_perm(n, seed, perm);
    memcpy(sel, perm, sizeof(int64_t) * (size_t)keep);
    free(perm);
    return keep;
}

/* ProductQuantizer::train: independent L2 k-means (256 codewords, niter 25) per subspace.
 * x [n,d] residuals; codebooks [M,256,dsub]. */
void orc_pq_train(int d, int M, int64_t n, const float* x, int niter, uint64_t seed, float* codebooks) {
    int dsub = d / M;
    float* xs = (float*)malloc(sizeof(float) * (size_t)n * dsub);
    for (int m = 0; m < M; m++) {
        for (int64_t i = 0; i < n; i++) memcpy(xs + i * dsub, x + i * d + m * dsub, sizeof(float) * (size_t)dsub);
        orc_kmeans(1, dsub, 256, n, xs, niter, seed + (uint64_t)m, codebooks + (int64_t)m * 256 * dsub);
    }
    free(xs);
}

/* ------------------------------------------------------------------------------------ */
/* Multi-shard merge (src/search.py:362-367): concat shard results in shard order, stable   */
/* sort by score descending (L2: ascending), keep k.  id < 0 entries are padding.          */

void orc_merge_topk(int nshards, int64_t nq, int k, int metric, const float* D, const int64_t* I,
                    float* Do, int64_t* Io) {
    for (int64_t q = 0; q < nq; q++) {
        int n = 0, tot = nshards * k;
        float* s = (float*)malloc(sizeof(float) * (size_t)tot);
        int64_t* id = (int64_t*)malloc(sizeof(int64_t) * (size_t)tot);
        for (int sh = 0; sh < nshards; sh++)
            for (int j = 0; j < k; j++) {
                int64_t v = I[((int64_t)sh * nq + q) * k + j];
                if (v < 0) continue;
                s[n] = D[((int64_t)sh * nq + q) * k + j]; id[n] = v; n++;
            }
        /* stable insertion sort */
        for (int a = 1; a < n; a++) {
            float sv = s[a]; int64_t iv = id[a]; int b = a - 1;
            while (b >= 0 && (metric ? (s[b] > sv) : (s[b] < sv))) { s[b + 1] = s[b]; id[b + 1] = id[b]; b--; }
            s[b + 1] = sv; id[b + 1] = iv;
        }
        for (int j = 0; j < k; j++) {
            if (j < n) { Do[q * k + j] = s[j]; Io[q * k + j] = id[j]; }
            else { Do[q * k + j] = metric ? INFINITY : -INFINITY; Io[q * k + j] = -1; }
        }
        free(s); free(id);
    }
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
