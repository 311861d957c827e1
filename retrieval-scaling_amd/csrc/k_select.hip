// k_select.hip — k-selection, probe set-up, (query,list) grouping, final exact re-rank and the
// multi-shard merge.  All selection kernels are ONE WAVE (64 lanes) per workgroup: ballots and
// popcounts give lane-private append slots with no LDS atomics, and the workgroup barrier of a
// single-wave group is free, so many independent selections share a CU.
//
// k_select_radix (short rows: exact K'-th key by an MSB-first radix walk + one small sort) and k_pq_prepass (the IVF-PQ
// threshold pre-pass in one launch) use 256 / 1024 threads per row instead.
//
// Replaces, inside faiss.Index*.search (reference call sites flat.py:139, ivf_flat.py:225,
// ivf_pq.py:230): the per-query heap / reservoir top-k, quantizer->search top-nprobe, and
// heap_reorder; and src/search.py:362-367 (multi-shard merge).
#include "rsx_internal.h"

namespace rsx {

// Descending bitonic sort of n (power of two) 64-bit keys in LDS by one wave.
__device__ inline void bitonic_sort_desc(uint64_t* buf, int n, int lane) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (n >> 1); t += 64) {
                int i = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
                int j = i + stride;
                bool desc = ((i & size) == 0);
                uint64_t x = buf[i], y = buf[j];
                if ((x < y) == desc) { buf[i] = y; buf[j] = x; }
            }
            __syncthreads();
        }
    }
}

// Streaming threshold selection.  The wave keeps up to BUF candidate keys in LDS; a key is
// appended only if it beats tau (the k-th best key at the last prune); when the buffer would
// overflow it is sorted, cut to KP keys, and tau is raised.  Exact: a dropped key always has
// >= k strictly better keys.
template <int MODE>
__global__ __launch_bounds__(64) void k_select(SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sel_buf[];
    uint64_t* buf = sel_buf;
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;
    const int seg = blockIdx.y + a.seg_base;
    const int KP = a.KP, BUF = a.BUF, k = a.k;
    int64_t n = a.row_n ? a.row_n[row * a.row_n_stride] : a.n_uniform;
    if (a.row_n && a.n_uniform > 0 && n > a.n_uniform) n = a.n_uniform;   // counts may exceed the buffer capacity
    int64_t start = (int64_t)seg * a.seg_len;
    int64_t end = start + a.seg_len;
    if (end > n) end = n;
    const float* sf = (const float*)a.in + row * a.row_stride;
    const uint64_t* sk = (const uint64_t*)a.in + row * a.row_stride;
    if (start >= end && !a.init) {  // segment beyond the row's length: no candidates
        uint64_t* o = a.out + row * a.out_row_stride + (int64_t)seg * KP;
        for (int i = lane; i < KP; i += 64) o[i] = 0;
        if (a.zero_cnt && lane == 0 && seg == 0) a.zero_cnt[row * CCS] = 0ull;
        return;
    }

    int cnt = 0;
    uint64_t tau = a.tau_ptr ? a.tau_ptr[row * a.tau_stride] : 0ull;
    if (a.init) {  // running state: KP keys sorted descending, zero padded
        for (int i = lane; i < KP; i += 64) buf[i] = a.init[row * KP + i];
        __syncthreads();
        cnt = KP;
        tau = buf[k - 1] > tau ? buf[k - 1] : tau;
        __syncthreads();
    }
    // Round 3 — a strong STARTING threshold for score rows (the probe selection: 4096 coarse scores -> 32 keys).  One cheap
    // pass takes every lane's largest key; the k-th largest of those 64 lane maxima is a lower bound of the row's k-th best
    // key (k distinct keys reach it), so the filter below admits ~1.4 k keys instead of filling and re-sorting its buffer
    // from an empty threshold (two 256-key sorts on one wave were most of the 50 us this selection took per 1024 rows).
    if (MODE == 0 && k <= 64 && end - start >= 1024) {
        uint64_t mx = 0ull;
        for (int64_t base = start; base < end; base += 2048) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {       // eight 1 KiB tiles in flight
                const int64_t e0 = base + u * 256 + lane * 4;
                v[u] = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
                if (e0 + 3 < end) v[u] = *reinterpret_cast<const float4*>(sf + e0);
                else {
                    if (e0 < end) v[u].x = sf[e0];
                    if (e0 + 1 < end) v[u].y = sf[e0 + 1];
                    if (e0 + 2 < end) v[u].z = sf[e0 + 2];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t i0 = a.idx_base + (uint32_t)(base + u * 256 + lane * 4);
                const uint64_t k0 = make_key(v[u].x, i0), k1 = make_key(v[u].y, i0 + 1), k2 = make_key(v[u].z, i0 + 2), k3 = make_key(v[u].w, i0 + 3);
                const uint64_t m01 = k0 > k1 ? k0 : k1, m23 = k2 > k3 ? k2 : k3, m = m01 > m23 ? m01 : m23;
                mx = m > mx ? m : mx;
            }
        }
        // bitonic sort of the 64 lane maxima across the wave (descending: lane i ends with the i-th largest)
#pragma unroll
        for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                const uint32_t olo = __shfl_xor((uint32_t)mx, stride), ohi = __shfl_xor((uint32_t)(mx >> 32), stride);
                const uint64_t o = ((uint64_t)ohi << 32) | olo;
                const bool keep_max = ((lane & stride) == 0) == ((lane & size) == 0);
                mx = keep_max ? (mx > o ? mx : o) : (mx < o ? mx : o);
            }
        }
        const uint32_t klo = __shfl((uint32_t)mx, k - 1), khi = __shfl((uint32_t)(mx >> 32), k - 1);
        const uint64_t kth = ((uint64_t)khi << 32) | klo;
        if (kth > 1ull && kth - 1ull > tau) tau = kth - 1ull;       // keys >= kth pass `key > tau`
    }
    // four 256-element tiles per round, their loads all in flight before the first one is filtered
    for (int64_t base4 = start; base4 < end; base4 += 1024) {
        float4 fv[4]; uint64_t kv[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t e0 = base4 + u * 256 + lane * 4;
            if (MODE == 0) {
                fv[u] = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
                if (e0 + 3 < end) fv[u] = *reinterpret_cast<const float4*>(sf + e0);
                else {
                    if (e0 < end) fv[u].x = sf[e0];
                    if (e0 + 1 < end) fv[u].y = sf[e0 + 1];
                    if (e0 + 2 < end) fv[u].z = sf[e0 + 2];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) kv[u][e] = (e0 + e < end) ? sk[e0 + e] : 0ull;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
        const int64_t base = base4 + u * 256;
        if (base >= end) break;
        int64_t e0 = base + lane * 4;
        uint64_t key[4];
        if (MODE == 0) {
            const float v[4] = {fv[u].x, fv[u].y, fv[u].z, fv[u].w};
#pragma unroll
            for (int e = 0; e < 4; e++) key[e] = make_key(v[e], a.idx_base + (uint32_t)(e0 + e));
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) key[e] = kv[u][e];
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            bool pass = key[e] > tau;
            uint64_t mask = __ballot(pass);
            if (mask == 0) continue;
            int np = __popcll(mask);
            if (cnt + np > BUF) {
                for (int i = cnt + lane; i < BUF; i += 64) buf[i] = 0;
                __syncthreads();
                bitonic_sort_desc(buf, BUF, lane);
                cnt = KP;
                tau = buf[k - 1] > tau ? buf[k - 1] : tau;
                __syncthreads();
                pass = key[e] > tau;
                mask = __ballot(pass);
                np = __popcll(mask);
            }
            if (pass) buf[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = key[e];
            cnt += np;
        }
        }
    }
    // final sort: the smallest power of two that holds the keys and the KP output slots (BUF is a power of two >= 2 KP)
    int nsort = KP;
    while (nsort < cnt) nsort <<= 1;
    if (nsort > BUF || (nsort & (nsort - 1))) nsort = BUF;
    for (int i = cnt + lane; i < nsort; i += 64) buf[i] = 0;
    __syncthreads();
    bitonic_sort_desc(buf, nsort, lane);
    uint64_t* o = a.out + row * a.out_row_stride + (int64_t)seg * KP;
    for (int i = lane; i < KP; i += 64) o[i] = (a.keep_last && i != KP - 1) ? 0ull : buf[i];
    if (a.zero_cnt && lane == 0 && seg == 0) a.zero_cnt[row * CCS] = 0ull;
}

// ---------------------------------------------------------------------------------------
// Radix selection for short rows (a few thousand elements: the threshold pre-pass, the filtered scans'
// candidate buffers).  k_select keeps a sorted LDS buffer and re-sorts 2*KP..512 keys whenever it fills —
// ~9 us per sort on one wave, 4-5 sorts per row.  Here the row's KP-th largest key is found exactly by an
// 8-bit MSB-first radix walk over the 64-bit keys (8 counting passes over data that sits in L1/L2), the keys
// >= it are compacted (keys are distinct: the index is part of the key) and only those KP are sorted.
// Same output contract as k_select with nseg == 1: KP keys descending, zero padded.
// ---------------------------------------------------------------------------------------
__device__ inline void bitonic_sort_desc_wg(uint64_t* buf, int n, int tid, int nt) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (n >> 1); t += nt) {
                int i = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
                int j = i + stride;
                bool desc = ((i & size) == 0);
                uint64_t x = buf[i], y = buf[j];
                if ((x < y) == desc) { buf[i] = y; buf[j] = x; }
            }
            __syncthreads();
        }
    }
}

// Top-KP (sorted descending into obuf, zero padded) of the N keys key_at(0..N-1) by a workgroup of NT threads.
// hist: 256 ints, ctl: 8 ints of LDS.  Zero keys are "no element".
// ONLY_KTH: just a threshold is wanted — obuf[0] (the only slot touched) receives a lower bound of the KP-th largest key that
// no unselected key reaches (the radix prefix; 0 when there are not more than KP keys): no compaction, no sort.
template <int NT, bool ONLY_KTH = false, class KeyAt>
__device__ inline void radix_topk_wg(KeyAt key_at, int N, int KP, uint64_t* obuf, int32_t* hist, int32_t* ctl) {
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 8) ctl[tid] = 0;          // [0] digit, [1] remaining, [2] valid count, [3] output cursor, [4] keys in the digit's bin
    if (tid == 0) { hist[0] = -1; hist[1] = 0; }      // min / max of the keys' high words (unsigned), gathered with the valid count
    if (!ONLY_KTH) for (int i = tid; i < KP; i += NT) obuf[i] = 0;
    __syncthreads();
    int myvalid = 0;
    uint32_t lmin = 0xffffffffu, lmax = 0u;
    // every pass over the keys requests EIGHT per thread before it looks at the first (round 6: a 32 768-column row was 136 dependent steps per
    // thread and pass — the K' = 2048 selection of Flat's threshold phase took 0.5 ms per 1024 rows)
    auto for_keys = [&](auto&& f) {
        for (int i0 = tid; i0 < N; i0 += 8 * NT) {
            uint64_t kk[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i = i0 + u * NT; kk[u] = i < N ? key_at(i) : 0ull; }
#pragma unroll
            for (int u = 0; u < 8; u++) f(kk[u]);
        }
    };
    for_keys([&](uint64_t key) {
        if (key != 0ull) { myvalid++; const uint32_t o = (uint32_t)(key >> 32); lmin = o < lmin ? o : lmin; lmax = o > lmax ? o : lmax; }
    });
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        myvalid += __shfl_xor(myvalid, off);
        const uint32_t a0 = __shfl_xor(lmin, off), a1 = __shfl_xor(lmax, off);
        lmin = a0 < lmin ? a0 : lmin; lmax = a1 > lmax ? a1 : lmax;
    }
    if (lane == 0 && myvalid) { atomicAdd(&ctl[2], myvalid); atomicMin(reinterpret_cast<uint32_t*>(&hist[0]), lmin); atomicMax(reinterpret_cast<uint32_t*>(&hist[1]), lmax); }
    __syncthreads();
    const int V = ctl[2];
    uint64_t kth = 1;                   // fewer than KP valid keys: take every valid one
    if (V > KP) {
        // Round 5: the walk starts at the first byte in which the keys differ.  Keys are (order word of the score) || ~index, and the scores
        // of one query's candidates share their top one or two bytes: a round over such a byte is N LDS atomics on one counter and tells nothing.
        const uint32_t gd = (uint32_t)hist[0] ^ (uint32_t)hist[1];
        const int nb = (gd >> 24) ? 0 : (gd >> 16) ? 1 : (gd >> 8) ? 2 : gd ? 3 : 4;      // common top bytes of the high words
        uint64_t prefix = nb ? ((uint64_t)((uint32_t)hist[1] & (nb == 4 ? 0xffffffffu : ~((1u << (32 - 8 * nb)) - 1u)))) << 32 : 0ull;
        if (tid == 0) ctl[1] = KP;
        __syncthreads();                // hist is zeroed by the first round: every thread has read the min / max
        for (int shift = 56 - 8 * nb; shift >= 0; shift -= 8) {
            for (int i = tid; i < 256; i += NT) hist[i] = 0;
            __syncthreads();
            for_keys([&](uint64_t key) {
                if (key != 0ull && (shift == 56 || (key >> (shift + 8)) == (prefix >> (shift + 8))))
                    atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
            });
            __syncthreads();
            if (tid < 64) {             // wave 0: 4 bins per lane, suffix sums from the top
                const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                const int sum4 = h0 + h1 + h2 + h3;
                int suf = sum4;         // inclusive suffix sum over lanes >= lane
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { int y = __shfl_down(suf, off); if (lane + off < 64) suf += y; }
                const int remaining = ctl[1];
                const uint64_t m = __ballot(suf >= remaining);
                const int L = 63 - __clzll((unsigned long long)m);      // m != 0: suf[0] = matching count >= remaining
                if (lane == L) {
                    int run = suf - sum4;                               // count in bins above this lane's
                    const int hb[4] = {h0, h1, h2, h3};
                    int d = 4 * L, rem = remaining;
                    int inbin = 0;
                    for (int b = 3; b >= 0; b--) {
                        if (run + hb[b] >= remaining) { d = 4 * L + b; rem = remaining - run; inbin = hb[b]; break; }
                        run += hb[b];
                    }
                    ctl[0] = d; ctl[1] = rem; ctl[4] = inbin;
                }
            }
            __syncthreads();
            prefix |= (uint64_t)(uint32_t)ctl[0] << shift;
            const bool all_in = ctl[1] == ctl[4];      // every key under this prefix is selected: no need to refine it
            __syncthreads();
            if (all_in) break;                          // (typically after 4 of the 8 rounds: the score bits are distinct)
        }
        kth = prefix;                                   // low bits zero after an early exit: a lower bound of those keys
    }
    if (ONLY_KTH) {     // obuf[0]: the threshold; 0 = the row holds no more than KP keys (no threshold), 1 = exactly KP valid keys
        if (tid == 0) obuf[0] = V > KP ? kth : 0ull;
        __syncthreads();
        return;
    }
    for_keys([&](uint64_t key) {
        if (key != 0ull && key >= kth) { const int pos = atomicAdd(&ctl[3], 1); if (pos < KP) obuf[pos] = key; }
    });
    __syncthreads();
    bitonic_sort_desc_wg(obuf, KP, tid, NT);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_select_radix(SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t rs_obuf[];   // [KP] output keys, then hist[256], ctl[8]
    const int KP = a.KP;
    int32_t* hist = reinterpret_cast<int32_t*>(rs_obuf + KP);
    int32_t* ctl = hist + 256;
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    if (a.row_filter && a.row_filter[row] != 1) return;
    int64_t n = a.row_n ? a.row_n[row * a.row_n_stride] : a.n_uniform;
    if (a.row_n && a.n_uniform > 0 && n > a.n_uniform) n = a.n_uniform;
    const int nin = (int)n;
    const int N = nin + (a.init ? KP : 0);      // the running state's keys are just more elements
    const float* sf = (const float*)a.in + row * a.row_stride;
    const uint64_t* sk = (const uint64_t*)a.in + row * a.row_stride;
    auto key_at = [&](int i) -> uint64_t {
        if (i >= nin) return a.init[row * KP + (i - nin)];
        return MODE == 0 ? make_key(sf[i], a.idx_base + (uint32_t)i) : sk[i];
    };
    radix_topk_wg<256>(key_at, N, KP, rs_obuf, hist, ctl);
    uint64_t* o = a.out + row * a.out_row_stride;
    for (int i = tid; i < KP; i += 256) o[i] = (a.keep_last && i != KP - 1) ? 0ull : rs_obuf[i];
    if (a.zero_cnt && tid == 0) a.zero_cnt[row * CCS] = 0ull;
}

// ---------------------------------------------------------------------------------------
// Candidate gather + selection for the rotated fast scan (PQGatherArgs, rsx_internal.h).  k_pq_scan_rot leaves a query's survivors
// in (item, wave, slot) runs of its waves' logs; k_pq_rot_compact appended them to the query's candidate row with one returning atomic per
// (item, query) — ~40 dependent device-scope atomics on ONE counter per query, 40-50 us per batch whatever else the kernel did —
// and k_select_radix then read the row back.  Here the query's workgroup finds its own segments (qitems: the inverse of the item
// order, written by k_pq_rot_items), sums their counts, copies the keys behind what the pre-pass emitted and selects the K'
// largest from LDS.  Key order inside the row is free (keys are distinct; every consumer sorts or re-scores).
// ---------------------------------------------------------------------------------------
constexpr int GS_LCAP = 2048;      // row lengths up to this are selected from LDS; longer rows are read back from the candidate row
#ifdef RSX_MEASURE
__device__ uint64_t g_gs_trace[256 * 8];          // tools/ builds only: [workgroup < 256][mark] wall clock (10 ns ticks) of k_pq_gather_select's phases
extern "C" int rsx_debug_gs_trace(uint64_t* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gs_trace), sizeof(uint64_t) * 256 * 8) == hipSuccess ? 0 : -1; }
#define GS_MARK(i) do { if (blockIdx.x < 256 && threadIdx.x == 0) g_gs_trace[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define GS_MARK(i)
#endif
__global__ __launch_bounds__(256) void k_pq_gather_select(PQGatherArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t gs_smem[];
    const int KP = a.KP, NP = a.nprobe * a.tmax, NS = NP * 16;
    uint64_t* obuf = gs_smem;                                      // [KP]
    uint64_t* lkeys = obuf + KP;                                   // [GS_LCAP]
    int32_t* hist = reinterpret_cast<int32_t*>(lkeys + GS_LCAP);   // [256]
    int32_t* ctl = hist + 256;                                     // [8]
    int32_t* misc = ctl + 8;                                       // [8]: 0 overflow flag, 1..4 wave totals
    int32_t* slots = misc + 8;                                     // [NP]: item * 4 + slot, -1 = no such tile
    uint32_t* sfirst = reinterpret_cast<uint32_t*>(slots + NP);    // [NS]: first key of the run in the log pool
    uint32_t* spre = sfirst + NS;                                  // [NS]: keys in the runs before this one (exclusive prefix)
    uint16_t* cnts = reinterpret_cast<uint16_t*>(spre + NS);       // [NS]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t q = blockIdx.x;
    if (tid < 8) misc[tid] = 0;
    GS_MARK(0);
    __syncthreads();
    // 1. the query's segments and their counts: three short phases, every load of a phase independent of the others
    //    (a) tiles per probed list, (b) the work item + slot of every (probe rank, tile), (c) the 16 wave counters of each
    int32_t* ntl = hist;                                           // [nprobe] (hist is free until the selection)
    for (int j = tid; j < a.nprobe; j += 256) {
        const int32_t l = a.probe_list[q * a.nprobe + j];
        ntl[j] = l >= 0 ? (int)((a.list_len[l] + a.tile_rows - 1) / a.tile_rows) : 0;
    }
    __syncthreads();
    for (int p = tid; p < NP; p += 256) {
        const int j = p / a.tmax, t = p - j * a.tmax;
        slots[p] = t < ntl[j] ? a.qitems[(q * a.nprobe + j) * a.tmax + t] : -1;
    }
    __syncthreads();
    {
        bool over = false;
        for (int s0 = tid; s0 < NS; s0 += 256 * 8) {
            uint32_t c[8], f[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int s2 = s0 + u * 256;
                const int32_t slot = s2 < NS ? slots[s2 >> 4] : -1;
                const uint2 dsc = slot >= 0 ? a.seg_desc[(size_t)(slot >> 2) * 64 + (size_t)(s2 & 15) * 4 + (slot & 3)] : make_uint2(0u, 0u);
                f[u] = dsc.x; c[u] = dsc.y;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int s2 = s0 + u * 256;
                if (s2 >= NS) break;
                if (c[u] >> 31) { over = true; c[u] &= 0x7fffffffu; }         // the wave's log was full: keys were dropped
                cnts[s2] = (uint16_t)c[u]; sfirst[s2] = f[u];
            }
        }
        if (over) misc[0] = 1;
    }
    const unsigned long long e0 = a.cand_cnt[q * CCS];             // keys the pre-pass emitted (row prefix)
    __syncthreads();
    GS_MARK(1);
    // 2. exclusive prefix of the run lengths (thread t owns the runs [t spt, (t + 1) spt); wave scan + one LDS hop)
    const int spt = (NS + 255) >> 8;
    int mine = 0;
    for (int u = 0; u < spt; u++) { const int s2 = tid * spt + u; if (s2 < NS) mine += cnts[s2]; }
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(incl, off); if (lane >= off) incl += y; }
    if (lane == 63) misc[1 + w] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int ww = 0; ww < 4; ww++) { const int v = misc[1 + ww]; if (ww < w) wbase += v; total += v; }
    {
        uint32_t run = (uint32_t)(wbase + incl - mine);
        for (int u = 0; u < spt; u++) { const int s2 = tid * spt + u; if (s2 < NS) { spre[s2] = run; run += cnts[s2]; } }
    }
    const bool over = misc[0] != 0;
    const unsigned long long e0c = e0 < (unsigned long long)a.cand_cap ? e0 : (unsigned long long)a.cand_cap;
    // the row's count: past the capacity when a segment dropped keys (k_finalize then flags the query), exactly like the compaction's
    if (tid == 0) a.cand_cnt[q * CCS] = e0 + (unsigned long long)total + (over ? (unsigned long long)a.cand_cap + 1ull : 0ull);
    const unsigned long long nrow_ = e0c + (unsigned long long)total;
    const int nrow = (int)(nrow_ < (unsigned long long)a.cand_cap ? nrow_ : (unsigned long long)a.cand_cap);   // keys that fit the row
    const bool in_lds = KP > 0 && nrow_ <= (unsigned long long)GS_LCAP;
    __syncthreads();        // spre is complete
    GS_MARK(2);
    // 3. copy, one KEY per thread and step (round 5): key p of the concatenated runs sits in the run found by bisection of the prefix
    // array — every load is independent of every other, where a thread used to walk its own runs one after the other (descriptor,
    // then keys: two dependent round trips per run, 20 of the kernel's 42 us per query in the phase trace)
    uint64_t* row = a.cand + q * a.cand_cap;
    if (in_lds) for (int i2 = tid; i2 < (int)e0c; i2 += 256) lkeys[i2] = row[i2];
    for (int p0 = tid; p0 < total; p0 += 256 * 4) {
        uint64_t kk[4]; int pp[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int pk = p0 + u * 256;
            pp[u] = -1; kk[u] = 0ull;
            if (pk >= total) continue;
            int lo = 0, hi = NS;         // largest run index with spre <= pk among the non-empty ones: spre is non-decreasing
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (spre[mid] <= (uint32_t)pk) lo = mid; else hi = mid; }
            kk[u] = a.log_keys[(size_t)sfirst[lo] + (size_t)((uint32_t)pk - spre[lo])];
            pp[u] = (int)e0c + pk;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (pp[u] < 0) continue;
            if (pp[u] < a.cand_cap) row[pp[u]] = kk[u];
            if (in_lds) lkeys[pp[u]] = kk[u];
        }
    }
    if (KP == 0) return;    // gather only: k_pq_final_tab works on the whole row
    __syncthreads();        // workgroup-scope release / acquire: the row (and lkeys) written above are visible to every thread
    GS_MARK(3);
    // 4. the K' largest keys, sorted, to the state row
    if (in_lds) {
        auto key_at = [&](int i2) -> uint64_t { return lkeys[i2]; };
        radix_topk_wg<256>(key_at, nrow, KP, obuf, hist, ctl);
    } else {
        auto key_at = [&](int i2) -> uint64_t { return row[i2]; };
        radix_topk_wg<256>(key_at, nrow, KP, obuf, hist, ctl);
    }
    GS_MARK(4);
    uint64_t* o = a.state + q * KP;
    for (int i2 = tid; i2 < KP; i2 += 256) o[i2] = obuf[i2];
    GS_MARK(5);
}

static size_t gather_select_lds(int nprobe, int tmax, int KP) {
    return (size_t)KP * 8 + (size_t)GS_LCAP * 8 + (256 + 8 + 8) * 4 + (size_t)nprobe * tmax * 4 + (size_t)nprobe * tmax * 16 * (2 + 8) + 64;
}
bool pq_gather_select_applies(int nprobe, int tmax, int KP) {
    return tmax >= 1 && tmax <= 16 && nprobe <= 256 && (int64_t)nprobe * tmax * 16 <= 16384 && KP <= 4096 && gather_select_lds(nprobe, tmax, KP) <= 150 * 1024;
    // (150 KiB: round 5's prefix arrays took the per-run LDS from 2 to 10 bytes, and under the old 96 KiB cap shapes like nprobe 128 x 4 tiles
    //  silently went back to the compaction + selection launches — ADVICE r5; a declined shape is counted: rsx_get_timing "pq_gather_declined")
}
void launch_pq_gather_select(const PQGatherArgs& a, int64_t nq, hipStream_t st) {
    if (nq <= 0) return;
    const size_t shm = gather_select_lds(a.nprobe, a.tmax, a.KP);
    static DevSize attr;
    attr.grow(shm, [&] { (void)hipFuncSetAttribute((const void*)k_pq_gather_select, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); });
    hipLaunchKernelGGL(k_pq_gather_select, dim3((unsigned)nq), dim3(256), shm, st, a);
}

// ---------------------------------------------------------------------------------------
// IVF-PQ threshold pre-pass in ONE launch: a workgroup per query scores the first pre_rows vectors of the
// query's closest NON-EMPTY probed list with the query's 8-bit table (24 KiB in LDS, byte gathers; the SAME integer sums and
// the same fp32 expression as k_pq_scan8, hence the same keys), selects their K'-th largest key in LDS and
// writes it as the query's threshold (state[q][KP-1], the other slots zero) and resets the candidate counter.
// Replaces pair grouping + k_pq_scan8<unfiltered> + selection (5 launches) for that step.
// ---------------------------------------------------------------------------------------
constexpr int PP_MAXSEG = 8;      // lists a threshold sample may span (k_pq_prepass)
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_pq_prepass(PQPrepassArgs a) {   // <= 64 VGPRs: two workgroups per CU
    extern __shared__ __attribute__((aligned(16))) uint64_t pp_smem[];
    uint64_t* obuf = pp_smem;                                        // [2]
    int32_t* hist = reinterpret_cast<int32_t*>(obuf + 2);            // [256]
    int32_t* ctl = hist + 256;                                       // [8]
    uint8_t* tab = reinterpret_cast<uint8_t*>(ctl + 8);              // [Mpad][256]
    uint16_t* sums = reinterpret_cast<uint16_t*>(tab + (size_t)a.Mpad * 256);   // [pre_rows]: integer table sum + 1 (0 = no vector)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t q = blockIdx.x;
    // the closest probed list that holds vectors HERE (in a list-sharded index most probes hit other ranks' lists)
    int32_t l = -1; int64_t len = 0; int j0 = 0;
    for (int j = 0; j < a.nprobe; j++) {
        const int32_t lj = a.probe_list[q * a.nprobe + j];
        if (lj >= 0 && a.list_len[lj] > 0) { l = lj; len = a.list_len[lj]; j0 = j; break; }
    }
    const int n = (int)(len < a.pre_rows ? len : a.pre_rows);
    const int nslab = (n + 63) >> 6;
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.lut8 + q * a.Mpad * 256);
        uint4* dst = reinterpret_cast<uint4*>(tab);
        for (int i = tid; i < a.Mpad * 16; i += 1024) dst[i] = src[i];
    }
    __syncthreads();
    const float scale = a.qparam[q * 4 + 0], bias = a.qparam[q * 4 + 1];
    const float dis0 = a.probe_dis0[q * a.nprobe + j0];
    const int64_t col = a.seg_start[q * (a.nprobe + 1) + j0];
    const int nch = a.Mpad >> 4;
    if (a.CB == 0 && a.Mpad == 16) {
        // rotated layout, M = 16: 64-vector blocks, a lane owns a vector's 16 bytes (byte s: m = (lane + s) & 15); lut8 is [q][code][16]
        const uint8_t* lb = a.codes + ((l >= 0 ? a.list_base[l] : 0) >> 6) * (int64_t)1024;
        for (int sl = w; sl < nslab; sl += 16) {
            const uint4 c = *reinterpret_cast<const uint4*>(lb + (int64_t)sl * 1024 + lane * 16);
            const uint32_t wds[4] = {c.x, c.y, c.z, c.w};
            uint32_t acc = 0;
#pragma unroll
            for (int s2 = 0; s2 < 16; s2++) acc += tab[((wds[s2 >> 2] >> (8 * (s2 & 3))) & 255u) * 16 + ((lane + s2) & 15)];
            const int64_t pos = (int64_t)sl * 64 + lane;
            sums[pos] = (pos < len) ? (uint16_t)(acc + 1u) : (uint16_t)0;
        }
    } else if (a.CB == 0) {
        // rotated layout: lut8 is [q][code][M]; lane (g, i) of a wave sums the entries of its 16 (+8) bytes of
        // vector i of a 16-vector block, the four lanes of a vector are added with two shuffles
        const int M = a.Mpad, NF = M >> 6, NH = (M >> 5) & 1;
        const int g = lane >> 4, i = lane & 15;
        const int nblk = nslab * 4;
        const uint8_t* lbase = a.codes + ((l >= 0 ? a.list_base[l] : 0) >> 4) * (int64_t)(16 * M);
        const int wu = __builtin_amdgcn_readfirstlane(w);
        // one block ahead: the next block's code loads are issued before this block's gathers (the loop used to pay one HBM
        // round trip per block, 8 in a row for the 2048-row sample).  Deeper batches were measured and rejected: two or four
        // blocks in flight need 96 / 116 VGPRs, which halves the workgroups per CU (86 -> 107 us).
        auto fetch = [&](int b, uint4 (&cf)[2], uint2& ch) {
            const uint8_t* bp = lbase + (int64_t)b * (16 * M);
#pragma unroll
            for (int p = 0; p < 2; p++) cf[p] = p < NF ? *reinterpret_cast<const uint4*>(bp + p * 1024 + lane * 16) : make_uint4(0u, 0u, 0u, 0u);
            ch = NH ? *reinterpret_cast<const uint2*>(bp + NF * 1024 + lane * 8) : make_uint2(0u, 0u);
        };
        uint4 cf[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)}, nf[2]; uint2 ch = make_uint2(0u, 0u), nh;
        if (wu < nblk) fetch(wu, cf, ch);
#pragma unroll 1
        for (int b = wu; b < nblk; b += 16) {
            fetch(b + 16 < nblk ? b + 16 : b, nf, nh);
            uint32_t acc = 0;
#pragma unroll 1
            for (int p = 0; p < NF; p++) {      // rolled (one copy of the 16 gathers): unrolled, the kernel needs 90 VGPRs = one workgroup per CU
                const uint4 c4 = p == 0 ? cf[0] : cf[1];
                const uint32_t wds[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                for (int s2 = 0; s2 < 16; s2++) {
                    const int m = 64 * p + 16 * g + ((i + s2) & 15);
                    acc += tab[((wds[s2 >> 2] >> (8 * (s2 & 3))) & 255u) * M + m];
                }
            }
            if (NH) {
                const uint32_t wds[2] = {ch.x, ch.y};
#pragma unroll
                for (int s2 = 0; s2 < 8; s2++) {
                    const int m = 64 * NF + 16 * (g & 1) + ((i + s2 + 8 * (g >> 1)) & 15);
                    acc += tab[((wds[s2 >> 2] >> (8 * (s2 & 3))) & 255u) * M + m];
                }
            }
            acc += __shfl_xor(acc, 16);
            acc += __shfl_xor(acc, 32);
            if (g == 0) {
                const int64_t pos = (int64_t)b * 16 + i;
                sums[pos] = (pos < len) ? (uint16_t)(acc + 1u) : (uint16_t)0;
            }
            cf[0] = nf[0]; cf[1] = nf[1]; ch = nh;
        }
    } else if (a.CB == PQ_SLICED) {
        // sliced layout (rsx_internal.h): lut8 is [q][slice][code][32]; lane (g, i) of a wave sums the entries of its 16 bytes per slice of
        // vector 16 (g >> 1) + i of a 32-vector block, the two lanes of a vector are added with one shuffle; one block ahead
        const int M = a.Mpad, NS = M >> 5;
        const int g = lane >> 4, i = lane & 15;
        const int nblk = nslab * 2;
        const uint8_t* lbase = a.codes + ((l >= 0 ? a.list_base[l] : 0) >> 5) * (int64_t)(32 * M);
        const int wu = __builtin_amdgcn_readfirstlane(w);
        uint4 cf[4], nf[4];
        auto fetch = [&](int b, uint4 (&c)[4]) {
            const uint8_t* bp = lbase + pq_sliced_off(b, 0, M) + lane * 16;           // the list starts on a group boundary
#pragma unroll
            for (int sl = 0; sl < 4; sl++) c[sl] = sl < NS ? *reinterpret_cast<const uint4*>(bp + sl * (PQ_SLICED_GB * 1024)) : make_uint4(0u, 0u, 0u, 0u);
        };
        if (wu < nblk) fetch(wu, cf);
#pragma unroll 1
        for (int b = wu; b < nblk; b += 16) {
            fetch(b + 16 < nblk ? b + 16 : b, nf);
            uint32_t acc = 0;
#pragma unroll 1
            for (int sl = 0; sl < NS; sl++) {
                const uint4 c4 = sl == 0 ? cf[0] : sl == 1 ? cf[1] : sl == 2 ? cf[2] : cf[3];
                const uint32_t wds[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                for (int s2 = 0; s2 < 16; s2++) {
                    const int m = 32 * sl + 16 * (g & 1) + ((i + s2) & 15);
                    acc += tab[((sl * 256 + ((wds[s2 >> 2] >> (8 * (s2 & 3))) & 255u)) << 5) + (m & 31)];       // lut8: [slice][code][32]
                }
            }
            acc += __shfl_xor(acc, 16);
            if ((g & 1) == 0) {
                const int64_t pos = (int64_t)b * 32 + 16 * (g >> 1) + i;
                sums[pos] = (pos < len) ? (uint16_t)(acc + 1u) : (uint16_t)0;
            }
#pragma unroll
            for (int sl = 0; sl < 4; sl++) cf[sl] = nf[sl];
        }
    } else
    for (int s = w; s < nslab; s += 16) {
        const uint8_t* sp = a.codes + ((a.list_base[l] >> 6) + s) * (int64_t)(64 * a.Mpad) + lane * 16;
        uint32_t acc = 0;
        for (int g = 0; g < nch; g++) {
            const uint4 c = *reinterpret_cast<const uint4*>(sp + g * 1024);
            const uint32_t wds[4] = {c.x, c.y, c.z, c.w};
            const uint8_t* tg = tab + g * 16 * 256;
#pragma unroll
            for (int b = 0; b < 16; b++) acc += tg[b * 256 + ((wds[b >> 2] >> (8 * (b & 3))) & 255u)];
        }
        const int64_t pos = (int64_t)s * 64 + lane;
        sums[pos] = (pos < len) ? (uint16_t)(acc + 1u) : (uint16_t)0;
    }
    __syncthreads();
    // Round 4 — a closest list SHORTER than 8 k vectors: the sample continues in the next closest lists (at most PP_MAXSEG - 1 more,
    // until it holds min(pre_rows, 8 k) rows).  A query whose closest list held fewer than k vectors used to get no threshold at
    // all, kept every vector of every probed list (781 k survivors on the bench index, 6.6 M at nprobe 512) and was re-run
    // exactly.  A vector of list j enters with the integer sum S + floor((dis0_j - dis0_0) / scale) - 2: dis0_0 + fma(scale, that,
    // bias) is a LOWER bound of its approximate score (vectors whose shifted sum would be negative are left out), so the k-th
    // largest of the sample's sums still yields a valid a_k — the sample holds k vectors whose approximate scores reach it.
    // Rare, so simple: one thread per vector, byte loads through pq_code_addr.
    int ntot = nslab * 64;
    {
        int32_t* seg = reinterpret_cast<int32_t*>(sums + a.pre_rows);      // [PP_MAXSEG][4]: list, rows, first sample slot, shift; then the count
        const int want = a.pre_rows < 8 * a.k ? a.pre_rows : 8 * a.k;
        if (tid == 0) {
            int ns = 0, off = ntot;
            if (l >= 0 && scale > 0.0f)
                for (int j = j0 + 1; j < a.nprobe && ns < PP_MAXSEG - 1 && off < want; j++) {
                    const int32_t lj = a.probe_list[q * a.nprobe + j];
                    if (lj < 0) continue;
                    const int64_t len_j = a.list_len[lj];
                    if (len_j <= 0) continue;
                    const float sh = floorf((a.probe_dis0[q * a.nprobe + j] - dis0) / scale) - 2.0f;
                    if (!(sh > -1.0e9f)) break;
                    const int rows = (int)(len_j < (int64_t)(a.pre_rows - off) ? len_j : (int64_t)(a.pre_rows - off));
                    seg[4 * ns + 0] = lj; seg[4 * ns + 1] = rows; seg[4 * ns + 2] = off; seg[4 * ns + 3] = sh < 0.0f ? (int)sh : 0;
                    off += (rows + 63) & ~63;
                    ns++;
                }
            seg[4 * PP_MAXSEG] = ns; seg[4 * PP_MAXSEG + 1] = off;
        }
        __syncthreads();
        const int nseg = seg[4 * PP_MAXSEG];
        for (int sg = 0; sg < nseg; sg++) {
            const int32_t ls = seg[4 * sg];
            const int rows = seg[4 * sg + 1], off = seg[4 * sg + 2], shift = seg[4 * sg + 3];
            const int64_t row0 = a.list_base[ls];
            for (int pos = tid; pos < ((rows + 63) & ~63); pos += 1024) {
                int v = -1;
                if (pos < rows) {
                    uint32_t acc = 0;
                    for (int m = 0; m < a.Mpad; m++) {
                        const uint32_t code = a.codes[pq_code_addr(row0 + pos, m, a.Mpad, a.CB)];
                        acc += tab[pq_lut8_index(0, (int)code, m, a.Mpad, a.CB == PQ_SLICED ? 2 : a.CB == 0 ? 1 : 0)];
                    }
                    v = (int)acc + shift;
                }
                sums[off + pos] = v >= 0 ? (uint16_t)(v + 1) : (uint16_t)0;
            }
        }
        ntot = seg[4 * PP_MAXSEG + 1];
    }
    __syncthreads();
    // The approximate score dis0 + fma(scale, S, bias) is monotone in the integer sum S: the k-th best score of the sample
    // is the score of the k-th largest S (a 16-bit radix walk).  Round 3 — threshold by CONSTRUCTION instead of by the K'-th
    // sample key: the sample holds k vectors with approximate score >= a_k, hence exact score >= a_k - eps, so the query's
    // exact k-th best score is >= a_k - eps; a vector with approximate score < tau = a_k - 2 eps has exact score
    // < a_k - eps and cannot be among the top k.  Every vector the scan drops is therefore provably irrelevant, for any k,
    // and the candidate count is what the data needs (measured: ~130 for k = 10, ~2300 for k = 1000) instead of a guess.
    auto key_at = [&](int i) -> uint64_t { return (uint64_t)sums[i] << 48; };
    radix_topk_wg<1024, true>(key_at, ntot, a.k, obuf, hist, ctl);
    uint64_t* o = a.state + q * a.KP;
    for (int i = tid; i < a.KP; i += 1024) o[i] = 0ull;              // the candidate merge starts from an empty state
    if (tid == 0) {
        uint64_t tau = 0ull;                                          // fewer than k sample vectors: no threshold
        const uint64_t kth = obuf[0];
        if (kth != 0ull) {
            const float eps = a.qparam[q * 4 + 2];
            const float a_k = dis0 + __fmaf_rn(scale, (float)((int)(kth >> 48) - 1), bias);   // kth: a lower bound of the k-th largest (S + 1)
            float t = __fmaf_rn(-2.0002f, eps, a_k);
            t -= fabsf(t) * 4.8e-7f + 1e-30f;                         // two ulps down: the subtraction's own rounding
            tau = make_key(t, 0xFFFFFFFFu);                           // low word 0: every key with score >= t compares greater
        }
        a.tau[q] = tau;
        // integer form of the threshold (the score is monotone in S): smallest S whose score reaches it
        int sthr = 0;
        if (tau != 0ull) {
            const float ts = key_score(tau);
            int b0 = 0, b1 = 255 * a.Mpad + 1;
            while (b0 < b1) { const int mid = (b0 + b1) >> 1; if (dis0 + __fmaf_rn(scale, (float)mid, bias) >= ts) b1 = mid; else b0 = mid + 1; }
            sthr = b0;
        }
        ctl[5] = sthr; ctl[6] = 0;
    }
    __syncthreads();
    // Emission (round 3): when the sample covers the whole first scan tile of this list, the sample's own candidates leave
    // from HERE — one workgroup per query, no contention — and the scan is told to drop this (query, list, tile 0): for
    // large k almost every candidate of a query sits in its closest list, far more than a scan wave's survivor segment holds.
    const int64_t tile0 = len < a.tile_rows ? len : a.tile_rows;
    const bool emit = a.cand != nullptr && l >= 0 && tile0 <= (int64_t)n;
    if (emit) {
        const int sthr = ctl[5];
        uint64_t* dst = a.cand + q * a.cand_cap;
        const int ne = (int)tile0;              // exactly the rows of tile 0: the scan still covers the list's other tiles
        for (int i0 = 0; i0 < ne; i0 += 1024) {
            const int i = i0 + tid;
            const int sv = i < ne ? (int)sums[i] : 0;
            const bool pass = sv != 0 && sv - 1 >= sthr;
            const uint64_t m = __builtin_amdgcn_ballot_w64(pass);
            int base = 0;
            if (lane == 0 && m) base = atomicAdd(&ctl[6], __builtin_popcountll(m));
            base = __shfl(base, 0);
            if (pass) {
                const int pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (pos < a.cand_cap) dst[pos] = make_key(dis0 + __fmaf_rn(scale, (float)(sv - 1), bias), (uint32_t)(col + i));
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.cand_cnt[q * CCS] = emit ? (unsigned long long)ctl[6] : 0ull;
        if (a.excl) a.excl[q] = (uint16_t)(emit ? (0x8000 | j0) : 0);
    }
}
void launch_pq_prepass(const PQPrepassArgs& a, int64_t nq, hipStream_t st) {
    if (nq <= 0) return;
    size_t shm = 16 + 264 * 4 + (size_t)a.Mpad * 256 + (size_t)a.pre_rows * 2 + (4 * PP_MAXSEG + 4) * 4 + 64;
    static DevSize attr;
    attr.grow(shm, [&] { hipFuncSetAttribute((const void*)k_pq_prepass, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); });
    hipLaunchKernelGGL(k_pq_prepass, dim3((unsigned)nq), dim3(1024), shm, st, a);
}

// rows short enough for the radix form (single segment; the counting passes re-read the row 9 times from L1/L2)
static bool select_radix_applies(const SelectArgs& a) {
    static const int off = measure_env("RSX_SELECT_V1", 0);
    // measured: wins for the threshold pre-pass (2048 scores -> K' = 128) and the candidate merge, loses for the
    // probe selection (4096 scores -> 32 keys: k_select's buffer hardly ever needs a second sort there)
    // ... unless only a few rows are in flight (latency path): 256 threads per row instead of one wave
    // ... or many keys are kept (nprobe 512 of 8192 lists: one wave re-sorted its 1024-key buffer for 1.07 ms per 1024 rows)
    return !off && (a.keep_last || a.in_is_keys || a.nrows <= 16 || a.KP >= 256) && a.nseg == 1 && a.seg_base == 0 && !a.tau_ptr && a.n_uniform > 0 &&
           (a.n_uniform <= 16384 || (a.in_is_keys && a.row_n && a.n_uniform <= 131072) || (a.KP >= 1024 && a.n_uniform <= 131072)) && a.KP <= 4096;
    // (K' >= 1024: one wave per 32 k-score segment re-sorts a 4096-key buffer over and over — 3 ms per 1024 x 65536 scores; the radix walk
    //  reads the row eight times from L2 instead)
}

void launch_select(const SelectArgs& a, hipStream_t st) {
    if (a.nrows <= 0 || a.nseg - a.seg_base <= 0) return;
    if (select_radix_applies(a)) {
        size_t shm2 = (size_t)a.KP * 8 + 264 * 4;
        if (a.in_is_keys) {
            if (shm2 > 48 * 1024) hipFuncSetAttribute((const void*)k_select_radix<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2);
            hipLaunchKernelGGL(k_select_radix<1>, dim3((unsigned)a.nrows), dim3(256), shm2, st, a);
        } else {
            if (shm2 > 48 * 1024) hipFuncSetAttribute((const void*)k_select_radix<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2);
            hipLaunchKernelGGL(k_select_radix<0>, dim3((unsigned)a.nrows), dim3(256), shm2, st, a);
        }
        return;
    }
    dim3 grid((unsigned)a.nrows, (unsigned)(a.nseg - a.seg_base));
    size_t shm = (size_t)a.BUF * sizeof(uint64_t);
    if (a.in_is_keys) {
        if (shm > 48 * 1024) hipFuncSetAttribute((const void*)k_select<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(k_select<1>, grid, dim3(64), shm, st, a);
    } else {
        if (shm > 48 * 1024) hipFuncSetAttribute((const void*)k_select<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(k_select<0>, grid, dim3(64), shm, st, a);
    }
}

// ---------------------------------------------------------------------------------------
// Probe set-up: decode the top-nprobe coarse keys of each query into list numbers and coarse
// scores (dis0), and lay the query's probed lists end to end in its row of the score buffer:
// seg_start[q][j] = first column of list j's scores, seg_start[q][nprobe] = row length.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_probe_setup(const uint64_t* keys, int KPp, int64_t nq, int nprobe,
                                                    const int64_t* list_len, int pad_to, int32_t* probe_list,
                                                    float* dis0, int64_t* seg_start) {
    extern __shared__ __attribute__((aligned(16))) uint64_t ps_buf[];
    int64_t* lens = (int64_t*)ps_buf;
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    for (int j = lane; j < nprobe; j += 64) {
        uint64_t key = keys[q * KPp + j];
        int32_t l = key ? (int32_t)key_idx(key) : -1;
        probe_list[q * nprobe + j] = l;
        dis0[q * nprobe + j] = key ? key_score(key) : -__builtin_inff();
        int64_t len = (l >= 0) ? list_len[l] : 0;
        lens[j] = (len + pad_to - 1) / pad_to * pad_to;
    }
    __syncthreads();
    if (lane == 0) {
        int64_t off = 0;
        int64_t* ss = seg_start + q * (nprobe + 1);
        for (int j = 0; j < nprobe; j++) { ss[j] = off; off += lens[j]; }
        ss[nprobe] = off;
    }
}

void launch_probe_setup(const uint64_t* probe_keys, int KPp, int64_t nq, int nprobe, const int64_t* list_len,
                        int pad_to, int32_t* probe_list, float* probe_dis0, int64_t* seg_start, hipStream_t st) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_probe_setup, dim3((unsigned)nq), dim3(64), (size_t)nprobe * 8, st, probe_keys, KPp, nq,
                       nprobe, list_len, pad_to, probe_list, probe_dis0, seg_start);
}

// ---------------------------------------------------------------------------------------
// Probe selection of the fast coarse quantiser (round 6): one workgroup per query turns the APPROXIMATE centroid scores of
// launch_coarse_approx into the EXACT top-nprobe — the lists, their exact scores (the oracle's `s = fmaf(x[t], c[t], s)` chain, what
// k_gemm_exact computes for all 4096) and the probe set-up — in one launch:
//   1. the nprobe-th largest approximate score a_n (a bit-by-bit walk over the row's order-preserving keys, held in registers);
//   2. candidates = every centroid with approximate score >= a_n - 2 e, e = a bound on |approximate - exact| for ANY centroid of this
//      query (fp16 rounding of both operands, fp32 accumulation, Cauchy-Schwarz with the largest centroid norm).  A centroid below that
//      line is beaten by nprobe others for certain, so the true top-nprobe — ties included — lies among the candidates;
//   3. the exact chain of each candidate (one lane each, its centroid row streamed with 16-byte loads), keys (score, list), bitonic sort;
//   4. probe lists, coarse scores and the query's row layout (what k_probe_setup writes).
// More candidates than the row holds (nprobe + 16, rounded up to 16s: mass ties, a zero query) or a non-finite score anywhere: bad[q] = 2 — the query is re-run by the exact path
// with the certificate's other failures (rerun_uncertified); the probes written for it are still valid lists.
// ---------------------------------------------------------------------------------------
#ifdef RSX_MEASURE
__device__ uint64_t g_cp_trace[256 * 8];          // tools/ builds only: [workgroup < 256][mark] wall clock (10 ns ticks) of k_coarse_pick's phases
extern "C" int rsx_debug_cp_trace(uint64_t* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cp_trace), sizeof(uint64_t) * 256 * 8) == hipSuccess ? 0 : -1; }
#define CP_MARK(i) do { if (blockIdx.x < 256 && threadIdx.x == 0) g_cp_trace[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define CP_MARK(i) do { } while (0)
#endif
template <int KPT, int NLD>      // keys per thread (a multiple of 4): nlist <= 256 KPT; 16-byte loads per thread and row chunk: cmax_rt <= 8 NLD
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_coarse_pick(CoarsePickArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cp_lds[];
    __shared__ uint64_t skey[CP_CMAX], sorted[CP_CMAX];
    __shared__ int32_t cand[CP_CMAX];
    __shared__ int64_t lens[CP_MAXPROBE];
    __shared__ uint32_t wcnt[2][4][4];
    __shared__ float red[4];
    __shared__ int s_cnt, s_bad;
    // workgroups behind the queries': pass 0 of the IVF-PQ table build (it needs the queries only; this kernel is a latency chain that leaves
    // the CUs' issue slots to them)
    if (blockIdx.x >= (unsigned)a.nq) { pq_lut_pass0_block(a.lp0, blockIdx.x - (unsigned)a.nq); return; }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t q = blockIdx.x;
    const int nlist = a.nlist, d = a.d, dp = (d + CP_CH - 1) / CP_CH * CP_CH;
    float* xq = reinterpret_cast<float*>(cp_lds);              // [dp]: the query, zero padded to whole chunks
    float* stage = xq + dp;                                    // [cmax_rt][CP_RS]
    if (tid == 0) { s_cnt = 0; s_bad = 0; }
    CP_MARK(0);
    float ss = 0.0f;
    for (int t = tid; t < dp; t += 256) { const float v = t < d ? a.Q32[q * a.ld + t] : 0.0f; xq[t] = v; ss = __fmaf_rn(v, v, ss); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if (lane == 0) red[w] = ss;
    // the row's order-preserving keys stay in registers: keys 4 u .. 4 u + 3 of thread tid = lists 4 (tid + 256 u) .. + 3 (0 = no list: below
    // every real key)
    bool badl = false;
    uint32_t rk[KPT];
    const float4* arow = reinterpret_cast<const float4*>(a.approx + q * a.nlp);          // nlp % 4 == 0
#pragma unroll
    for (int u = 0; u < KPT / 4; u++) {
        const int i = 4 * (tid + 256 * u);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nlist) v = arow[tid + 256 * u];          // (the row's tail beyond nlist, < 4 floats, is inside the row's stride)
        const float sv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j2 = 0; j2 < 4; j2++) {
            rk[4 * u + j2] = 0u;
            if (i + j2 < nlist) {
                if (!(fabsf(sv[j2]) < __builtin_inff())) badl = true;
                rk[4 * u + j2] = f2ord(sv[j2] + 0.0f);
            }
        }
    }
    __syncthreads();
    CP_MARK(1);
    const float xn = sqrtf(red[0] + red[1] + red[2] + red[3]);
    // 1. a lower bound of the nprobe-th largest key, within 2^-15 of it: its top 24 bits, two per step — the largest v (low 8 bits clear) with
    // |{keys >= v}| >= nprobe.  Counting is ballots and a few words exchanged per step — a radix histogram of 4096 keys that share their leading
    // bytes is thousands of LDS atomics on two or three counters.  (A bound is enough: the candidate line below only moves down with it.)
    uint32_t prefix = 0u;
    for (int bit = 30, it = 0; bit >= 8; bit -= 2, it++) {
        uint32_t c1 = 0u, c2 = 0u, c3 = 0u;
        const uint32_t v1 = prefix | (1u << bit), v2 = prefix | (2u << bit), v3 = prefix | (3u << bit);
#pragma unroll
        for (int u = 0; u < KPT; u++) {
            c1 += (uint32_t)__popcll(__ballot(rk[u] >= v1)); c2 += (uint32_t)__popcll(__ballot(rk[u] >= v2)); c3 += (uint32_t)__popcll(__ballot(rk[u] >= v3));
        }
        if (lane == 0) { wcnt[it & 1][w][0] = c1; wcnt[it & 1][w][1] = c2; wcnt[it & 1][w][2] = c3; }
        __syncthreads();
        const uint32_t t1 = wcnt[it & 1][0][0] + wcnt[it & 1][1][0] + wcnt[it & 1][2][0] + wcnt[it & 1][3][0];
        const uint32_t t2 = wcnt[it & 1][0][1] + wcnt[it & 1][1][1] + wcnt[it & 1][2][1] + wcnt[it & 1][3][1];
        const uint32_t t3 = wcnt[it & 1][0][2] + wcnt[it & 1][1][2] + wcnt[it & 1][2][2] + wcnt[it & 1][3][2];
        const uint32_t np = (uint32_t)a.nprobe;
        prefix = t3 >= np ? v3 : t2 >= np ? v2 : t1 >= np ? v1 : prefix;
    }
    // 2. candidates
    CP_MARK(2);
    const float an = ord2f(prefix);
    // |approximate - exact chain| for ANY centroid: fp16 rounding of the operands (relative a.ef, Cauchy-Schwarz), the fp32 accumulation of both
    // sums in any order (2 d 2^-24, doubled), and the ABSOLUTE rounding of operand values below fp16's normal range (2^-25 each: centroid
    // components against |x|, fp32 query components against the largest centroid norm)
    const float e = xn * (a.cmax * (a.ef * 1.002f + (float)d * 2.4e-7f) + sqrtf((float)d) * 1.2e-7f) + sqrtf((float)d) * a.cmax * 6.0e-8f;
    float T = an - 2.0f * e;
    T -= fabsf(T) * 1e-6f + 1e-30f;
    if (!(fabsf(T) < __builtin_inff())) badl = true;
    const uint32_t Tk = f2ord(T);
#pragma unroll
    for (int u = 0; u < KPT; u++) {
        const int i = 4 * (tid + 256 * (u >> 2)) + (u & 3);
        if (rk[u] >= Tk && i < nlist) { const int pos = atomicAdd(&s_cnt, 1); if (pos < a.cmax_rt) cand[pos] = i; }
    }
    if (tid < CP_CMAX) sorted[tid] = 0ull;
    __syncthreads();
    const int ncand = s_cnt < a.cmax_rt ? s_cnt : a.cmax_rt;
    if (s_cnt > a.cmax_rt || s_cnt < a.nprobe) badl = true;
    // 3. exact chains: lane = candidate.  The rows arrive in chunks of CP_CH dimensions, fetched by ALL threads (coalesced, every load
    // independent; a lane streaming its own 3 KB row exposes one L2 round trip per 128-byte line) into an LDS stage whose row stride of
    // CP_RS floats keeps 16-byte reads conflict-free; the next chunk's loads are in flight while the lanes run the current one
    uint64_t key = 0ull;
    float s = 0.0f;
    const int myl = tid < ncand ? cand[tid] : 0;
    CP_MARK(3);
    constexpr int C4 = CP_CH / 4;           // float4 per row chunk
    const bool vec = (d & 3) == 0;
    const int nld = (ncand * C4 + 255) / 256;
    float4 preA[NLD], preB[NLD];          // TWO chunks in flight: the rows come from beyond the L2 (12.6 MB of centroids, 123 MB of row reads per batch)
    auto fetch = [&](float4* pre, int t0) {
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int i = tid + 256 * u, r = i / C4, c = i % C4;
            if (u < nld && r < ncand && t0 < dp) {
                const float* src = a.C + (int64_t)cand[r] * d + t0 + 4 * c;
                if (vec) { if (t0 + 4 * c < d) pre[u] = *reinterpret_cast<const float4*>(src); }
                else {
                    if (t0 + 4 * c < d) pre[u].x = src[0];
                    if (t0 + 4 * c + 1 < d) pre[u].y = src[1];
                    if (t0 + 4 * c + 2 < d) pre[u].z = src[2];
                    if (t0 + 4 * c + 3 < d) pre[u].w = src[3];
                }
            }
        }
    };
    auto step = [&](float4* pre, int t0) {          // chunk t0 (in `pre`) -> LDS, its registers re-armed with chunk t0 + 2 CP_CH, the lanes' chains over it
        if (t0) __syncthreads();          // the lanes are done with the previous chunk
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const int i = tid + 256 * u, r = i / C4, c = i % C4;
            if (u < nld && r < ncand) *reinterpret_cast<float4*>(stage + r * CP_RS + 4 * c) = pre[u];
        }
        __syncthreads();
        fetch(pre, t0 + 2 * CP_CH);
        if (tid < ncand) {
            const float4* row = reinterpret_cast<const float4*>(stage + tid * CP_RS);
            const float4* xx = reinterpret_cast<const float4*>(xq + t0);
#pragma unroll 8
            for (int t = 0; t < C4; t++) {
                const float4 c = row[t], x = xx[t];
                s = __fmaf_rn(x.x, c.x, s); s = __fmaf_rn(x.y, c.y, s); s = __fmaf_rn(x.z, c.z, s); s = __fmaf_rn(x.w, c.w, s);
            }
        }
    };
    fetch(preA, 0);
    fetch(preB, CP_CH);
    for (int t0 = 0; t0 < dp; t0 += 2 * CP_CH) {
        step(preA, t0);
        if (t0 + CP_CH < dp) step(preB, t0 + CP_CH);
    }
    if (tid < ncand) {
        if (!(fabsf(s) < __builtin_inff())) badl = true;
        key = make_key(s, (uint32_t)myl);
    }
    if (tid < CP_CMAX) skey[tid] = key;
    if (badl) s_bad = 1;
    __syncthreads();
    CP_MARK(4);
    // order by rank: the keys are distinct (they carry the list number)
    if (tid < ncand && key) {
        int rank = 0;
        for (int j2 = 0; j2 < ncand; j2++) rank += skey[j2] > key;
        sorted[rank] = key;
    }
    __syncthreads();
    // 4. probes + row layout
    CP_MARK(5);
    if (tid < a.nprobe) {
        const uint64_t kk = sorted[tid];
        const int32_t l = kk ? (int32_t)key_idx(kk) : -1;
        a.probe_list[q * a.nprobe + tid] = l;
        a.dis0[q * a.nprobe + tid] = kk ? key_score(kk) : -__builtin_inff();
        const int64_t len = l >= 0 ? a.list_len[l] : 0;
        lens[tid] = (len + a.pad_to - 1) / a.pad_to * a.pad_to;
    }
    __syncthreads();
    if (tid == 0) {
        int64_t off = 0;
        int64_t* ssg = a.seg_start + q * (a.nprobe + 1);
        for (int j = 0; j < a.nprobe; j++) { ssg[j] = off; off += lens[j]; }
        ssg[a.nprobe] = off;
        a.bad[q] = s_bad ? 2 : 0;
    }
    CP_MARK(6);
}
int coarse_pick_cmax(int nprobe) { int c = (nprobe + 16 + 15) / 16 * 16; return c > CP_CMAX ? CP_CMAX : c; }      // candidate rows: nprobe + a margin
size_t coarse_pick_lds(int nlist, int d, int nprobe) { return ((size_t)((d + CP_CH - 1) / CP_CH * CP_CH) + (size_t)coarse_pick_cmax(nprobe) * CP_RS) * 4; }
void launch_coarse_pick(const CoarsePickArgs& a0, int64_t nq, hipStream_t st) {
    if (nq <= 0) return;
    CoarsePickArgs a = a0;
    a.cmax_rt = coarse_pick_cmax(a.nprobe);
    const size_t shm = coarse_pick_lds(a.nlist, a.d, a.nprobe);
    const int ki = a.nlist <= 1024 ? 0 : a.nlist <= 4096 ? 1 : 2, li = a.cmax_rt <= 48 ? 0 : 1;
    void (*const kerns[3][2])(CoarsePickArgs) = {{k_coarse_pick<4, 6>, k_coarse_pick<4, 8>}, {k_coarse_pick<16, 6>, k_coarse_pick<16, 8>}, {k_coarse_pick<64, 6>, k_coarse_pick<64, 8>}};
    auto kern = kerns[ki][li];
    static DevSize attr[6];
    attr[2 * ki + li].grow(shm, [&] { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); });
    a.nq = nq;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nq + a.lp0.nblocks)), dim3(256), shm, st, a);
}

// ---------------------------------------------------------------------------------------
// Group the (query, probe) pairs by inverted list so that a list's vectors are streamed from
// HBM once for all the queries of the batch that probe it (list-major scheduling).
// ---------------------------------------------------------------------------------------
__global__ void k_zero_i32(int32_t* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
// jmin/jmax: only pairs whose probe rank j = i % nprobe lies in [jmin, jmax) take part
__global__ void k_pair_hist(const int32_t* probe_list, int64_t npairs, int nprobe, int jmin, int jmax, int32_t* cnt) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npairs) {
        int j = (int)(i % nprobe);
        int32_t l = probe_list[i];
        if (l >= 0 && j >= jmin && j < jmax) atomicAdd(&cnt[l], 1);
    }
}
// Exclusive scan of three per-thread values over a 1024-thread workgroup (wave shuffles + one LDS hop);
// tot[0..2] receive the workgroup totals.  scratch: 3 * 16 ints of LDS.
__device__ inline void block_excl_scan3(int32_t& a, int32_t& b, int32_t& c, int32_t* scratch, int32_t* tot) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int32_t ia = a, ib = b, ic = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int32_t ya = __shfl_up(ia, off), yb = __shfl_up(ib, off), yc = __shfl_up(ic, off);
        if (lane >= off) { ia += ya; ib += yb; ic += yc; }
    }
    if (lane == 63) { scratch[w] = ia; scratch[16 + w] = ib; scratch[32 + w] = ic; }
    __syncthreads();
    if (t < 16) {
        int32_t va = scratch[t], vb = scratch[16 + t], vc = scratch[32 + t];
        int32_t ja = va, jb = vb, jc = vc;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            int32_t ya = __shfl_up(ja, off), yb = __shfl_up(jb, off), yc = __shfl_up(jc, off);
            if (t >= off) { ja += ya; jb += yb; jc += yc; }
        }
        scratch[t] = ja - va; scratch[16 + t] = jb - vb; scratch[32 + t] = jc - vc;   // exclusive wave offsets
        if (t == 15) { tot[0] = ja; tot[1] = jb; tot[2] = jc; }
    }
    __syncthreads();
    a = ia - a + scratch[w]; b = ib - b + scratch[16 + w]; c = ic - c + scratch[32 + w];
}

// single workgroup exclusive scan over lists: pair_off (pairs) and group_off (groups of G pairs)
__global__ __launch_bounds__(1024) void k_pair_scan(const int32_t* cnt, int nlist, int G, int32_t* pair_off,
                                                    int32_t* group_off, int32_t* total_groups,
                                                    const int64_t* list_len, int tile_rows, int tile_cap,
                                                    int32_t* item_off, int32_t* total_items) {
    __shared__ int32_t sp[48], sg[3];
    // This single workgroup runs on the side stream BESIDE the threshold pre-pass, whose 1024-thread workgroups fill every CU: alone
    // it takes 10 us, sharing a CU's issue slots 75 us — and the item records (hence the scan) wait for it.  Top issue priority, the
    // tile counts kept in registers between the two passes, and a 32-bit division where the list length allows (always, in practice).
    __builtin_amdgcn_s_setprio(3);
    // tiles of list l that become work items (tile_cap > 0 limits them, e.g. to the first tile only)
    auto ntiles = [&](int l) {
        const int64_t len = list_len[l];
        int32_t t = len <= 0x7ffffffe - tile_rows ? (int32_t)(((uint32_t)len + (uint32_t)tile_rows - 1u) / (uint32_t)tile_rows)
                                                  : (int32_t)((len + tile_rows - 1) / tile_rows);
        return (tile_cap > 0 && t > tile_cap) ? tile_cap : t;
    };
    int t = threadIdx.x;
    int per = (nlist + 1023) / 1024;
    int lo = t * per, hi = lo + per;
    if (hi > nlist) hi = nlist;
    constexpr int KEEP = 8;                     // lists per thread whose counts stay in registers (nlist <= 8192)
    int32_t c_[KEEP], n_[KEEP];
    int32_t ap = 0, ag = 0, ai = 0;
#pragma unroll
    for (int j = 0; j < KEEP; j++) {
        const int l = lo + j;
        c_[j] = 0; n_[j] = 0;
        if (l < hi) {
            c_[j] = cnt[l];
            const int ng = (c_[j] + G - 1) / G;
            if (tile_rows > 0) n_[j] = ng * ntiles(l);
            ap += c_[j]; ag += ng; ai += n_[j];
        }
    }
    for (int l = lo + KEEP; l < hi; l++) {
        int ng = (cnt[l] + G - 1) / G;
        ap += cnt[l]; ag += ng;
        if (tile_rows > 0) ai += ng * ntiles(l);
    }
    block_excl_scan3(ap, ag, ai, sp, sg);
    if (t == 0) {
        pair_off[nlist] = sg[0]; group_off[nlist] = sg[1]; *total_groups = sg[1];
        if (tile_rows > 0) { item_off[nlist] = sg[2]; *total_items = sg[2]; }
    }
#pragma unroll
    for (int j = 0; j < KEEP; j++) {
        const int l = lo + j;
        if (l < hi) {
            const int ng = (c_[j] + G - 1) / G;
            pair_off[l] = ap; group_off[l] = ag;
            if (tile_rows > 0) { item_off[l] = ai; ai += n_[j]; }
            ap += c_[j]; ag += ng;
        }
    }
    for (int l = lo + KEEP; l < hi; l++) {
        int ng = (cnt[l] + G - 1) / G;
        pair_off[l] = ap; group_off[l] = ag;
        if (tile_rows > 0) { item_off[l] = ai; ai += ng * ntiles(l); }
        ap += cnt[l]; ag += ng;
    }
}
__global__ void k_pair_scatter(const int32_t* probe_list, int64_t npairs, int nprobe, int jmin, int jmax,
                               const int32_t* pair_off, int32_t* cursor, int32_t* pairs_sorted) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npairs) {
        int j = (int)(i % nprobe);
        int32_t l = probe_list[i];
        if (l >= 0 && j >= jmin && j < jmax) { int s = atomicAdd(&cursor[l], 1); pairs_sorted[pair_off[l] + s] = (int32_t)i; }
    }
}
// The same grouping in ONE launch of one workgroup (histogram and cursors in LDS) for the sizes a search batch
// has: five dependent launches cost ~45 us of launch latency, far more than the work of a few thousand pairs.
#define GP1_MAX_LISTS 8192
__global__ __launch_bounds__(1024) void k_group_pairs_1wg(const int32_t* probe_list, int npairs, int nlist, int G,
                                                          int32_t* pair_off, int32_t* group_off, int32_t* total_groups,
                                                          int32_t* pairs_sorted, const int64_t* list_len, int tile_rows,
                                                          int32_t* item_off, int32_t* total_items, int nprobe, int jmin,
                                                          int jmax, int tile_cap) {
    extern __shared__ int32_t gp_lds[];
    int32_t* cnt = gp_lds;                 // [nlist] histogram, then running cursor
    int32_t* sp = gp_lds + nlist;          // [48] scan scratch
    int32_t* sg = sp + 48;                 // [3] totals
    const int t = threadIdx.x;
    auto ntiles = [&](int l) {
        int32_t n = (int32_t)((list_len[l] + tile_rows - 1) / tile_rows);
        return (tile_cap > 0 && n > tile_cap) ? tile_cap : n;
    };
    for (int l = t; l < nlist; l += 1024) cnt[l] = 0;
    __syncthreads();
    for (int i = t; i < npairs; i += 1024) {
        const int j = i % nprobe;
        const int32_t l = probe_list[i];
        if (l >= 0 && j >= jmin && j < jmax) atomicAdd(&cnt[l], 1);
    }
    __syncthreads();
    const int per = (nlist + 1023) / 1024;
    const int lo = t * per;
    int hi = lo + per; if (hi > nlist) hi = nlist;
    int32_t ap = 0, ag = 0, ai = 0;
    for (int l = lo; l < hi; l++) {
        const int ng = (cnt[l] + G - 1) / G;
        ap += cnt[l]; ag += ng;
        if (tile_rows > 0) ai += ng * ntiles(l);
    }
    block_excl_scan3(ap, ag, ai, sp, sg);
    if (t == 0) {
        pair_off[nlist] = sg[0]; group_off[nlist] = sg[1]; *total_groups = sg[1];
        if (tile_rows > 0) { item_off[nlist] = sg[2]; *total_items = sg[2]; }
    }
    for (int l = lo; l < hi; l++) {
        const int c = cnt[l];
        const int ng = (c + G - 1) / G;
        pair_off[l] = ap; group_off[l] = ag;
        if (tile_rows > 0) { item_off[l] = ai; ai += ng * ntiles(l); }
        cnt[l] = ap;                       // from here on: the list's write cursor
        ap += c; ag += ng;
    }
    __syncthreads();
    for (int i = t; i < npairs; i += 1024) {
        const int j = i % nprobe;
        const int32_t l = probe_list[i];
        if (l >= 0 && j >= jmin && j < jmax) pairs_sorted[atomicAdd(&cnt[l], 1)] = (int32_t)i;
    }
}

void launch_group_pairs(const int32_t* probe_list, int64_t npairs, int nlist, int group_size, int32_t* cnt,
                        int32_t* cursor, int32_t* pair_off, int32_t* group_off, int32_t* total_groups,
                        int32_t* pairs_sorted, const int64_t* list_len, int tile_rows, int32_t* item_off,
                        int32_t* total_items, int nprobe, int jmin, int jmax, int tile_cap, hipStream_t st) {
    if (nlist <= GP1_MAX_LISTS && npairs <= 8192) {   // larger batches: the multi-launch form is parallel and faster (32 k pairs: 45 vs 65 us)
        size_t shm = ((size_t)nlist + 64) * 4;
        if (shm > 48 * 1024) hipFuncSetAttribute((const void*)k_group_pairs_1wg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(k_group_pairs_1wg, dim3(1), dim3(1024), shm, st, probe_list, (int)npairs, nlist, group_size, pair_off,
                           group_off, total_groups, pairs_sorted, list_len, tile_rows, item_off, total_items, nprobe, jmin, jmax,
                           tile_cap);
        return;
    }
    if (cursor == cnt + (nlist + 1)) {      // the callers lay the two arrays end to end: one launch
        hipLaunchKernelGGL(k_zero_i32, dim3((2 * nlist + 1 + 255) / 256), dim3(256), 0, st, cnt, 2 * nlist + 1);
    } else {
        hipLaunchKernelGGL(k_zero_i32, dim3((nlist + 255) / 256), dim3(256), 0, st, cnt, nlist);
        hipLaunchKernelGGL(k_zero_i32, dim3((nlist + 255) / 256), dim3(256), 0, st, cursor, nlist);
    }
    hipLaunchKernelGGL(k_pair_hist, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, probe_list, npairs, nprobe, jmin, jmax, cnt);
    hipLaunchKernelGGL(k_pair_scan, dim3(1), dim3(1024), 0, st, cnt, nlist, group_size, pair_off, group_off, total_groups,
                       list_len, tile_rows, tile_cap, item_off, total_items);
    hipLaunchKernelGGL(k_pair_scatter, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, probe_list, npairs,
                       nprobe, jmin, jmax, pair_off, cursor, pairs_sorted);
}

// ---------------------------------------------------------------------------------------
// Finalize: one workgroup (1-4 waves) per query.  Resolve each surviving candidate to its storage row and id,
// re-score Flat / IVF-Flat candidates EXACTLY (fp64 accumulation of exact products, rounded
// once to fp32 — the oracle's canonical arithmetic), sort by (score desc, id asc) and emit k.
// IVF-PQ scores are already canonical fp32 (sequential LUT sum) and are only re-ordered.
// ---------------------------------------------------------------------------------------
__device__ inline double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Exact IVF-PQ score of one stored vector: sum over m (ascending, fp32) of the table entry of its code.
// The codes of a vector sit in CB-byte granules (one per 64*CB-byte slab row block); one granule is
// fetched ahead so the dependent code -> codeword loads of different m overlap.
template <int CB>
__device__ inline float pq_exact_sum(const uint8_t* sp, int v, int M, const float* T, const float* qv,
                                     const float* codebooks, int dsub) {
    constexpr int NW = CB / 4;
    const int G = (M + CB - 1) / CB;
    const uint8_t* gp = sp + v * CB;
    uint32_t cur[NW], nxt[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) cur[i] = ((const uint32_t*)gp)[i];
    float sum = 0.0f;
    for (int g = 0; g < G; g++) {
        const uint32_t* np = (const uint32_t*)(gp + (int64_t)(g + 1 < G ? g + 1 : g) * (64 * CB));
#pragma unroll
        for (int i = 0; i < NW; i++) nxt[i] = np[i];
#pragma unroll
        for (int b = 0; b < CB; b++) {
            const int m = g * CB + b;
            const int mc = m < M ? m : M - 1;
            const uint32_t code = (cur[b >> 2] >> (8 * (b & 3))) & 255u;
            float t;
            if (T) {
                t = T[mc * 256 + code];
            } else {   // the table entry, recomputed with k_pq_lut's fmaf chain (bit-identical)
                const float* qs = qv + mc * dsub;
                const float* cw = codebooks + ((int64_t)mc * 256 + code) * dsub;
                t = 0.0f;
                if (dsub == 8) {
                    float4 x = ((const float4*)cw)[0], y = ((const float4*)cw)[1];
                    t = __fmaf_rn(qs[0], x.x, t); t = __fmaf_rn(qs[1], x.y, t); t = __fmaf_rn(qs[2], x.z, t); t = __fmaf_rn(qs[3], x.w, t);
                    t = __fmaf_rn(qs[4], y.x, t); t = __fmaf_rn(qs[5], y.y, t); t = __fmaf_rn(qs[6], y.z, t); t = __fmaf_rn(qs[7], y.w, t);
                } else {
                    for (int tt = 0; tt < dsub; tt++) t = __fmaf_rn(qs[tt], cw[tt], t);
                }
            }
            sum = m < M ? sum + t : sum;
        }
#pragma unroll
        for (int i = 0; i < NW; i++) cur[i] = nxt[i];
    }
    return sum;
}

// Rotated layout (CB = 0): the M code bytes of a vector through pq_code_addr, eight at a time so that the
// byte -> table-entry loads of different m overlap; summed in m order like the granule form.
__device__ inline float pq_exact_sum_rot(const uint8_t* codes, int64_t row, int M, const float* T, const float* qv,
                                         const float* codebooks, int dsub, int CB = 0) {
    float sum = 0.0f;
    for (int m0 = 0; m0 < M; m0 += 8) {
        uint32_t code[8];
#pragma unroll
        for (int j = 0; j < 8; j++) code[j] = codes[pq_code_addr(row, m0 + j, M, CB)];
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int m = m0 + j;
            if (T) {
                t[j] = T[m * 256 + code[j]];
            } else {
                const float* qs = qv + m * dsub;
                const float* cw = codebooks + ((int64_t)m * 256 + code[j]) * dsub;
                float e = 0.0f;
                if (dsub == 8) {
                    float4 x = ((const float4*)cw)[0], y = ((const float4*)cw)[1];
                    e = __fmaf_rn(qs[0], x.x, e); e = __fmaf_rn(qs[1], x.y, e); e = __fmaf_rn(qs[2], x.z, e); e = __fmaf_rn(qs[3], x.w, e);
                    e = __fmaf_rn(qs[4], y.x, e); e = __fmaf_rn(qs[5], y.y, e); e = __fmaf_rn(qs[6], y.z, e); e = __fmaf_rn(qs[7], y.w, e);
                } else {
                    for (int tt = 0; tt < dsub; tt++) e = __fmaf_rn(qs[tt], cw[tt], e);
                }
                t[j] = e;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) sum += t[j];
    }
    return sum;
}

// The same sum with the vector's code bytes fetched as the 16-byte (8-byte) pieces the rotated layout stores (rsx_internal.h):
// for every run of 16 sub-quantisers the piece of lane group g, rotated left by i = row & 15 bytes, holds the codes in m
// order.  One round trip for all M codes instead of M byte loads, and 16 independent codeword loads per run; table entries
// by the table builder's fmaf chain, added in m order — the identical fp32 value.  dsub == 8, M in {32, 64, 96, 128}.
__device__ inline void rot16_bytes(uint32_t (&w)[4], int i) {     // byte t of the result = byte (t - i) & 15 of the input
    const int wi = i >> 2, bi = i & 3;
    if (wi & 1) { const uint32_t t = w[3]; w[3] = w[2]; w[2] = w[1]; w[1] = w[0]; w[0] = t; }
    if (wi & 2) { uint32_t t = w[0]; w[0] = w[2]; w[2] = t; t = w[1]; w[1] = w[3]; w[3] = t; }
    if (bi) {
        const int l = 8 * bi, r = 32 - l;
        const uint32_t r0 = (w[0] << l) | (w[3] >> r), r1 = (w[1] << l) | (w[0] >> r), r2 = (w[2] << l) | (w[1] >> r), r3 = (w[3] << l) | (w[2] >> r);
        w[0] = r0; w[1] = r1; w[2] = r2; w[3] = r3;
    }
}
__device__ inline float pq_exact_sum_rot_wide(const uint8_t* codes, int64_t row, int M, const float* qv, const float* codebooks, int CB = 0,
                                              const uint8_t* plain = nullptr, int plain_stride = 0) {
    const int i = plain ? 0 : (int)(row & 15);       // the row-major copy holds the bytes in m order already: nothing to rotate back
    const int nrun = M >> 4;
    // the piece of a run as two 8-byte halves (pq_piece_ptrs: rotated or sliced layout; or straight from the row-major copy)
    auto piece = [&](int r, uint2& lo, uint2& hi) {
        const uint8_t* p0; const uint8_t* p1;
        if (plain) { p0 = plain + row * (plain_stride ? plain_stride : M) + r * 16; p1 = p0 + 8; }
        else pq_piece_ptrs(codes, row, M, CB, r, p0, p1);
        lo = *reinterpret_cast<const uint2*>(p0); hi = *reinterpret_cast<const uint2*>(p1);
    };
    uint2 nlo, nhi;
    piece(0, nlo, nhi);
    float sum = 0.0f;
    for (int run = 0; run < nrun; run++) {
        uint32_t w[4] = {nlo.x, nlo.y, nhi.x, nhi.y};
        if (run + 1 < nrun) piece(run + 1, nlo, nhi);        // the next run's codes travel while this run's codewords do
        rot16_bytes(w, i);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int jj = half * 8 + j, m = run * 16 + jj;
                const uint32_t code = (w[jj >> 2] >> (8 * (jj & 3))) & 255u;
                const float* qs = qv + m * 8;
                const float* cw = codebooks + ((int64_t)m * 256 + code) * 8;
                const float4 x = ((const float4*)cw)[0], y = ((const float4*)cw)[1];
                float e = 0.0f;
                e = __fmaf_rn(qs[0], x.x, e); e = __fmaf_rn(qs[1], x.y, e); e = __fmaf_rn(qs[2], x.z, e); e = __fmaf_rn(qs[3], x.w, e);
                e = __fmaf_rn(qs[4], y.x, e); e = __fmaf_rn(qs[5], y.y, e); e = __fmaf_rn(qs[6], y.z, e); e = __fmaf_rn(qs[7], y.w, e);
                t[j] = e;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) sum += t[j];
        }
    }
    return sum;
}

__global__ __launch_bounds__(1024) void k_finalize(FinalizeArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t fin_buf[];
    const int KP = a.KP;
    int64_t* sid = (int64_t*)fin_buf;
    int64_t* srow = sid + KP;
    uint32_t* sord = (uint32_t*)(srow + KP);
    const int tid = threadIdx.x, nt = blockDim.x;   // 1..4 waves
    const int lane = tid & 63, wv = tid >> 6, nwv = nt >> 6;
    const int64_t q = blockIdx.x;
    if (a.row_filter && a.row_filter[q] != 1) return;
    // IVF: the query's probe table in LDS (round 4) — a candidate's list is found by bisection of the row offsets, which on global
    // memory is five dependent L2 round trips per candidate, twice (location, then the coarse score of the re-score)
    const bool plds = a.probe_lds_off != 0 && a.kind != KIND_FLAT;
    int64_t* s_ss = reinterpret_cast<int64_t*>(reinterpret_cast<unsigned char*>(fin_buf) + a.probe_lds_off);   // [nprobe + 1]
    int64_t* s_lb = s_ss + (a.nprobe + 1);                                                                       // [nprobe]
    float* s_d0 = reinterpret_cast<float*>(s_lb + a.nprobe);                                                     // [nprobe]
    if (plds) {
        for (int j = tid; j <= a.nprobe; j += nt) s_ss[j] = a.seg_start[q * (a.nprobe + 1) + j];
        for (int j = tid; j < a.nprobe; j += nt) {
            const int32_t l = a.probe_list[q * a.nprobe + j];
            s_lb[j] = l >= 0 ? a.list_base[l] : 0;
            s_d0[j] = a.probe_dis0 ? a.probe_dis0[q * a.nprobe + j] : 0.0f;
        }
        __syncthreads();
    }
    const int64_t* ssq = plds ? s_ss : a.seg_start + q * (a.nprobe + 1);

    const int KPv = (a.KPv > 0 && a.KPv < KP) ? a.KPv : KP;
    // candidates [c_lo, c_hi) of the state row -> (id, storage row, approximate order word) in slot c; the others of [c_lo, c_end) become sinks
    auto resolve = [&](int c_lo, int c_hi, int c_end) {
    for (int c = c_lo + tid; c < c_end; c += nt) {
        uint64_t key = c < c_hi ? a.state[q * KP + c] : 0ull;
        int64_t row = -1, id = INT64_MAX;
        uint32_t ord = 0;
        if (key) {
            uint32_t idx = key_idx(key);
            if (a.kind == KIND_FLAT) {
                row = idx;
            } else {
                const int64_t* ss = ssq;
                int lo = 0, hi = a.nprobe;  // find j with ss[j] <= idx < ss[j+1]
                while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (ss[mid] <= (int64_t)idx) lo = mid; else hi = mid; }
                row = (plds ? s_lb[lo] : a.list_base[a.probe_list[q * a.nprobe + lo]]) + ((int64_t)idx - ss[lo]);
            }
            id = a.ids ? a.ids[row] : row;
            ord = (uint32_t)(key >> 32);
        }
        sid[c] = id; srow[c] = row; sord[c] = ord;
    }
    __syncthreads();
    };
    resolve(0, KPv, KP);

    if (a.kind == KIND_IVFPQ && a.pq_rescore && a.par_entries) {
        // The K' x M table entries of the candidates' codes are independent — all threads compute them (code byte -> codeword -> 8 fmaf,
        // the table builder's chain) into LDS, then one thread per candidate adds its row in m order: the same canonical sum, without M
        // dependent loads per thread.  The latency path (few queries in flight: launch_finalize).  Round 5: the code bytes arrive as 8-byte
        // PIECES (a thread per (candidate, 16-sub-quantiser run): 2 x 8 bytes, rotated back into m order, 16 bytes to LDS) instead of one
        // byte load per entry.  (Measured for full batches at K' = 128 as well: 16 waves x 52 KiB per query leave two queries per CU, and
        // finalize went 0.060 -> 0.14 ms per 1024 queries — one thread per candidate stays the form for batches.)
        float* ent = reinterpret_cast<float*>(fin_buf) + (((size_t)KP * 20 + 15) / 16) * 4;      // [KP][M + 1]
        const int M = a.M, es = M + 1;
        uint8_t* cbytes = reinterpret_cast<uint8_t*>(ent + (size_t)KP * es);                      // [KP][M] (M >= 32: staged by pieces)
        const bool pieces = M >= 32 && (M & 15) == 0;
        if (pieces) {
            const int nrun = M >> 4;
            for (int u = tid; u < KP * nrun; u += nt) {
                const int c = u / nrun, run = u - c * nrun;
                const int64_t row = srow[c];
                if (row < 0) continue;
                const int i = (int)(row & 15);
                const uint8_t* p0; const uint8_t* p1;
                pq_piece_ptrs(a.codes, row, M, a.CB, run, p0, p1);
                const uint2 lo2 = *reinterpret_cast<const uint2*>(p0), hi2 = *reinterpret_cast<const uint2*>(p1);
                uint32_t w[4] = {lo2.x, lo2.y, hi2.x, hi2.y};
                rot16_bytes(w, i);
                *reinterpret_cast<uint4*>(cbytes + (size_t)c * M + run * 16) = make_uint4(w[0], w[1], w[2], w[3]);
            }
            __syncthreads();
        }
        const float* qv = a.Q32 + q * a.ldq;
        for (int e = tid; e < KP * M; e += nt) {
            const int c = e / M, m = e - c * M;
            const int64_t row = srow[c];
            if (row < 0) continue;
            const uint32_t code = pieces ? (uint32_t)cbytes[(size_t)c * M + m] : (uint32_t)a.codes[pq_code_addr(row, m, M, a.CB)];
            const float* qs = qv + m * 8;
            const float* cw = a.codebooks + ((int64_t)m * 256 + code) * 8;
            const float4 x = ((const float4*)cw)[0], y = ((const float4*)cw)[1];
            float t = 0.0f;
            t = __fmaf_rn(qs[0], x.x, t); t = __fmaf_rn(qs[1], x.y, t); t = __fmaf_rn(qs[2], x.z, t); t = __fmaf_rn(qs[3], x.w, t);
            t = __fmaf_rn(qs[4], y.x, t); t = __fmaf_rn(qs[5], y.y, t); t = __fmaf_rn(qs[6], y.z, t); t = __fmaf_rn(qs[7], y.w, t);
            ent[c * es + m] = t;
        }
        __syncthreads();
        for (int c = tid; c < KP; c += nt) {
            if (srow[c] < 0) continue;
            const uint32_t idx = key_idx(a.state[q * KP + c]);
            const int64_t* ss = ssq;
            int lo = 0, hi = a.nprobe;
            while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (ss[mid] <= (int64_t)idx) lo = mid; else hi = mid; }
            const float dis0 = plds ? s_d0[lo] : a.probe_dis0[q * a.nprobe + lo];
            float sum = 0.0f;
            for (int m = 0; m < M; m++) sum += ent[c * es + m];
            sord[c] = f2ord((dis0 + sum) + 0.0f);
        }
        __syncthreads();
    } else if (a.kind == KIND_IVFPQ && a.pq_rescore) {
        // one thread per candidate: canonical score = dis0 + (((0 + T[0][c0]) + T[1][c1]) + ...), fp32
        const float* T = a.lut32 ? a.lut32 + q * a.Mpad * 256 : nullptr;
        const float* qv = a.Q32 + q * a.ldq;
        for (int c = tid; c < KP; c += nt) {
            int64_t row = srow[c];
            if (row < 0) continue;
            uint32_t idx = key_idx(a.state[q * KP + c]);
            const int64_t* ss = ssq;
            int lo = 0, hi = a.nprobe;
            while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (ss[mid] <= (int64_t)idx) lo = mid; else hi = mid; }
            const float dis0 = plds ? s_d0[lo] : a.probe_dis0[q * a.nprobe + lo];
            const int64_t slab = row >> 6; const int v = (int)(row & 63);
            const uint8_t* sp = a.codes + slab * (int64_t)(64 * a.Mpad);
            const float sum = pq_rot_family(a.CB) ? ((!T && a.dsub == 8 && a.M >= 32) ? pq_exact_sum_rot_wide(a.codes, row, a.M, qv, a.codebooks, a.CB, a.codes_plain, a.plain_stride)
                                                                  : pq_exact_sum_rot(a.codes, row, a.M, T, qv, a.codebooks, a.dsub, a.CB))
                            : a.CB == 16 ? pq_exact_sum<16>(sp, v, a.M, T, qv, a.codebooks, a.dsub)
                                         : pq_exact_sum<4>(sp, v, a.M, T, qv, a.codebooks, a.dsub);
            sord[c] = f2ord((dis0 + sum) + 0.0f);
        }
        __syncthreads();
    }
    auto rescore_rows = [&](int c_lo, int c_hi) {
        // one wave per candidate, lanes over the dimensions; FOUR candidates per wave in flight (round 4: at the reference's n_docs =
        // 1000 a query re-scores 2048 rows of 1.5 KB that sit anywhere in HBM — one row at a time per wave was 3.2 ms per 1024 queries)
        const float* qv = a.Q32 + q * a.ldq;
        for (int c0 = c_lo + 4 * wv; c0 < c_hi; c0 += 4 * nwv) {
            int64_t rows[4]; double acc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { rows[u] = c0 + u < c_hi ? srow[c0 + u] : -1; acc[u] = 0.0; }   // wave-uniform (LDS values)
            if (a.x_f16 && (a.ldq & 7) == 0 && a.ldq >= a.ld) {
                // fp16 rows: 16 bytes (8 values) per lane and load — a 1536-byte row is 1.5 wave loads instead of 12 two-byte ones (the
                // rows' zero padding up to ld contributes exact zeros; fp64 sums of products of fp16-valued numbers do not depend on the order)
                for (int c8 = lane; c8 < (a.ld >> 3); c8 += 64) {
                    const float4 qa = *reinterpret_cast<const float4*>(qv + 8 * c8), qb = *reinterpret_cast<const float4*>(qv + 8 * c8 + 4);
                    const float qf[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
                    uint4 xr[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) xr[u] = rows[u] >= 0 ? *reinterpret_cast<const uint4*>((const __half*)a.X + rows[u] * a.ld + 8 * c8) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t wds[4] = {xr[u].x, xr[u].y, xr[u].z, xr[u].w};
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const __half hx = __ushort_as_half((unsigned short)((wds[e >> 1] >> (16 * (e & 1))) & 0xffffu));
                            const double qd = (double)qf[e], xd = (double)__half2float(hx);
                            if (a.metric == 0) acc[u] += qd * xd; else { const double df = qd - xd; acc[u] += df * df; }
                        }
                    }
                }
            } else
            for (int t = lane; t < a.d; t += 64) {
                const double qd = (double)qv[t];
                double xd[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    xd[u] = 0.0;
                    if (rows[u] >= 0) xd[u] = a.x_f16 ? (double)__half2float(((const __half*)a.X)[rows[u] * a.ld + t]) : (double)((const float*)a.X)[rows[u] * a.ld + t];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) { if (a.metric == 0) acc[u] += qd * xd[u]; else { const double df = qd - xd[u]; acc[u] += df * df; } }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (rows[u] < 0) continue;
                const double sacc = wave_sum_f64(acc[u]);
                if (lane == 0) {
                    const float sc = (float)sacc;
                    sord[c0 + u] = f2ord((a.metric == 0 ? sc : 0.0f - sc) + 0.0f);
                }
            }
        }
        __syncthreads();
    };
    if (a.kind != KIND_IVFPQ) rescore_rows(0, KPv);

    // order by (ord desc, id asc); invalid entries (ord 0, id INT64_MAX) sink to the end.  Up to FIN_RANK_MAX candidates by
    // counting: a candidate's position is the number of candidates that beat it (every thread walks the same LDS words —
    // broadcasts — and there are two barriers instead of the 28 of a 128-key bitonic network); larger sets by the network.
    auto sort_all = [&]() {
    if (a.rank_sort) {
        uint32_t* sord2 = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(fin_buf) + (((size_t)KP * 20 + 15) / 16) * 16);
        int64_t* sid2 = reinterpret_cast<int64_t*>(sord2 + KP + (KP & 1));
        for (int c = tid; c < KP; c += nt) {
            const uint32_t oi = sord[c];
            const int64_t ii = sid[c];
            int rank = 0;
            for (int j = 0; j < KP; j++) {
                const uint32_t oj = sord[j];
                const int64_t ij = sid[j];
                rank += ((oj > oi) || (oj == oi && (ij < ii || (ij == ii && j < c)))) ? 1 : 0;
            }
            sord2[rank] = oi; sid2[rank] = ii;
        }
        __syncthreads();
        for (int c = tid; c < KP; c += nt) { sord[c] = sord2[c]; sid[c] = sid2[c]; }
        __syncthreads();
    } else
    for (int size = 2; size <= KP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (KP >> 1); t += nt) {
                int i = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
                int j = i + stride;
                bool desc = ((i & size) == 0);
                uint32_t oi = sord[i], oj = sord[j];
                int64_t ii = sid[i], ij = sid[j];
                bool i_worse = (oi < oj) || (oi == oj && ii > ij);
                if (i_worse == desc) {
                    sord[i] = oj; sord[j] = oi; sid[i] = ij; sid[j] = ii;
                }
            }
            __syncthreads();
        }
    }
    };
    sort_all();
    if (a.kind == KIND_IVFPQ && a.pq_rescore && tid == 0) {
        // certificate: every vector outside the candidate set has approximate score <= the K'-th
        // candidate's, hence exact score <= that + eps; if this is below the exact k-th best
        // candidate, the top k (ties included) lies inside the candidate set.
        const float eps = reinterpret_cast<const float*>(a.qparam)[q * 4 + 2];
        uint64_t last = a.state[q * KP + (KPv - 1)];
        int bad = 0;
        if (last != 0 && !a.no_cert) {  // the candidate buffer is full: vectors were excluded
            float a_last = key_score(last);
            bool have_k = sord[a.k - 1] != 0 && sid[a.k - 1] != INT64_MAX;
            float s_k = have_k ? ord2f(sord[a.k - 1]) : -__builtin_inff();
            if (!(a_last + eps < s_k)) bad = 1;
        }
        if (a.cand_cnt && a.cand_cnt[q * CCS] > (unsigned long long)a.cand_cap) bad |= 2;  // candidates were dropped (reason bit 1: overflow)
        // second chance without a certificate (the K2 best of a complete, exactly scored row, selected by (score, INDEX)): when the
        // selection was full and its last candidate ties with the k-th score, a lower-ID vector of the same score may have been cut
        // (identical codes in one list tie exactly) -> the exact re-run settles the query
        if (a.no_cert && last != 0 && sord[a.k - 1] != 0 && sord[KP - 1] == sord[a.k - 1]) bad |= 2;
        a.uncertain[q] = bad;
    }
    // certificate of the MFMA scan (see FinalizeArgs) against the `considered`-th approximate candidate.  EVERY wave evaluates it (|q|^2 is a
    // wave sum, the rest reads LDS and the state row): the verdict is the same in all of them and needs no LDS word to travel — the sort
    // buffers fill the whole 160 KiB at K' = 8192
    auto certify_rows = [&](int considered) -> int {
        const float* qv = a.Q32 + q * a.ldq;
        int bad = 0;
        {
            double q2 = 0.0;
            for (int t = lane; t < a.d; t += 64) q2 += (double)qv[t] * (double)qv[t];
            q2 = wave_sum_f64(q2);
            if (lane == 0) {
                const uint64_t last = a.state[q * KP + (considered - 1)];
                if (last != 0) {     // K' candidates were kept: vectors were excluded on their approximate score
                    const float qn = (float)sqrt(q2) * 1.0000002f;
                    const float rel = a.cert_rel + ((a.cert_qflag && *a.cert_qflag) ? a.cert_rel_qlossy : 0.0f);
                    const float eps = rel * qn * a.cert_xmax + a.cert_abs * (qn + a.cert_xmax);
                    const float a_last = key_score(last);
                    const bool have_k = sord[a.k - 1] != 0 && sid[a.k - 1] != INT64_MAX;
                    const float s_k = have_k ? ord2f(sord[a.k - 1]) : -__builtin_inff();
                    if (a.metric == 0) {
                        if (!(a_last + eps < s_k)) bad = 1;
                    } else {
                        // L2: the scan ranks by r = <q,x> - |x|^2/2 (larger = closer), the exact re-score is s = -|q - x|^2 =
                        // 2 r - |q|^2.  Round 3 fix: the two used to be compared as they stood (different quantities: every L2
                        // query with more than K' rows failed the certificate and took the exact path).  Bring the excluded
                        // vectors' bound into the re-score's units: s <= 2 (r~_last + eps) - |q|^2, plus the roundings of the
                        // conversion and of the fp32 result.
                        const double a_conv = 2.0 * (double)a_last - q2;
                        const double e_conv = 2.0 * (double)eps + 3.0e-7 * (fabs(a_conv) + q2 + fabs((double)s_k));
                        if (!(a_conv + e_conv < (double)s_k)) bad = 1;
                    }
                }
                if (a.cand_cnt)
                    for (int sn = 0; sn < (a.cand_cnt_n > 0 ? a.cand_cnt_n : 1); sn++)
                        if (a.cand_cnt[(int64_t)sn * a.cand_cnt_stride + q * CCS] > (unsigned long long)a.cand_cap) bad = 1;
            }
        }
        return __shfl(bad, 0);
    };
    if (a.kind != KIND_IVFPQ && a.uncertain) {
        // Round 5: the first pass re-scores only the KPv = k + max(8, k / 16) best approximate candidates (rows of 2 d bytes anywhere in
        // HBM) and certifies against the KPv-th.  A query it cannot clear is not sent to the exact re-run yet: the remaining candidates
        // of its state row (up to K', the power of two the selection kept) are resolved, re-scored and sorted in, and the certificate
        // runs again against the K'-th — what every query paid in round 4 (k = 1000: 2048 rows; finalize 1.05 -> 0.6 ms per 1024 queries).
        int bad = certify_rows(KPv);
        if (bad && KPv < KP) {
            resolve(KPv, KP, KP);         // slots [KPv, KP) held sinks after the first sort
            rescore_rows(KPv, KP);
            sort_all();
            bad = certify_rows(KP);
        }
        if (tid == 0) a.uncertain[q] = bad;
    }
    for (int j = tid; j < a.k; j += nt) {
        bool valid = (j < KP) && sord[j] != 0 && sid[j] != INT64_MAX;
        float s = valid ? ord2f(sord[j]) : -__builtin_inff();
        if (a.metric != 0) s = valid ? (0.0f - s) : __builtin_inff();
        a.D[q * a.k + j] = s;
        a.I[q * a.k + j] = valid ? sid[j] : -1;
    }
}

// one wave per score-buffer column; lanes over the dimensions; fp64 accumulation of exact products
__global__ __launch_bounds__(256) void k_exact_scores(ExactScoreArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t q = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= a.tstride) return;
    int64_t row = -1;
    if (a.kind == KIND_FLAT) { if (c < a.flat_n) row = c; }
    else {
        const int64_t* ss = a.seg_start + q * (a.nprobe + 1);
        if (c < ss[a.nprobe]) {
            int lo = 0, hi = a.nprobe;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ss[mid] <= c) lo = mid; else hi = mid; }
            const int32_t l = a.probe_list[q * a.nprobe + lo];
            if (l >= 0 && c - ss[lo] < a.list_len[l]) row = a.list_base[l] + (c - ss[lo]);
        }
    }
    float out = -__builtin_inff();
    if (row >= 0) {      // wave-uniform
        const float* qv = a.Q32 + q * a.ldq;
        double acc = 0.0;
        if (a.x_f16) {
            const __half* xv = (const __half*)a.X + row * a.ld;
            for (int t = lane; t < a.d; t += 64) {
                const double qd = (double)qv[t], xd = (double)__half2float(xv[t]);
                if (a.metric == 0) acc += qd * xd; else { const double df = qd - xd; acc += df * df; }
            }
        } else {
            const float* xv = (const float*)a.X + row * a.ld;
            for (int t = lane; t < a.d; t += 64) {
                const double qd = (double)qv[t], xd = (double)xv[t];
                if (a.metric == 0) acc += qd * xd; else { const double df = qd - xd; acc += df * df; }
            }
        }
        acc = wave_sum_f64(acc);
        const float s = (float)acc;
        out = (a.metric == 0 ? s : 0.0f - s) + 0.0f;
    }
    if (lane == 0) a.temp[q * a.tstride + c] = out;
}
// The same scores for fp16 rows, SIXTEEN consecutive columns per wave, four rows in flight at a time with 16-byte loads (k_finalize's
// rescore_rows), the probed list found by ONE bisection per wave and a walk from there.  Round 6: with k_exact_scores a query of the exact
// re-run cost ~2 ms at 1.2 M probed rows (seven dependent loads of the bisection and twelve 2-byte loads per row and wave) — the price of every
// certificate failure and, since this round, of every overflowed candidate row of the Flat / IVF-Flat filters.  fp64 sums of exact products: the
// order of the terms differs from k_exact_scores' as rescore_rows' does (oracle: "fp64 accumulation of exact products, rounded once").
#define XS_COLS 16
__global__ __launch_bounds__(256) void k_exact_scores_f16(ExactScoreArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t q = blockIdx.y;
    const int64_t c0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * XS_COLS;
    if (c0 >= a.tstride) return;
    const int64_t* ss = a.kind == KIND_FLAT ? nullptr : a.seg_start + q * (a.nprobe + 1);
    const int64_t total = a.kind == KIND_FLAT ? a.flat_n : ss[a.nprobe];
    int lo = 0;
    int64_t seg_lo = 0, seg_hi = 0, seg_base = 0, seg_len = 0;      // the probed list under the cursor: columns [seg_lo, seg_hi), rows seg_base + [0, seg_len)
    auto load_seg = [&]() {
        seg_lo = ss[lo]; seg_hi = ss[lo + 1];
        const int32_t l = a.probe_list[q * a.nprobe + lo];
        seg_base = l >= 0 ? a.list_base[l] : 0; seg_len = l >= 0 ? a.list_len[l] : 0;
    };
    if (ss && c0 < total) {
        int hi = a.nprobe;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ss[mid] <= c0) lo = mid; else hi = mid; }
        load_seg();
    }
    const float* qv = a.Q32 + q * a.ldq;
    for (int u4 = 0; u4 < XS_COLS; u4 += 4) {
        int64_t rows[4]; double acc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t c = c0 + u4 + u;
            rows[u] = -1; acc[u] = 0.0;
            if (c >= a.tstride || c >= total) continue;
            if (!ss) { rows[u] = c; continue; }
            while (c >= seg_hi && lo + 1 < a.nprobe) { lo++; load_seg(); }      // (padded segments: columns past the list's rows stay -inf)
            if (c >= seg_lo && c - seg_lo < seg_len) rows[u] = seg_base + (c - seg_lo);
        }
        for (int c8 = lane; c8 < (a.ld >> 3); c8 += 64) {
            const float4 qa = *reinterpret_cast<const float4*>(qv + 8 * c8), qb = *reinterpret_cast<const float4*>(qv + 8 * c8 + 4);
            const float qf[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
            uint4 xr[4];
#pragma unroll
            for (int u = 0; u < 4; u++) xr[u] = rows[u] >= 0 ? *reinterpret_cast<const uint4*>((const __half*)a.X + rows[u] * a.ld + 8 * c8) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t wds[4] = {xr[u].x, xr[u].y, xr[u].z, xr[u].w};
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const __half hx = __ushort_as_half((unsigned short)((wds[e >> 1] >> (16 * (e & 1))) & 0xffffu));
                    const double qd = (double)qf[e], xd = (double)__half2float(hx);
                    if (a.metric == 0) acc[u] += qd * xd; else { const double df = qd - xd; acc[u] += df * df; }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t c = c0 + u4 + u;
            if (c >= a.tstride) continue;
            float out = -__builtin_inff();
            if (rows[u] >= 0) {
                const float sc = (float)wave_sum_f64(acc[u]);
                out = (a.metric == 0 ? sc : 0.0f - sc) + 0.0f;
            }
            if (lane == 0) a.temp[q * a.tstride + c] = out;
        }
    }
}
void launch_exact_scores(const ExactScoreArgs& a, hipStream_t st) {
    if (a.nq <= 0 || a.tstride <= 0) return;
    // fp16 rows whose zero padding up to ld the query rows cover as well (ldq >= ld: the padding contributes exact zeros)
    if (a.x_f16 && (a.ldq & 7) == 0 && a.ldq >= a.ld && (a.ld & 7) == 0) {
        hipLaunchKernelGGL(k_exact_scores_f16, dim3((unsigned)((a.tstride + 4 * XS_COLS - 1) / (4 * XS_COLS)), (unsigned)a.nq), dim3(256), 0, st, a);
        return;
    }
    hipLaunchKernelGGL(k_exact_scores, dim3((unsigned)((a.tstride + 3) / 4), (unsigned)a.nq), dim3(256), 0, st, a);
}

#define FIN_RANK_MAX 512
// Exact re-score of EVERY candidate of the flagged queries, in place: cand[q][c] = (approximate score, index) becomes
// (canonical fp32 score, index).  With the round-3 threshold the candidate row of a query holds every vector that can reach
// its top k (unless the row overflowed), so a query whose K' best approximate candidates could not be certified is settled
// from its own row — |C| x M table look-ups — instead of an exact scan of all its probed lists.
__global__ __launch_bounds__(256) void k_pq_rescore_all(FinalizeArgs a, uint64_t* cand, int cand_cap) {
    const int64_t q = blockIdx.x;
    if (a.row_filter[q] != 1) return;
    unsigned long long n = a.cand_cnt[q * CCS];
    if (n > (unsigned long long)cand_cap) n = (unsigned long long)cand_cap;
    const float* T = a.lut32 ? a.lut32 + q * a.Mpad * 256 : nullptr;
    const float* qv = a.Q32 + q * a.ldq;
    const int64_t* ss = a.seg_start + q * (a.nprobe + 1);
    for (unsigned long long c = (unsigned long long)blockIdx.y * 256 + threadIdx.x; c < n; c += (unsigned long long)gridDim.y * 256) {
        const uint64_t key = cand[q * cand_cap + c];
        if (!key) continue;
        const uint32_t idx = key_idx(key);
        int lo = 0, hi = a.nprobe;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (ss[mid] <= (int64_t)idx) lo = mid; else hi = mid; }
        const int32_t l = a.probe_list[q * a.nprobe + lo];
        const int64_t row = a.list_base[l] + ((int64_t)idx - ss[lo]);
        const float dis0 = a.probe_dis0[q * a.nprobe + lo];
        const int64_t slab = row >> 6; const int v = (int)(row & 63);
        const uint8_t* sp = a.codes + slab * (int64_t)(64 * a.Mpad);
        const float sum = pq_rot_family(a.CB) ? ((!T && a.dsub == 8 && a.M >= 32) ? pq_exact_sum_rot_wide(a.codes, row, a.M, qv, a.codebooks, a.CB, a.codes_plain, a.plain_stride)
                                                              : pq_exact_sum_rot(a.codes, row, a.M, T, qv, a.codebooks, a.dsub, a.CB))
                        : a.CB == 16 ? pq_exact_sum<16>(sp, v, a.M, T, qv, a.codebooks, a.dsub)
                                     : pq_exact_sum<4>(sp, v, a.M, T, qv, a.codebooks, a.dsub);
        cand[q * cand_cap + c] = make_key(dis0 + sum, idx);
    }
}
void launch_pq_rescore_all(const FinalizeArgs& a, uint64_t* cand, int cand_cap, hipStream_t st) {
    if (a.nq <= 0) return;
    hipLaunchKernelGGL(k_pq_rescore_all, dim3((unsigned)a.nq, 16), dim3(256), 0, st, a, cand, cand_cap);
}

// ---------------------------------------------------------------------------------------
// IVF-PQ finalize from the COMPLETE candidate row (round 4): k_pq_final_tab.
// With the threshold that is valid by construction (tau = a_k - 2 eps from the pre-pass sample) a query's candidate
// row holds EVERY vector that can reach its top k.  Rounds 1-3 still cut the row to its K' best approximate keys, re-scored
// those through the codebooks (K' x M dependent codeword loads per query: 1.5 ms per 1024 queries at K' = 4096 / M = 96, 4.9 ms
// at M = 16 where an entry is a 48-term chain) and checked a certificate that fails for a quarter of the queries at M = 16 —
// each of which then went through k_pq_rescore_all + a second selection + a second finalize.  Here ONE workgroup per query
//   1. builds the query's fp32 table T[m][c] = <q_m, cb[m][c]> in LDS (M KiB; the table builder's fmaf chain: the oracle's bits),
//   2. re-scores every key of the row in place: dis0 + (((0 + T[0][c0]) + T[1][c1]) + ...) — M LDS look-ups per candidate,
//   3. finds the k-th largest exact score by a 4-round radix walk over the 32 score bits, collects the keys at or above it
//      (all ties of the k-th included) into LDS, resolves their ids and sorts that handful by (score desc, id asc),
//   4. emits k.  No certificate, no K', no second chance: a row is complete or it overflowed (flag 2 -> exact re-run).
// Rotated code layout only (CB = 0, M in {16, 32, 64, 96, 128}); P = sort capacity (power of two >= k; more ties than that at
// the k-th score send the query to the exact re-run).  row_filter != null: only the queries flagged 1 (second chance of the
// K' path) are processed.
// ---------------------------------------------------------------------------------------
#ifdef RSX_MEASURE
__device__ uint64_t g_ft_trace[256 * 8];          // tools/ builds only: [workgroup < 256][mark] wall clock (10 ns ticks) of k_pq_final_tab's phases
extern "C" int rsx_debug_ft_trace(uint64_t* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ft_trace), sizeof(uint64_t) * 256 * 8) == hipSuccess ? 0 : -1; }
#define FT_MARK(i) do { if (blockIdx.x < 256 && threadIdx.x == 0) g_ft_trace[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
__device__ uint64_t g_ft_trace2[256 * 8];         // ... and of the re-score phase's parts: count, k-th approximate, stage 1, k-th exact of stage 1, stage 2
extern "C" int rsx_debug_ft_trace2(uint64_t* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ft_trace2), sizeof(uint64_t) * 256 * 8) == hipSuccess ? 0 : -1; }
#define FT_MARK2(i) do { if (blockIdx.x < 256 && threadIdx.x == 0) g_ft_trace2[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define FT_MARK(i)
#define FT_MARK2(i)
#endif
// f(c, row[c]) for every c < n, a 1024-thread workgroup: EIGHT keys per thread are requested before the first is looked at.  (Round 6: the selection
// passes of k_pq_final_tab walked the row one key per thread and step — at n = 6400 seven dependent L2 round trips per pass, four passes per
// radix walk, three walks per query.)
template <class F>
__device__ __forceinline__ void ft_for_keys(const uint64_t* row, int n, int tid, F&& f) {
    for (int c0 = tid; c0 < n; c0 += 8 * 1024) {
        uint64_t kk[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int c = c0 + u * 1024; kk[u] = c < n ? row[c] : 0ull; }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int c = c0 + u * 1024; if (c < n) f(c, kk[u]); }
    }
}
template <int NRUN>      // M / 16: the 16-sub-quantiser runs of a code vector (1, 2, 4, 6, 8)
__global__ __launch_bounds__(1024) void k_pq_final_tab(FinalizeArgs a, uint64_t* cand, int cand_cap, int P, uint64_t* tie_ws, int stage_probes, int bm_off, int wq_off, int wq_cap) {
    extern __shared__ __attribute__((aligned(16))) float ft_T[];
    const int M = a.M, dsub = a.dsub;
    int64_t* sid = reinterpret_cast<int64_t*>(ft_T + (size_t)M * 256);     // [P]
    uint32_t* sord = reinterpret_cast<uint32_t*>(sid + P);                 // [P]
    uint64_t* tsel = reinterpret_cast<uint64_t*>(sord + P);                // [2]: the tie selection's id threshold
    int32_t* hist = reinterpret_cast<int32_t*>(tsel + 2);                  // [256]
    int32_t* ctl = hist + 256;                                             // [8]: 0 digit, 1 remaining, 2 valid, 3 cursor, 4 in-bin, 5 tie cursor, 6 exact-threshold flag
    int32_t* ctl2 = ctl + 8;                                               // [8]: the tie selection's radix state
    // the query's probe table in LDS (stage_probes): row offsets, list bases, coarse scores — a candidate's list is found by bisection,
    // five dependent L2 round trips per candidate when it runs on global memory
    int64_t* s_ss = reinterpret_cast<int64_t*>(ctl2 + 8);                   // [nprobe + 1]
    int64_t* s_lb = s_ss + (a.nprobe + 1);                                 // [nprobe]
    float* s_d0 = reinterpret_cast<float*>(s_lb + a.nprobe);               // [nprobe]
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t q = blockIdx.x;
    if (a.row_filter && a.row_filter[q] != 1) return;
    const unsigned long long n_raw = a.cand_cnt[q * CCS];
    if (n_raw > (unsigned long long)cand_cap) {         // keys were dropped (row or log overflow): the exact re-run settles this query
        if (tid == 0) a.uncertain[q] = 2;
        return;
    }
    const int n = (int)n_raw;
    uint64_t* row = cand + q * (int64_t)cand_cap;
    const float* qv = a.Q32 + q * a.ldq;
    FT_MARK(0);
    // 1. the table: copied when the table builder stored it (lut32: the same fmaf chains), else built here from the codebooks
    if (a.lut32) {
        const float4* src = reinterpret_cast<const float4*>(a.lut32 + q * (int64_t)a.Mpad * 256);
        for (int e = tid; e < M * 64; e += 1024) reinterpret_cast<float4*>(ft_T)[e] = src[e];
    } else
    for (int e = tid; e < M * 256; e += 1024) {
        const int m = e >> 8;
        const float* qs = qv + m * dsub;
        const float* cw = a.codebooks + (int64_t)e * dsub;
        float t = 0.0f;
        if ((dsub & 3) == 0) {
            for (int tt = 0; tt < dsub; tt += 4) {
                const float4 x = *reinterpret_cast<const float4*>(cw + tt);
                t = __fmaf_rn(qs[tt], x.x, t); t = __fmaf_rn(qs[tt + 1], x.y, t); t = __fmaf_rn(qs[tt + 2], x.z, t); t = __fmaf_rn(qs[tt + 3], x.w, t);
            }
        } else {
            for (int tt = 0; tt < dsub; tt++) t = __fmaf_rn(qs[tt], cw[tt], t);
        }
        ft_T[e] = t;
    }
    if (tid < 16) ctl[tid] = 0;
    const int64_t* ss_g = a.seg_start + q * (a.nprobe + 1);
    if (stage_probes) {
        for (int j = tid; j <= a.nprobe; j += 1024) s_ss[j] = ss_g[j];
        for (int j = tid; j < a.nprobe; j += 1024) {
            const int32_t l = a.probe_list[q * a.nprobe + j];
            s_lb[j] = l >= 0 ? a.list_base[l] : 0;
            s_d0[j] = a.probe_dis0[q * a.nprobe + j];
        }
    }
    __syncthreads();
    FT_MARK(1);
    // candidate index -> storage row (the probed list that holds it, by bisection of the query's row offsets)
    const int64_t* ss = stage_probes ? s_ss : ss_g;
    auto locate = [&](uint32_t idx, int& lo) -> int64_t {
        lo = 0; int hi = a.nprobe;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ss[mid] <= (int64_t)idx) lo = mid; else hi = mid; }
        return (stage_probes ? s_lb[lo] : a.list_base[a.probe_list[q * a.nprobe + lo]]) + ((int64_t)idx - ss[lo]);
    };
    // the kk-th largest 32-bit score word of the row's non-zero keys (ties counted with multiplicity) by an MSB-first radix walk — or,
    // when every key under a prefix is selected, that prefix with zero low bits: a lower bound of it that selects the same keys.
    // ctl[1] / ctl[4] / ctl[6] describe the last digit's bin afterwards (wanted, present, ties straddle rank kk).  Needs > kk - 1 valid keys.
    auto kth_word = [&](int kk, const uint32_t* only = nullptr) -> uint32_t {      // only: bitmap of the row entries that take part (null: all)
        uint32_t prefix = 0;
        // Round 5: the walk starts at the first byte in which the keys DIFFER.  The scores of a query's candidates sit in a narrow band —
        // their order words share the top one or two bytes — and a round over a byte all keys share is n LDS atomics on ONE counter
        // (10-16 us per round at n = 6000: the phase trace), for no information.
        if (tid == 0) { ctl[1] = kk; ctl[4] = 0; ctl[6] = 0; ctl2[0] = -1; ctl2[1] = 0; }     // ctl2[0] / [1]: min / max word (as unsigned)
        __syncthreads();
        {
            uint32_t lmin = 0xffffffffu, lmax = 0u;
            ft_for_keys(row, n, tid, [&](int c, uint64_t key) {
                if (key != 0ull && (!only || ((only[c >> 5] >> (c & 31)) & 1u))) { const uint32_t o = (uint32_t)(key >> 32); lmin = o < lmin ? o : lmin; lmax = o > lmax ? o : lmax; }
            });
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t a0 = __shfl_xor(lmin, off), a1 = __shfl_xor(lmax, off);
                lmin = a0 < lmin ? a0 : lmin; lmax = a1 > lmax ? a1 : lmax;
            }
            if (lane == 0) { atomicMin(reinterpret_cast<uint32_t*>(&ctl2[0]), lmin); atomicMax(reinterpret_cast<uint32_t*>(&ctl2[1]), lmax); }
            __syncthreads();
        }
        const uint32_t gmin = (uint32_t)ctl2[0], gmax = (uint32_t)ctl2[1];
        const uint32_t gdiff = gmin ^ gmax;
        const int shift0 = (gdiff >> 24) ? 24 : (gdiff >> 16) ? 16 : (gdiff >> 8) ? 8 : 0;
        if (shift0 < 24) prefix = gmax & ~((1u << (shift0 + 8)) - 1u);       // the bytes above the first differing one are common to all keys
        __syncthreads();
        for (int shift = shift0; shift >= 0; shift -= 8) {
            for (int i2 = tid; i2 < 256; i2 += 1024) hist[i2] = 0;
            __syncthreads();
            ft_for_keys(row, n, tid, [&](int c, uint64_t key) {
                const uint32_t o = (uint32_t)(key >> 32);
                if (key != 0ull && (!only || ((only[c >> 5] >> (c & 31)) & 1u)) && (shift == 24 || (o >> (shift + 8)) == (prefix >> (shift + 8))))
                    atomicAdd(&hist[(o >> shift) & 255u], 1);      // (at shift0 < 24 every key matches the common prefix)
            });
            __syncthreads();
            if (tid < 64) {
                const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                const int sum4 = h0 + h1 + h2 + h3;
                int suf = sum4;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_down(suf, off); if (lane + off < 64) suf += y; }
                const int remaining = ctl[1];
                const uint64_t mk = __ballot(suf >= remaining);
                const int L = 63 - __clzll((unsigned long long)mk);
                if (lane == L) {
                    int run = suf - sum4;
                    const int hb[4] = {h0, h1, h2, h3};
                    int d = 4 * L, rem = remaining, inbin = 0;
                    for (int b = 3; b >= 0; b--) {
                        if (run + hb[b] >= remaining) { d = 4 * L + b; rem = remaining - run; inbin = hb[b]; break; }
                        run += hb[b];
                    }
                    ctl[0] = d; ctl[1] = rem; ctl[4] = inbin;
                    if (shift == 0 && rem != inbin) ctl[6] = 1;     // thr is the k-th score itself and its ties straddle rank k
                }
            }
            __syncthreads();
            prefix |= (uint32_t)ctl[0] << shift;
            const bool all_in = ctl[1] == ctl[4];       // every key under this prefix is selected: no need to refine it
            __syncthreads();
            if (all_in) break;
        }
        return prefix;
    };
    // 1b. (round 5) which candidates are worth an exact score.  The row holds EVERY vector that can matter (threshold by construction:
    // tau = the SAMPLE's a_k - 2 eps), but the sample's k-th best is a weak quantile of the probed lists: at k = 1000 the row holds ~5200
    // keys.  The row's own k-th largest approximate score a_(k) is a much better bound: k candidates have a >= a_(k), hence exact score
    // >= a_(k) - eps, so a candidate with a < a_(k) - 2 eps (exact score < a_(k) - eps) cannot be among the top k, ties included.  Its
    // code bytes — 12 pieces of 8 bytes in 12 different 64-byte sectors of the rotated layout at M = 96: the phase trace put the re-score
    // at 106 of the kernel's 160 us per query, HBM-bound on ~770 fetched bytes per candidate — are never read.
    uint32_t cut = 0u, ak_word = 0u;
    float eps_q = 0.0f;
    uint32_t* bm = bm_off ? reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(ft_T) + bm_off) : nullptr;      // [ceil(cand_cap / 32)]
    {
        int cntv = 0;
        ft_for_keys(row, n, tid, [&](int, uint64_t key) { cntv += key != 0ull; });
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cntv += __shfl_xor(cntv, off);
        if (lane == 0 && cntv) atomicAdd(&ctl[7], cntv);
        __syncthreads();
        FT_MARK2(0);
        if (ctl[7] > a.k && a.qparam) {       // (without the per-query error bounds there is no cut: every key of the row is re-scored — ADVICE r5)
            const uint32_t ak = kth_word(a.k);               // a lower bound of the k-th largest approximate score word
            const float eps = reinterpret_cast<const float*>(a.qparam)[q * 4 + 2];
            float cf = ord2f(ak) - 2.0f * eps;
            cf -= fabsf(cf) * 4.8e-7f + 1e-37f;              // the subtraction's own rounding, and then some
            cut = f2ord(cf + 0.0f);
            ak_word = ak; eps_q = eps;
            __syncthreads();
            if (tid == 0) { ctl[1] = 0; ctl[4] = 0; ctl[6] = 0; }
        }
    }
    // 2. exact scores of the whole row, in place.  Round 5: a candidate's code bytes are 2 (M = 16: 1) x M / 16 loads of 8 bytes anywhere in
    // HBM; the loop used to walk the runs one after the other — six dependent round trips per candidate at M = 96, ~2 us each — and a
    // thread took its ~5 candidates (k = 1000) in turn: 106 of the kernel's 160 us per query (phase trace, profiles/r05e_*).  Now all of a candidate's
    // loads are in flight before the first table look-up, and two candidates per thread and pass where the registers allow (M <= 64).
    const int NF = M >> 6;
    constexpr int nrun = NRUN, FT_RUNS = NRUN;
    int myvalid = 0;
    struct Cand { uint64_t key; uint32_t idx; int64_t r; float dis0; uint2 lo2[FT_RUNS], hi2[FT_RUNS]; uint4 w16; };
    int stage = 0;
    // does row entry c (key) get an exact score in this stage?  (side effects: the first stage marks it in the bitmap, the last drops what is cut)
    auto admit = [&](int c, uint64_t key) -> bool {
        if (!key) return false;
        const uint32_t aw = (uint32_t)(key >> 32);
        if (stage == 1) {                 // first stage: only the candidates at or above the row's k-th approximate score
            if (aw < ak_word) return false;
            atomicOr(&bm[c >> 5], 1u << (c & 31));
            return true;
        }
        if (stage == 2 && ((bm[c >> 5] >> (c & 31)) & 1u)) return false;      // re-scored in the first stage: row[c] is exact already
        if (aw < cut) { row[c] = 0ull; return false; }      // cannot reach the top k: never re-scored
        return true;
    };
    auto fetch = [&](int c, uint32_t idx, Cand& x) {       // c < 0: nothing
        x.key = c >= 0 ? 1ull : 0ull;
        if (!x.key) return;
        x.idx = idx;
        int lo;
        x.r = locate(x.idx, lo);
        x.dis0 = stage_probes ? s_d0[lo] : a.probe_dis0[q * a.nprobe + lo];
        if (NRUN == 1) {     // M = 16: 64-vector blocks of 1 KiB: vector v's 16 bytes at v * 16, byte s = sub-quantiser (v + s) & 15
            x.w16 = *reinterpret_cast<const uint4*>(a.codes + (x.r >> 6) * 1024 + (x.r & 63) * 16);
        } else {             // 16-vector blocks (pq_exact_sum_rot_wide's pieces), 16 sub-quantisers per run in m order
#pragma unroll
            for (int run = 0; run < FT_RUNS; run++) {
                if (run >= nrun) break;
                const uint8_t* p0; const uint8_t* p1;
                if (a.codes_plain) { p0 = a.codes_plain + x.r * (a.plain_stride ? a.plain_stride : M) + run * 16; p1 = p0 + 8; }
                else pq_piece_ptrs(a.codes, x.r, M, a.CB, run, p0, p1);
                x.lo2[run] = *reinterpret_cast<const uint2*>(p0); x.hi2[run] = *reinterpret_cast<const uint2*>(p1);
            }
        }
    };
    auto score = [&](int c, Cand& x) {
        if (!x.key) return;
        float sum = 0.0f;
        if (NRUN == 1) {
            uint32_t w[4] = {x.w16.x, x.w16.y, x.w16.z, x.w16.w};
            rot16_bytes(w, (int)(x.r & 15));
#pragma unroll
            for (int j = 0; j < 16; j++) sum += ft_T[j * 256 + ((w[j >> 2] >> (8 * (j & 3))) & 255u)];
        } else {
            const int i = a.codes_plain ? 0 : (int)(x.r & 15);
#pragma unroll
            for (int run = 0; run < FT_RUNS; run++) {
                if (run >= nrun) break;
                uint32_t w[4] = {x.lo2[run].x, x.lo2[run].y, x.hi2[run].x, x.hi2[run].y};
                rot16_bytes(w, i);
                const float* Tr = ft_T + run * 16 * 256;
#pragma unroll
                for (int j = 0; j < 16; j++) sum += Tr[j * 256 + ((w[j >> 2] >> (8 * (j & 3))) & 255u)];
            }
        }
        const uint64_t nk = make_key(x.dis0 + sum, x.idx);
        row[c] = nk;
        myvalid += nk != 0ull;
    };
    // One pass over the row = an ADMISSION scan (four keys per thread and step, their loads independent) that queues the row indices
    // to re-score in LDS, then the re-score itself over the queue: every thread gets the same number of candidates, and none of its steps
    // waits for a key that turns out to be cut.  (The first form walked the row entry by entry per thread — key load, decision, piece
    // loads, table look-ups in turn: ~16 us per step whatever the decision, 113 us at k = 1000 with 2300 or with 5200 candidates to score.)
    uint2* wq = wq_off ? reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(ft_T) + wq_off) : nullptr;       // [wq_cap] {row index, candidate index}
    int32_t* wq_n = reinterpret_cast<int32_t*>(wq + wq_cap);
    auto rescore_one = [&](int c, uint32_t idx) { Cand xa; fetch(c, idx, xa); score(c, xa); };
    auto rescore_pass = [&]() {
        if (wq && tid == 0) *wq_n = 0;
        if (wq) __syncthreads();
        for (int c0 = tid; c0 < n; c0 += 4096) {
            uint64_t kk[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int c = c0 + u * 1024; kk[u] = c < n ? row[c] : 0ull; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int c = c0 + u * 1024;
                if (!admit(c, kk[u])) continue;
                int pos = wq ? atomicAdd(wq_n, 1) : wq_cap;
                if (pos < wq_cap) wq[pos] = make_uint2((uint32_t)c, key_idx(kk[u]));
                else rescore_one(c, key_idx(kk[u]));          // no queue, or full: in place
            }
        }
        if (!wq) return;
        __syncthreads();
        const int qn = *wq_n < wq_cap ? *wq_n : wq_cap;
        if (NRUN <= 4) {
            for (int i2 = tid; i2 < qn; i2 += 2048) {
                Cand xa, xb;
                const uint2 ea = wq[i2], eb = i2 + 1024 < qn ? wq[i2 + 1024] : make_uint2(0u, 0u);
                fetch((int)ea.x, ea.y, xa);
                fetch(i2 + 1024 < qn ? (int)eb.x : -1, eb.y, xb);
                score((int)ea.x, xa);
                score((int)eb.x, xb);
            }
        } else {             // M >= 96: two candidates' pieces do not fit the 128 registers of a 1024-thread workgroup
            for (int i2 = tid; i2 < qn; i2 += 1024) { const uint2 e = wq[i2]; rescore_one((int)e.x, e.y); }
        }
        __syncthreads();     // the queue is free again
    };
    if (bm && cut != 0u) {
        // Two stages (round 5): the candidates at or above a_(k) first — at least k of them; the k-th largest of THEIR exact scores, s_lo,
        // is a lower bound of the query's exact k-th best, and it sits near a_(k) where the one-stage bound had to assume a_(k) - eps.  A
        // remaining candidate with a < s_lo - eps has exact score < s_lo and is out (strictly: ties stay in): the window of the second
        // stage is ~eps wide instead of 2 eps (k = 1000 on the bench index: 6400 keys in the row, 4150 above the one-stage cut,
        // ~2600 re-scored this way).  The bitmap says which row entries hold exact keys already.
        for (int i2 = tid; i2 < ((n + 31) >> 5); i2 += 1024) bm[i2] = 0u;
        __syncthreads();
        FT_MARK2(1);
        stage = 1;
        rescore_pass();
        __syncthreads();
        FT_MARK2(2);
        const uint32_t slo = kth_word(a.k, bm);           // (a lower bound of) the k-th largest exact score word among the first stage's
        float cf = ord2f(slo) - eps_q;
        cf -= fabsf(cf) * 4.8e-7f + 1e-37f;
        const uint32_t cut2 = f2ord(cf + 0.0f);
        if (cut2 > cut) cut = cut2;
        __syncthreads();
        if (tid == 0) { ctl[1] = 0; ctl[4] = 0; ctl[6] = 0; }
        FT_MARK2(3);
        stage = 2;
        rescore_pass();
        FT_MARK2(4);
    } else {
        rescore_pass();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) myvalid += __shfl_xor(myvalid, off);
    if (lane == 0 && myvalid) atomicAdd(&ctl[2], myvalid);
    __syncthreads();        // workgroup-scope release / acquire: every thread sees the re-scored row
    FT_MARK(2);
    // 3. threshold = the k-th largest 32-bit score word (ties counted with multiplicity)
    const int V = ctl[2];
    if (a.stat && tid == 0) atomicAdd(a.stat, (unsigned long long)V);
    uint32_t thr = 1u;      // fewer than k valid keys: take every valid one
    if (V > a.k) thr = kth_word(a.k);
    FT_MARK(3);
    // 4. collect.  Usually the keys at or above the threshold are just k (+ a few ties) and all of them go to the sort.  PQ codes
    // are coarse, though: vectors with IDENTICAL codes in one list have bit-equal scores (at M = 16 on the bench mixture whole
    // data clusters do: 40 % of adjacent results tie, up to 16 000 vectors at one score), and the order among them is id
    // ascending.  When the ties of the k-th score do not fit the sort, their ids go to a scratch row, the t-th smallest id is
    // found by a second radix walk (ids are distinct), and only the t ties at or below it join the keys above the threshold.
    const bool straddle = ctl[6] != 0;
    const int t_need = ctl[1], g_ties = ctl[4];                      // (straddle only) ties wanted / ties present
    const bool tie_select = straddle && (a.k - t_need) + g_ties > P;
    for (int i2 = tid; i2 < P; i2 += 1024) { sord[i2] = 0u; sid[i2] = INT64_MAX; }
    __syncthreads();
    uint64_t* tws = tie_ws + q * (int64_t)cand_cap;
    ft_for_keys(row, n, tid, [&](int, uint64_t key) {
        const uint32_t o = (uint32_t)(key >> 32);
        if (key == 0ull || o < thr) return;
        int lo;
        const int64_t r = locate(key_idx(key), lo);
        const int64_t id = a.ids ? a.ids[r] : r;
        if (tie_select && o == thr) {
            tws[atomicAdd(&ctl[5], 1)] = ~((uint64_t)id ^ 0x8000000000000000ull);      // larger = smaller id; never 0 (id != INT64_MAX)
        } else {
            const int pos = atomicAdd(&ctl[3], 1);
            if (pos < P) { sord[pos] = o; sid[pos] = id; }
        }
    });
    __syncthreads();
    if (tie_select) {
        auto key_at = [&](int i2) -> uint64_t { return tws[i2]; };
        radix_topk_wg<1024, true>(key_at, g_ties, t_need, tsel, hist, ctl2);      // tsel[0]: the t-th largest of the flipped ids (0: take all)
        const uint64_t kth = tsel[0];
        for (int c = tid; c < g_ties; c += 1024) {
            const uint64_t f = tws[c];
            if (f < kth) continue;
            const int pos = atomicAdd(&ctl[3], 1);
            if (pos < P) { sord[pos] = thr; sid[pos] = (int64_t)((~f) ^ 0x8000000000000000ull); }
        }
        __syncthreads();
    }
    if (ctl[3] > P) {        // cannot happen with the tie selection; kept as a guard: the exact re-run (full-row selection) settles it
        if (tid == 0) a.uncertain[q] = 2 | 4 | (int)((unsigned)(ctl[3] > 0xfffff ? 0xfffff : ctl[3]) << 8);    // bit 2 + the count: diagnostics
        return;
    }
    FT_MARK(4);
    int ns = 2;              // sort only the power of two that holds the keys
    while (ns < ctl[3]) ns <<= 1;
    for (int size = 2; size <= ns; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (ns >> 1); t += 1024) {
                const int i2 = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
                const int j2 = i2 + stride;
                const bool desc = ((i2 & size) == 0);
                const uint32_t oi = sord[i2], oj = sord[j2];
                const int64_t ii = sid[i2], ij = sid[j2];
                const bool i_worse = (oi < oj) || (oi == oj && ii > ij);
                if (i_worse == desc) { sord[i2] = oj; sord[j2] = oi; sid[i2] = ij; sid[j2] = ii; }
            }
            __syncthreads();
        }
    }
    FT_MARK(5);
    for (int j = tid; j < a.k; j += 1024) {
        const bool valid = j < ns && sord[j] != 0u && sid[j] != INT64_MAX;
        a.D[q * a.k + j] = valid ? ord2f(sord[j]) : -__builtin_inff();
        a.I[q * a.k + j] = valid ? sid[j] : -1;
    }
    if (tid == 0) a.uncertain[q] = 0;
    FT_MARK(6);
}
// Row-major copy of the codes for the finalize kernels (large K'): one thread per (storage row, 16-sub-quantiser run) reads the run's piece
// of the scan layout, rotates it back into m order and writes 16 contiguous bytes.
__global__ __launch_bounds__(256) void k_pq_plain_rows(const uint8_t* __restrict__ codes, int64_t nrows, int M, int CB, uint8_t* __restrict__ out, int stride) {
    const int nrun = M >> 4;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nrows * nrun) return;
    const int64_t row = e / nrun;
    const int run = (int)(e - row * nrun);
    const uint8_t* p0; const uint8_t* p1;
    pq_piece_ptrs(codes, row, M, CB, run, p0, p1);
    const uint2 lo = *reinterpret_cast<const uint2*>(p0), hi = *reinterpret_cast<const uint2*>(p1);
    uint32_t w[4] = {lo.x, lo.y, hi.x, hi.y};
    rot16_bytes(w, (int)(row & 15));
    *reinterpret_cast<uint4*>(out + row * stride + run * 16) = make_uint4(w[0], w[1], w[2], w[3]);
}
// row stride of the row-major code copy: a row never straddles a 128-byte line (M = 96 -> 128: a candidate is ONE line from HBM instead of 1.75 on
// average — the large-k finalize kernels run at the HBM's random-line rate)
int pq_plain_stride(int M) {
    if (M >= 128) return (M + 127) / 128 * 128;
    int s2 = 16; while (s2 < M) s2 <<= 1;
    return s2;
}
void launch_pq_plain_rows(const uint8_t* codes, int64_t nrows, int M, int CB, uint8_t* out, hipStream_t st) {
    if (nrows <= 0) return;
    const int64_t units = nrows * (M >> 4);
    hipLaunchKernelGGL(k_pq_plain_rows, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, codes, nrows, M, CB, out, pq_plain_stride(M));
}

// sort capacity of k_pq_final_tab for this (M, k), 0 when the kernel does not apply (layout, or the table + sort do not fit the LDS)
int pq_final_tab_capacity(int M, int CB, int k) {
    if (!pq_rot_family(CB) || !pq_rot_applies(M)) return 0;
    const auto fits = [&](int P) { return (size_t)M * 1024 + (size_t)P * 12 + 16 + (256 + 16) * 4 + 64 <= (size_t)160 * 1024; };
    int P = 64; while (P < k + 16) P <<= 1;
    if (fits(P)) return P;
    P = 64; while (P < k) P <<= 1;
    return fits(P) ? P : 0;
}
void launch_pq_final_tab(const FinalizeArgs& a, uint64_t* cand, int cand_cap, uint64_t* tie_ws, hipStream_t st) {
    if (a.nq <= 0) return;
    const int P = pq_final_tab_capacity(a.M, a.CB, a.k);
    size_t shm = (size_t)a.M * 1024 + (size_t)P * 12 + 16 + (256 + 16) * 4 + 64;
    const size_t probes = (size_t)(a.nprobe + 1) * 8 + (size_t)a.nprobe * 12 + 16;
    const int stage_probes = shm + probes <= (size_t)160 * 1024 ? 1 : 0;
    if (stage_probes) shm += probes;
    // bitmap of the row entries the first re-score stage has settled (two-stage cut): one bit per candidate slot, behind everything else
    int bm_off = 0;
    {
        const size_t off = (shm + 15) / 16 * 16, need = ((size_t)cand_cap + 31) / 32 * 4;
        if (a.qparam && off + need <= (size_t)160 * 1024) { bm_off = (int)off; shm = off + need; }
    }
    // work queue of the re-score (row index + candidate index per entry, then the counter): as many entries as the LDS has left, 512 .. 4096
    int wq_off = 0, wq_cap = 0;
    {
        const size_t off = (shm + 15) / 16 * 16;
        size_t room = off + 64 < (size_t)160 * 1024 ? ((size_t)160 * 1024 - off - 64) / 8 : 0;
        if (room > 4096) room = 4096;
        if (room >= 512) { wq_off = (int)off; wq_cap = (int)room; shm = off + room * 8 + 16; }
    }
    static DevSize attr;
    auto kern = a.M == 16 ? k_pq_final_tab<1> : a.M == 32 ? k_pq_final_tab<2> : a.M == 64 ? k_pq_final_tab<4> : a.M == 96 ? k_pq_final_tab<6> : k_pq_final_tab<8>;
    attr.grow(shm, [&] {
        (void)hipFuncSetAttribute((const void*)k_pq_final_tab<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_pq_final_tab<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_pq_final_tab<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_pq_final_tab<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_pq_final_tab<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL(kern, dim3((unsigned)a.nq), dim3(1024), shm, st, a, cand, cand_cap, P, tie_ws, stage_probes, bm_off, wq_off, wq_cap);
}

void launch_finalize(const FinalizeArgs& a0, hipStream_t st) {
    if (a0.nq <= 0) return;
    FinalizeArgs a = a0;
    size_t shm = (size_t)a.KP * (8 + 8 + 4);
    int waves = std::min(4, std::max(1, (a.KP + 63) / 64));
    // a handful of queries leave the chip idle: 16 waves per query and the parallel table-entry form of the IVF-PQ re-score
    const size_t base20 = (((size_t)a.KP * 20 + 15) / 16) * 16;
    const size_t par_shm = ((base20 + (size_t)a.KP * (a.M + 1) * 4 + 15) / 16) * 16 + (size_t)a.KP * a.M;      // entries + the staged code bytes
    a.par_entries = (a.kind == KIND_IVFPQ && a.pq_rescore && pq_rot_family(a.CB) && !a.lut32 && a.dsub == 8 && a.nq <= 64 && par_shm <= 64 * 1024) ? 1 : 0;
    a.rank_sort = (a.KP <= FIN_RANK_MAX && a.nq <= 64) ? 1 : 0;     // fewer barriers, more instructions: a latency trade, not a throughput one
    if (a.rank_sort) shm = base20 + (size_t)a.KP * 12 + 16;       // the second copy shares the table-entry region (used earlier)
    if (a.par_entries) { shm = std::max(shm, par_shm); waves = 16; }
    else if (a.kind != KIND_IVFPQ && a.nq <= 64) waves = std::min(16, std::max(waves, a.KP / 4));   // one wave per candidate re-score
    else if (a.KP >= 1024) waves = 16;     // the reference's n_docs = 1000: 2048 candidates per query to re-score and sort (4 waves: 3.2 ms per 1024 queries)
    a.probe_lds_off = 0;
    if (a.kind != KIND_FLAT && a.nprobe <= 1024) {      // the probe table behind everything else (20 bytes per probe)
        a.probe_lds_off = (int)((shm + 15) / 16 * 16);
        shm = (size_t)a.probe_lds_off + (size_t)(a.nprobe + 1) * 8 + (size_t)a.nprobe * 12 + 16;
    }
    static DevSize big;
    if (shm > 48 * 1024) big.grow(shm, [&] { hipFuncSetAttribute((const void*)k_finalize, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); });
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)a.nq), dim3(64 * waves), shm, st, a);
}

// ---------------------------------------------------------------------------------------
// Multi-shard merge (src/search.py:362-367): concatenate the shards' top-k in shard order and
// stable-sort by score; ties keep the earlier shard, then the within-shard order.
// ---------------------------------------------------------------------------------------
// Inputs are addressed as D32[sh * d_sstride + (q * k + j) * d_estride] (32-bit words) and
// I[sh * i_sstride + q * k + j]: plain [nshards, nq, k] arrays, or the packed [nshards, 2, nq, k] int64
// buffer an all-gather of rsx_pack_topk outputs produces (score bits in the low word).
__global__ __launch_bounds__(64) void k_merge_topk(int nshards, int64_t nq, int k, int metric, const float* D,
                                                   int64_t d_sstride, int d_estride, const int64_t* I, int64_t i_sstride,
                                                   float* Do, int64_t* Io, int NP) {
    extern __shared__ __attribute__((aligned(16))) uint64_t mg_buf[];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    const int n = nshards * k;
    for (int p = lane; p < NP; p += 64) {
        uint64_t key = 0;
        if (p < n) {
            int sh = p / k, j = p % k;
            const int64_t e = q * k + j;
            if (I[sh * i_sstride + e] >= 0) {
                float s = D[sh * d_sstride + e * d_estride];
                s = (metric == 0 ? s : 0.0f - s) + 0.0f;
                if (s == s) key = ((uint64_t)f2ord(s) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)p);
            }
        }
        mg_buf[p] = key;
    }
    __syncthreads();
    bitonic_sort_desc(mg_buf, NP, lane);
    for (int j = lane; j < k; j += 64) {
        uint64_t key = mg_buf[j];
        if (key) {
            int p = (int)key_idx(key);
            int sh = p / k, jj = p % k;
            const int64_t e = q * k + jj;
            Do[q * k + j] = D[sh * d_sstride + e * d_estride];
            Io[q * k + j] = I[sh * i_sstride + e];
        } else {
            Do[q * k + j] = metric == 0 ? -__builtin_inff() : __builtin_inff();
            Io[q * k + j] = -1;
        }
    }
}

static void launch_merge_strided(int nshards, int64_t nq, int k, int metric, const float* D, int64_t d_sstride, int d_estride,
                                 const int64_t* I, int64_t i_sstride, float* Do, int64_t* Io, hipStream_t st) {
    if (nq <= 0) return;
    int NP = 64;
    while (NP < nshards * k) NP <<= 1;
    size_t shm = (size_t)NP * 8;
    if (shm > 48 * 1024) hipFuncSetAttribute((const void*)k_merge_topk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(k_merge_topk, dim3((unsigned)nq), dim3(64), shm, st, nshards, nq, k, metric, D, d_sstride, d_estride, I,
                       i_sstride, Do, Io, NP);
}
// Any shard count x k (round 5; the reference backends' default is k = 4096, src/indicies/flat.py:138, and eight ranks of it are
// 32768 keys per query): one launch while nshards * k <= 16384 keys fit the kernel's LDS sort, else ROUNDS over groups of
// G = max(2, 8192 / k) consecutive blocks.  The rule (score, then earlier shard, then in-shard rank) survives the rounds: a group's
// output is already in that order and the groups stay in shard order, so a tie between two groups resolves to the earlier shards
// and a tie inside a group to the earlier position — the one-launch result, bit for bit (tests/test_gpu_merge_rounds.py).  The
// round buffers are stream-ordered allocations on the caller's stream: no host synchronisation.  Returns false when k > 8192.
static bool merge_any(int nshards, int64_t nq, int k, int metric, const float* D, int64_t d_sstride, int d_estride,
                      const int64_t* I, int64_t i_sstride, float* Do, int64_t* Io, hipStream_t st) {
    if (nq <= 0) return true;
    if ((int64_t)nshards * k <= 16384) { launch_merge_strided(nshards, nq, k, metric, D, d_sstride, d_estride, I, i_sstride, Do, Io, st); return true; }
    if (k > 8192) return false;
    const int G = 8192 / k < 2 ? 2 : 8192 / k;
    const size_t blk = (size_t)nq * k;
    const int g0 = (nshards + G - 1) / G, g1 = (g0 + G - 1) / G;
    float* tD[2] = {nullptr, nullptr}; int64_t* tI[2] = {nullptr, nullptr};
    auto need = [&](int which, int groups) {
        if (tD[which]) return true;
        return hipMallocAsync((void**)&tD[which], (size_t)groups * blk * 4, st) == hipSuccess &&
               hipMallocAsync((void**)&tI[which], (size_t)groups * blk * 8, st) == hipSuccess;
    };
    bool ok = need(0, g0);
    if (ok) for (int g = 0; g < g0; g++) {          // round 1 reads the caller's (possibly packed) layout
        const int n = nshards - g * G < G ? nshards - g * G : G;
        launch_merge_strided(n, nq, k, metric, D + (int64_t)g * G * d_sstride, d_sstride, d_estride, I + (int64_t)g * G * i_sstride, i_sstride,
                             tD[0] + (size_t)g * blk, tI[0] + (size_t)g * blk, st);
    }
    int cur = g0, src = 0;
    while (ok && cur > G) {
        const int groups = (cur + G - 1) / G;
        ok = need(src ^ 1, src == 0 ? g1 : g0);
        if (!ok) break;
        for (int g = 0; g < groups; g++) {
            const int n = cur - g * G < G ? cur - g * G : G;
            launch_merge_strided(n, nq, k, metric, tD[src] + (size_t)g * G * blk, (int64_t)blk, 1, tI[src] + (size_t)g * G * blk, (int64_t)blk,
                                 tD[src ^ 1] + (size_t)g * blk, tI[src ^ 1] + (size_t)g * blk, st);
        }
        cur = groups; src ^= 1;
    }
    if (ok) launch_merge_strided(cur, nq, k, metric, tD[src], (int64_t)blk, 1, tI[src], (int64_t)blk, Do, Io, st);
    for (int w = 0; w < 2; w++) { if (tD[w]) (void)hipFreeAsync(tD[w], st); if (tI[w]) (void)hipFreeAsync(tI[w], st); }
    if (!ok) (void)hipGetLastError();
    return ok;
}
bool launch_merge_topk(int nshards, int64_t nq, int k, int metric, const float* D, const int64_t* I, float* Do,
                       int64_t* Io, hipStream_t st) {
    return merge_any(nshards, nq, k, metric, D, nq * k, 1, I, nq * k, Do, Io, st);
}
// packed: [nshards, 2, nq, k] int64 — plane 0 = score bits (low word), plane 1 = ids
bool launch_merge_packed(int nshards, int64_t nq, int k, int metric, const int64_t* packed, float* Do, int64_t* Io,
                         hipStream_t st) {
    return merge_any(nshards, nq, k, metric, reinterpret_cast<const float*>(packed), 4 * nq * k, 2, packed + nq * k,
                     2 * nq * k, Do, Io, st);
}

// Merge for the single-process multi-GPU handle (rsx_sharded_create): the shards are pieces of ONE logical index, so the
// merged order is the single index's canonical order — score descending, ties by id ascending — not the reference's
// shard-order rule above (which rsx_merge_topk keeps for its per-shard indexes).  One workgroup per query, bitonic sort of
// (ordered score, id) pairs in LDS.
__global__ __launch_bounds__(256) void k_merge_topk_byid(int nshards, int64_t nq, int k, int metric, const float* D, const int64_t* I,
                                                         float* Do, int64_t* Io, int NP) {
    extern __shared__ __attribute__((aligned(16))) uint64_t mb_buf[];
    int64_t* sid = reinterpret_cast<int64_t*>(mb_buf);
    uint32_t* sord = reinterpret_cast<uint32_t*>(sid + NP);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int64_t q = blockIdx.x;
    const int n = nshards * k;
    for (int p = tid; p < NP; p += nt) {
        uint32_t ord = 0; int64_t id = INT64_MAX;
        if (p < n) {
            const int sh = p / k, j = p - sh * k;
            const int64_t e = ((int64_t)sh * nq + q) * k + j;
            const int64_t ii = I[e];
            if (ii >= 0) {
                float sc = D[e];
                sc = (metric == 0 ? sc : 0.0f - sc) + 0.0f;
                if (sc == sc) { ord = f2ord(sc); id = ii; }
            }
        }
        sord[p] = ord; sid[p] = id;
    }
    __syncthreads();
    for (int size = 2; size <= NP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (NP >> 1); t += nt) {
                const int i = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
                const int j = i + stride;
                const bool desc = ((i & size) == 0);
                const uint32_t oi = sord[i], oj = sord[j];
                const int64_t ii = sid[i], ij = sid[j];
                const bool i_worse = (oi < oj) || (oi == oj && ii > ij);
                if (i_worse == desc) { sord[i] = oj; sord[j] = oi; sid[i] = ij; sid[j] = ii; }
            }
            __syncthreads();
        }
    }
    for (int j = tid; j < k; j += nt) {
        const bool valid = sord[j] != 0 && sid[j] != INT64_MAX;
        float sc = valid ? ord2f(sord[j]) : -__builtin_inff();
        if (metric != 0) sc = valid ? (0.0f - sc) : __builtin_inff();
        Do[q * k + j] = sc;
        Io[q * k + j] = valid ? sid[j] : -1;
    }
}
void launch_merge_topk_byid(int nshards, int64_t nq, int k, int metric, const float* D, const int64_t* I, float* Do,
                            int64_t* Io, hipStream_t st) {
    if (nq <= 0) return;
    int NP = 64;
    while (NP < nshards * k) NP <<= 1;
    const size_t shm = (size_t)NP * 12;
    if (shm > 48 * 1024) hipFuncSetAttribute((const void*)k_merge_topk_byid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(k_merge_topk_byid, dim3((unsigned)nq), dim3(256), shm, st, nshards, nq, k, metric, D, I, Do, Io, NP);
}

// one rank's (D, I) -> the packed [2, nq, k] int64 block it contributes to the all-gather; ids get the shard's offset
__global__ void k_pack_topk(int64_t n, const float* D, const int64_t* I, int64_t id_offset, int64_t* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = (int64_t)(uint64_t)__float_as_uint(D[i]);
    const int64_t id = I[i];
    out[n + i] = id >= 0 ? id + id_offset : id;
}
void launch_pack_topk(int64_t n, const float* D, const int64_t* I, int64_t id_offset, int64_t* out, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pack_topk, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, D, I, id_offset, out);
}

}  // namespace rsx
