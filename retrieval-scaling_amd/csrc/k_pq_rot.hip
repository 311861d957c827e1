// k_pq_rot.hip — IVF-PQ scans of the ROTATED code layout (rsx_internal.h: CB = 0; M in {16, 32, 64, 96, 128}).
// Reference call site: IndexIVFPQ.search, src/indicies/ivf_pq.py:229-232 (IP, by_residual):
//     score(q, v) = <q, centroid(list(v))> + sum_m T[q][m][code_v[m]].
//
//  k_pq_scan_rot       : THE hot kernel of the bench configuration.  Same work decomposition, tables, approximate-score
//                        expression, candidate keys and certificate as k_pq_scan8 (k_pq.hip) — what changes is how a
//                        table entry is reached and who adds it up:
//    * k_pq_scan8 gathers T[code][m] with every lane on the SAME m: the bank is a function of the random code and a
//      wave-level ds_read_b32 pays ~3.5-way conflicts (PMC: 45 % of its LDS cycles).  Here the bytes of a 16-vector block
//      are stored so that lane (g, i) of a wave reaches, at step s, sub-quantiser m = 16 g + ((i + s) & 15): the 32 lanes
//      of a half-wave always hold 32 different m % 32, the table row of a code is 256 bytes = [m % 64] dwords, hence
//      bank = m % 32 and EVERY gather is conflict-free, whatever the codes are (2 LDS cycles instead of ~7).
//    * the address (code << 8) | rot_byte(lane, s) is ONE v_perm_b32 (the rotation bytes are per-lane constants).
//    * the gathered dword = four queries' int8 table entries is an A operand of v_mfma_i32_16x16x64_i8 against a constant
//      one-hot B (column n picks byte n): the matrix core adds 16 gathers x 16 vectors x 4 queries per instruction as
//      exact integers, the VALU does no accumulation at all.  (This is an adder tree, not a GEMM reshaping of the search:
//      the contraction is over look-ups the LDS has already performed.)
//    * candidates: an integer compare of the MFMA result against a per-(query, list) threshold found by bisection on the
//      scan's own fp32 score expression (exactly the keys that can beat the query's threshold key); the rare survivors
//      queue in LDS and are appended with one reservation per work item and query.
//    Bound: HBM (each code byte is read once per query GROUP; LDS 2 clk, VALU 1 op, MFMA 1/4 op per gather stay under
//    it).  Measured stand-alone (tools/proto/rot_gather.hip): 5.9 TB/s of code bytes on MI355X.
//  k_pq_scan_rot_exact : per-(query, list) exact scan (fp32 table in LDS, sequential sums in m order = oracle bits) for the
//                        certificate fallback, scan_kernel = 1/2 and pq_fast = 0 on this layout.  Byte loads through
//                        pq_code_addr: slow and simple on purpose, it runs for ~1e-4 of the queries.
#include <climits>
#include <cstdlib>

#include "rsx_internal.h"

namespace rsx {

typedef int v4i __attribute__((ext_vector_type(4)));

#ifndef ROT_DEPTH
#define ROT_DEPTH 4
#endif
#ifndef SL8_NB
#define SL8_NB 16              // k_pq_scan_sl8: blocks of 32 vectors per wave and sub-tile (their partial sums are parked in 4 VGPRs each)
#endif
#ifndef SL8_VAR
#define SL8_VAR 0
#endif
#ifndef SL8_AUX
#define SL8_AUX 0             // ... cache policy of the code loads (2 = non-temporal)
#endif
#ifndef SL8_PR
#define SL8_PR 1              // ... blocks a wave keeps one issue priority for
#endif
#ifndef SL8_RD
#define SL8_RD 2              // ... and 1 KiB code loads in flight per wave
#endif
constexpr int ROT_D = ROT_DEPTH;   // 16-vector code blocks in flight per wave (16 M bytes each); must divide the tile's blocks per wave
#ifndef ROT_CW
#define ROT_CW 4              // items (waves) per workgroup of k_pq_rot_compact
#endif

// pacing word: explicit LDS-space accesses (a generic volatile pointer compiles to flat_load, whose in-order vmcnt wait would
// drain the wave's outstanding code loads at every loop iteration)
__device__ __forceinline__ uint32_t lds_rd32_volatile(uint32_t addr) {
    return *reinterpret_cast<const volatile __attribute__((address_space(3))) uint32_t*>((uintptr_t)addr);
}
__device__ __forceinline__ void lds_wr32(uint32_t addr, uint32_t v) {
    *reinterpret_cast<volatile __attribute__((address_space(3))) uint32_t*>((uintptr_t)addr) = v;
}
// refresh the sibling-progress slots: an LDS-DMA load (lane i -> slot i).  Issued from inline assembly on purpose: the
// compiler then knows neither a destination register nor an LDS write, so it inserts no vmcnt wait for it — with the builtin
// (whose LDS write may alias the raw-address table gathers) or with a load into a register it drains vmcnt, i.e. the wave's
// whole code prefetch, wherever the value is used.  sib_landed() is the explicit wait.  sc0: bypass this CU's L1 — the
// siblings run on the same XCD, the words only have to be coherent at its L2.
__device__ __forceinline__ void sib_refresh(const uint32_t* gp, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off sc0" : : "v"(gp), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void sib_landed() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
// largest progress word among the 8 slots that belongs to a sibling on its FIRST pass (0 = not started, bit 30 = wrapped
// around, 0x7fffffff = done / not a sibling); 0 when there is none
__device__ __forceinline__ uint32_t sib_max8_first_pass(uint32_t a) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 x = *reinterpret_cast<const volatile __attribute__((address_space(3))) u4*>((uintptr_t)a);
    const u4 y = *reinterpret_cast<const volatile __attribute__((address_space(3))) u4*>((uintptr_t)(a + 16));
    const uint32_t v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
    uint32_t m = 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) { const uint32_t t = v[i] < 0x40000000u ? v[i] : 0u; m = t > m ? t : m; }
    return m;
}
__device__ __forceinline__ uint32_t lds_rd32(uint32_t addr) {
    // raw LDS address: the kernels below declare no static LDS, so the dynamic segment starts at 0
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>((uintptr_t)addr);
}
typedef unsigned int rot_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rot_v2u lds_rd64(uint32_t addr) {      // ds_read_b64: 2 LDS cycles per wave, bank = (addr / 4) % 64
    return *reinterpret_cast<const __attribute__((address_space(3))) rot_v2u*>((uintptr_t)addr);
}

// registers of rotation bytes a phase needs: plane 0 packs 4 per dword, planes >= 1 pack 3 + the plane byte
__host__ __device__ constexpr int rot_nreg(int plane, int steps) { return plane == 0 ? steps / 4 : (steps + 2) / 3; }

// ---------------------------------------------------------------------------------------
// Work items, resolved AHEAD of the scan by a trivially parallel kernel: everything a workgroup needs to start a
// (list, tile, 4-query group) item — list extent, the four queries, their per-(query, list) score parameters, the
// threshold key and the integer threshold derived from it — sits in ONE 176-byte record.  With a single workgroup
// resident per CU (the table fills the LDS) every dependent load at the start of an item is exposed latency; the scan
// kernel prefetches the next record while it scans the current item.
// ---------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) PQRotItem {
    int32_t l, tile, np, pad0;
    int64_t len, base_row;
    int32_t q[4];
    float dis0[4], scale[4], bias[4];
    int64_t off[4];        // filtered: the query's row column of the list (candidate index space); unfiltered: offset into temp
    uint64_t tau[4];       // threshold key (0: none)
    int32_t cinit[4];      // MFMA accumulator init = -(integer threshold), see k_pq_scan_rot
};
static_assert(sizeof(PQRotItem) == 176, "PQRotItem is copied as 11 x 16 bytes");

// ngq = records per work item: 1, or — M = 16, filtered scan (k_pq_scan_rot16) — 4: a work item is then a group of up to 16 probing
// queries of a list tile, described by four consecutive 4-query records (a record without queries has np = 0).
template <int M, bool FILTER>
__global__ __launch_bounds__(256) void k_pq_rot_items(PQScan8Args A, PQRotItem* items, uint32_t* xcd_ctr, uint32_t* prog, int ngq, int lds_lists) {
    const PQScanArgs& a = A.b;
    const int rec = blockIdx.x * 256 + threadIdx.x;
    if (rec < 8) xcd_ctr[rec * 32] = 0u;     // the scan's per-XCD work counters (one per 128-byte line)
    const int ti = *A.total_items;
    const int item = rec / ngq, h = rec - item * ngq;
    // the item -> list bisection walks item_off: 12 dependent L2 round trips per thread on global memory (6 of the kernel's 17 us) — on a copy
    // in LDS when the launcher granted one (lds_lists = nlist + 1 words)
    extern __shared__ int32_t ri_off[];
    const int32_t* ioff = A.item_off;
    if (lds_lists > 0) {
        for (int i = threadIdx.x; i < lds_lists; i += 256) ri_off[i] = A.item_off[i];
        __syncthreads();
        ioff = ri_off;
    }
    if (item >= ti) return;
    int lo = 0, hi = A.nlist;   // largest l with item_off[l] <= item
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ioff[mid] <= item) lo = mid; else hi = mid; }
    const int ng = A.group_off[lo + 1] - A.group_off[lo];
    const int r = item - ioff[lo];
    const int tile = r / ng, gi = r - tile * ng;
    const int cnt = A.pair_off[lo + 1] - A.pair_off[lo];
    const int first = 4 * (gi * ngq + h);                 // this record's first pair within the list's sorted pairs
    int npr = cnt - first; if (npr > 4) npr = 4; if (npr < 0) npr = 0;
    const int pair0 = A.pair_off[lo] + (npr > 0 ? first : 0);
    PQRotItem d;
    d.l = lo; d.tile = tile; d.np = npr; d.pad0 = 0;
    // the item's FAMILY = the ng query groups of one list tile, adjacent in the item order: [item - gi, item - gi + ng)
    const int fam = ((gi < 0x3fff ? gi : 0x3fff) << 4) | ((ng < 0x3fff ? ng : 0x3fff) << 18);
    if (h == 0) prog[item] = 0u;
    d.len = a.list_len[lo]; d.base_row = a.list_base[lo];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int pi = A.pairs_sorted[pair0 + (k < npr ? k : 0)];
        const int64_t q = pi / a.nprobe;
        const PQQParam p = A.qp[q];
        const int64_t col = a.seg_start[q * (a.nprobe + 1) + (pi - (int)q * a.nprobe)];
        const float dis0 = a.probe_dis0[pi];
        d.q[k] = (int32_t)q; d.dis0[k] = dis0; d.scale[k] = p.scale; d.bias[k] = p.bias;
        if (FILTER && A.qitems && k < npr && tile < A.qitems_tmax) A.qitems[(int64_t)pi * A.qitems_tmax + tile] = rec * 4 + k;
        d.off[k] = FILTER ? col : q * a.tstride + col;
        const uint64_t tau = FILTER ? A.tau_key[q * A.tau_stride] : 0ull;
        d.tau[k] = tau;
        // threshold on the integer sum: the smallest S whose score dis0 + fma(scale, S, bias) (the expression the survivors are
        // scored with, monotone in S) reaches the threshold key's score.  The MFMA accumulates C = S - 128 M starting from
        // cinit = -(that threshold - 128 M): the block's result is then >= 0 exactly for the survivors.
        int thr = INT_MIN;
        if (k >= npr) thr = INT_MAX;
        else if (FILTER && tile == 0 && A.excl && A.excl[q] == (uint16_t)(0x8000 | (pi - (int)q * a.nprobe))) thr = INT_MAX;   // emitted by the pre-pass
        else if (FILTER && tau != 0ull) {
            const float ts = key_score(tau);
            // p.pad = the largest sum this query's table can give any code vector (<= 255 M): nothing in this list can
            // reach the threshold when even that falls short
            int smax = (int)p.pad; if (smax <= 0 || smax > 255 * M) smax = 255 * M;
            if (!(dis0 + __fmaf_rn(p.scale, (float)smax, p.bias) >= ts)) thr = INT_MAX;
            else {
                int b0 = 0, b1 = smax;
                while (b0 < b1) {
                    const int mid = (b0 + b1) >> 1;
                    if (dis0 + __fmaf_rn(p.scale, (float)mid, p.bias) >= ts) b1 = mid; else b0 = mid + 1;
                }
                thr = b0 - 128 * M;
            }
        }
        d.cinit[k] = thr == INT_MAX ? -(1 << 30) : thr == INT_MIN ? (1 << 30) : -thr;
    }
    // opt-in pair pruning (rsx_set_param "pq_prune"): no query of the group can produce a survivor here -> the scan skips
    // the item (no table staging, no gathers).  Exact: the bound is on the very integer sums the scan would compute.
    if (FILTER && A.prune && ngq == 1 && d.cinit[0] == -(1 << 30) && d.cinit[1] == -(1 << 30) && d.cinit[2] == -(1 << 30) && d.cinit[3] == -(1 << 30))
        d.l = -2;
    d.np |= fam;
    items[rec] = d;
}

// ---------------------------------------------------------------------------------------
// The scan.  PERSISTENT workgroups (one per CU: the table owns the LDS): workgroup b serves XCD b % 8 and walks that XCD's
// contiguous range of the list-major item order with stride gridDim / 8, so the query groups of one list tile still run
// on one XCD close together in time (one HBM fetch per tile).  Per item: [barrier] stage the group's table, copy out the
// PREVIOUS item's survivor queues (their global reservations were issued before the staging and have landed by now),
// [barrier] scan.  Nothing at an item boundary waits on a dependent global load except the table rows themselves.
// ---------------------------------------------------------------------------------------
#ifdef RSX_MEASURE
// tools/ builds only: per-item trace of the scan (start / end wall clock in 10 ns ticks, workgroup, family) for the
// sibling start-skew measurement (tools/exp_trace.py); the shipped library has neither the array nor the stores
__device__ uint64_t g_rot_trace[4 * 65536];
__device__ uint32_t g_rot_wave[16 * 16384];       // per (item < 16384, wave): scan-loop duration in 10 ns ticks
extern "C" int rsx_debug_rot_trace(uint64_t* out, int n_items) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rot_trace), (size_t)n_items * 32) == hipSuccess ? 0 : -1;
}
extern "C" int rsx_debug_rot_wave(uint32_t* out, int n_items) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rot_wave), (size_t)n_items * 64) == hipSuccess ? 0 : -1;
}
#endif

// NQ = 1: M = 16 (NF = NH = 0): 64-vector blocks, lane (g, i) = vector 16 g + i, every MFMA column used (column 4 g + q = vector group g, query q)
// G = 2 (round 6, M = 64, filtered): EIGHT queries per pass over a list tile.  A work item carries two 4-query records (k_pq_rot_items,
// ngq = 2) and a table entry is 8 bytes — the eight queries' int8 values of one (code, m) — fetched by ONE ds_read_b64: 2 LDS cycles
// per wave like the ds_read_b32 of the 4-query form (256 B/clk against 128), one v_perm for the address, so the look-up side costs
// the same per (vector, sub-quantiser) for twice the queries; only the MFMA count doubles (two gathers fill an A operand instead of
// four).  Table image: [h = m >> 5][code][m & 31] x 8 B = two 64 KiB halves with 256-byte rows; lane (g, i) looks up
// m = 16 g + ((i + s) & 15), i.e. half h = g >> 1 (a per-lane constant that rides in the rotation registers' fourth byte) and
// slot 16 (g & 1) + ((i + s) & 15): the 32 lanes of a half-wave read 32 different 8-byte slots of their rows = all 64 banks once.
template <int NF, int NH, bool FILTER, int NQ = 0, int G = 1>
__global__ __launch_bounds__(1024) void k_pq_scan_rot(PQScan8Args A, const PQRotItem* __restrict__ items, uint64_t* __restrict__ log_keys,
                                                      uint2* __restrict__ seg_desc, uint32_t* xcd_ctr, uint32_t* prog, int log_cap, int bpw, int pace_arg, int var_arg) {
#ifdef RSX_MEASURE
    const int var = var_arg;        // tools/ builds only: cost-split variants (skip staging / scan / survivor path)
#else
    constexpr int var = 0;          // the shipped library has no measurement branches
#endif
    static_assert(G == 1 || (G == 2 && NF == 1 && NH == 0 && NQ == 0 && FILTER), "the 8-query form exists for the filtered M = 64 scan");
    constexpr int M = 64 * NF + 32 * NH + 16 * NQ;
    constexpr int BV = NQ ? 64 : 16;           // vectors per code block
    constexpr int BB = NQ ? 1024 : 16 * M;     // bytes per code block
    constexpr int NL = NQ ? 1 : NF;            // 16-byte code loads per lane and block
    // code blocks in flight per wave.  M = 128: two 2 KiB blocks — four would not fit 128 VGPRs.  M = 96 (round 5, the last A/B of the round,
    // profiles/r05zd_ab_rot_depth.txt): TWO 1.5 KiB blocks = 48 KiB in flight per CU scan 1.5 % faster than four (2.440 against 2.476-2.478 ms,
    // interleaved on one box; ONE block: 3.13 ms) — the IVF-Flat row streams said the same: past what covers the latency, bytes in flight cost
    constexpr int RD = (NF >= 2 || BB >= 1536) ? 2 : ROT_D;
    constexpr int NPH = G == 2 ? 2 : NF + NH + NQ;     // phases = table planes (G = 2: the two halves m < 32, m >= 32 of the 8-byte-entry table)
    constexpr int TAB = NPH * 65536;           // plane p at p * 64 KiB; row = code * 256; a half phase uses 128 B of the row
    constexpr int ISZ = 2 * G * 176;           // current / next item: G records each
    constexpr int NG = NQ ? 16 : M / 4;        // gathers per lane per block
    constexpr int NR1 = G == 2 ? 6 : NPH > 1 ? rot_nreg(1, NF >= 2 ? 16 : 8) : 0;      // G = 2: all 16 steps as three rotation bytes + the half byte
    constexpr int NR0 = G == 2 ? 0 : (NF >= 1 || NQ) ? 4 : rot_nreg(0, 8);     // M = 32: the half phase IS plane 0
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) uint32_t rot_s[];
    uint8_t* sb = reinterpret_cast<uint8_t*>(rot_s);
    PQRotItem* islot = reinterpret_cast<PQRotItem*>(sb + TAB);                  // [2][G] current / next item's records

    const PQScanArgs& a = A.b;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15, n = lane & 15, nq4 = n & 3;

    // ---- the workgroup's items: XCD b % 8 owns the contiguous item range [xlo, xhi) of the list-major order (the query groups
    // of a list tile stay on one XCD, close together in time); its workgroups draw items from one counter.  Only wave 0
    // talks to the counter, one item ahead, and publishes each item's record (or an end marker) through LDS.
    const int ti = *A.total_items;
    const int per_xcd = (ti + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    int xlo = xcd * per_xcd;
    int xhi = xlo + per_xcd; if (xhi > ti) xhi = ti;
    if (xlo >= xhi) return;
    uint32_t* ctr = xcd_ctr + xcd * 32;
    int cx = xcd, hops = 0;      // wave 0: the range being drawn from — its own XCD's, then (work stealing) the others' in turn
    unsigned drawn = 0;          // wave 0, lane 0: the counter value of the last draw
    // wave 0: turn the last draw into an item index; when the range is exhausted move on to the next XCD's range and draw
    // there (the ranges hold equal item COUNTS, not equal work: measured finish times of the 8 XCDs spread by 10 % of the
    // kernel; a stolen item loses its L2 neighbourhood, which only matters for the last few)
    auto resolve_draw = [&]() -> int {
        for (;;) {
            const int i = xlo + (int)__builtin_amdgcn_readfirstlane(drawn);
            if (i < xhi) return i;
            if (hops >= 7) return 0x7fffffff;
            hops++;
            cx = (cx + 1) & 7;
            xlo = cx * per_xcd;
            xhi = xlo + per_xcd; if (xhi > ti) xhi = ti;
            ctr = xcd_ctr + cx * 32;
            if (xlo >= xhi) { drawn = 0u; xlo = 0; xhi = 0; continue; }
            if (lane == 0) drawn = atomicAdd(ctr, 1u);
        }
    };
    // ---- per-lane constants, once per workgroup: rotation bytes, the one-hot B operand, the survivor-queue geometry
    uint32_t R0[NR0 > 0 ? NR0 : 1], R1[NR1 > 0 ? NR1 : 1];
#pragma unroll
    for (int r = 0; r < NR0; r++) {
        uint32_t v = 0;
#pragma unroll
        for (int bb = 0; bb < 4; bb++) {
            const int s = r * 4 + bb;
            const uint32_t rot = NQ ? (uint32_t)(64 * (g & 1) + 4 * ((i + s) & 15))           // M = 16: copy g & 1 of the 16 entries
                               : NF >= 1 ? (uint32_t)(64 * g + 4 * ((i + s) & 15))
                                         : (uint32_t)(64 * (g & 1) + 4 * ((i + s + 8 * (g >> 1)) & 15));
            v |= rot << (8 * bb);
        }
        R0[r] = v;
    }
#pragma unroll
    for (int r = 0; r < NR1; r++) {
        uint32_t v = G == 2 ? (uint32_t)(g >> 1) << 24 : 0x01000000u;     // byte 3 = plane 1 (G = 2: the lane's table half) -> address bit 16
#pragma unroll
        for (int bb = 0; bb < 3; bb++) {
            const int s = r * 3 + bb;
            const uint32_t rot = G == 2 ? (uint32_t)(128 * (g & 1) + 8 * ((i + s) & 15))                   // 8-byte slots: 16 (g & 1) + ((i + s) & 15)
                               : NF >= 2 ? (uint32_t)(64 * g + 4 * ((i + s) & 15))
                                         : (uint32_t)(64 * (g & 1) + 4 * ((i + s + 8 * (g >> 1)) & 15));
            v |= (rot & 255u) << (8 * bb);
        }
        R1[r] = v;
    }
    // B one-hot: K index 16 gK + 4 j + byte; column n picks byte n of every K group (n < 4) — or, M = 16, byte n & 3 of K group n >> 2 only.
    // G = 2: an A operand is two 8-byte gathers, dwords j = 0, 2 hold queries 0-3 and j = 1, 3 queries 4-7: column n < 8 picks byte n & 3 of
    // the dwords with j & 1 == n >> 2
    const int bsel = NQ ? (((n >> 2) == g) ? (1 << (8 * (n & 3))) : 0) : (n < 4 ? (1 << (8 * n)) : 0);
    const int bsel_hi = G == 2 ? ((n >= 4 && n < 8) ? (1 << (8 * (n & 3))) : 0) : bsel;
    const v4i Bm = {bsel, bsel_hi, bsel, bsel_hi};
    const int vo16 = lane * 16, vo8 = lane * 8;
    // Survivors leave the scan with PLAIN stores into WAVE-PRIVATE LOGS (round 4): every (workgroup, wave, query slot) owns one
    // append-only log of log_cap keys in HBM for the whole launch; the survivors of an item's query k go to the end of the wave's
    // log k — slot = the wave's running count plus the survivor's rank among this step's survivors of the same query (ballot +
    // mbcnt) — and one 8-byte descriptor {first key, count} per (item, wave, query) tells k_pq_gather_select / k_pq_rot_compact
    // where the item's run lies.  Rounds 2-3 gave every (item, wave, query) its own fixed segment of seg_cap keys: with 8.4 M
    // segments (the reference's nprobe 512) the pool allowed 128 keys each, and a query whose closest list is dense — 12 % of the
    // queries at M = 16, every query at k >= 1000 — overflowed and was re-run exactly.  A log only overflows when ONE wave
    // collects more than log_cap survivors of one slot in the whole launch (tens of thousands; counted, dropped, its queries
    // re-run exactly — never silently lost).  Nothing here returns a value: an LDS ds_add_rtn costs ~300 clk of the whole CU's
    // LDS pipe, and a returning global atomic sits in the wave's in-order vmcnt queue in front of the next item's table loads
    // (measured: 0.6 ms of a 3.8 ms scan for 1.9 M survivors).
    const uint64_t QM = NQ ? (0x1111111111111111ull << (n & 3))         // M = 16: every lane with n & 3 == query
                           : (n < 4 * G ? (0x0001000100010001ull << n) : 0ull);   // the four lanes that own query n (G = 2: n = 4 record + slot)
    const int rq = G == 2 ? (n >> 2) & 1 : 0;                                // the record of the item this lane's query column belongs to

    const size_t mylog_i = ((size_t)blockIdx.x * 16 + (size_t)w) * (4 * G) + (size_t)(4 * rq + nq4);     // this lane's log (record rq, query slot nq4 of this wave)
    uint64_t* const mylog = log_keys + mylog_i * (size_t)log_cap;
    uint32_t lcur = 0;                    // keys appended to the lane's log so far (equal in all lanes of a query slot); persists across items
    int item = 0;
    if (w == 0) {
        if (lane == 0) { drawn = atomicAdd(ctr, 1u); }
        item = resolve_draw();
        uint4 r0 = make_uint4(0xffffffffu, 0, 0, 0);       // l = -1: end marker
        if (lane < 11 * G && item != 0x7fffffff) r0 = reinterpret_cast<const uint4*>(&items[(size_t)item * G])[lane];
        if (lane < 11 * G) reinterpret_cast<uint4*>(&islot[0])[lane] = r0;
        if (lane == 0) islot[0].pad0 = item;
    }
    int buf = 0;
#pragma unroll 1
    for (;; buf ^= 1) {
        __syncthreads();    // #1: every wave has left the previous item's scan (table free), the item record is in LDS
        const PQRotItem* it = &islot[buf * G];
        const int item_l = __builtin_amdgcn_readfirstlane(it->l);
        if (item_l == -1) break;
        item = __builtin_amdgcn_readfirstlane(it->pad0);
        const bool skip_item = item_l < -1;         // pruned: no survivors possible, only the bookkeeping below runs
        const int np_raw = __builtin_amdgcn_readfirstlane(it->np);
        const int np = np_raw & 15;
        const int64_t len = ((int64_t)__builtin_amdgcn_readfirstlane((int)(it->len >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)it->len);
        const int64_t base_row = ((int64_t)__builtin_amdgcn_readfirstlane((int)(it->base_row >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)it->base_row);
        const int nblk = NQ ? (int)((len + 63) >> 6) : (int)(((len + 63) >> 6) << 2);     // code blocks of the list, slab padding included
        const int tb0 = __builtin_amdgcn_readfirstlane(it->tile) * (16 * bpw);
        const uint8_t* lp = a.codes + (base_row / BV) * (int64_t)BB;
        // ---- code loads: buffer instructions on a descriptor of THIS list's blocks (base + size in SGPRs, the block's byte
        // offset in an SGPR, lane * 16 in one constant VGPR): no address VALU, and a block past the list's end reads zeros
        // instead of needing a clamp (its sums are garbage that the pos < len test of the survivor path drops)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)lp, 0, nblk * BB, 0x00020000);
        v4u ca[RD][NL > 0 ? NL : 1]; v2u cb[RD];
        // ---- circular scan (round 3): the sibling groups of a list tile (the item's family: adjacent items, different CUs of this
        // XCD) read the same code lines, and a line lives ~5 us in the XCD's L2.  A workgroup that starts while a sibling is
        // already under way therefore does not begin at the tile's first block: it JOINS the most advanced running sibling at
        // its current position, scans to the end of the tile alongside it (one of the two fetches a line from HBM, the other
        // finds it in the L2), then wraps around and scans the part it skipped.  No workgroup ever waits for another; block
        // order never affects results (survivors are keys).  The start position needs the siblings' progress words, which
        // wave 0 requests here (LDS-DMA, lane i -> slot i) and reads after its share of the table staging.
        const int join_on = (pace_arg >> 7) & 1;
        int f0 = item - ((np_raw >> 4) & 0x3fff), f1 = f0 + ((np_raw >> 18) & 0x3fff);
        {   // a family stays inside the item's own XCD range (its other part runs on another XCD: no shared L2)
            const int rlo = (item / per_xcd) * per_xcd;
            int rhi = rlo + per_xcd; if (rhi > ti) rhi = ti;
            if (f0 < rlo) f0 = rlo;
            if (f1 > rhi) f1 = rhi;
        }
        if (f1 - f0 > 8) {                    // the 8-item window around this item (one 32-byte LDS read)
            int a0 = item - 3;
            if (a0 < f0) a0 = f0;
            if (a0 > f1 - 8) a0 = f1 - 8;
            f0 = a0; f1 = a0 + 8;
        }
        constexpr uint32_t chunk_a = (uint32_t)(TAB + ISZ);                        // LDS word: next unassigned chunk sequence number
        constexpr uint32_t i0_a = (uint32_t)(TAB + ISZ + 4);                       // LDS word: the item's first loop iteration (join)
        constexpr uint32_t sib_a = (uint32_t)(TAB + ((ISZ + 8 + 31) & ~31));       // LDS [8]: the siblings' progress as last seen (G = 1: TAB + 384)
        const bool family = join_on && !skip_item && f1 - f0 > 1 && item >= f0 && item < f1;
        if (w == 0) {
            if (lane < 8) lds_wr32(sib_a + 4u * (uint32_t)lane, 0x7fffffffu);       // own slot and the lanes beyond the family never count
            if (family && f0 + lane < f1 && f0 + lane != item) sib_refresh(&prog[f0 + lane], sib_a);
        }
        // ---- stage the group's table: work unit = (code, 4 consecutive m) -> 4 dwords (one per m: byte k = query k, as int8 = u8 - 128).
        // All the loads of a thread's 256 * (M / 4) / 1024 units are issued before the first one is used (round 3: the loop
        // used to pay one L2 round trip per unit, 6 in a row for M = 96).
        if constexpr (G == 2) {
            // eight queries: unit = (code, 4 consecutive m) -> the eight queries' dwords -> four 8-byte entries (bytes 0-3: record 0's
            // queries, 4-7: record 1's), 32 contiguous bytes of the code's row in half m >> 5
            const PQRotItem* itb = &islot[buf * G + 1];
            const int npb = __builtin_amdgcn_readfirstlane(itb->np) & 15;
            const int64_t qq[8] = {it->q[0], it->q[1], it->q[2], it->q[3], itb->q[0], itb->q[1], itb->q[2], itb->q[3]};
            constexpr int NU = 256 * (M / 4) / 1024;              // 4
            uint32_t in[NU][8];
            const bool live = !((var & 4) || skip_item);
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int e = tid + u * 1024;
                const int c = e / (M / 4), m4 = e - c * (M / 4);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    in[u][k] = (live && (k < 4 ? k < np : k - 4 < npb)) ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(A.lut8 + (qq[k] * 256 + c) * M + m4 * 4)) : 0u;
            }
            if (live) {
#pragma unroll
                for (int u = 0; u < NU; u++) {
                    const int e = tid + u * 1024;
                    const int c = e / (M / 4), m4 = e - c * (M / 4);
                    uint32_t o[2][4];
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {
                        const uint32_t a0 = in[u][4 * hh], a1 = in[u][4 * hh + 1], a2 = in[u][4 * hh + 2], a3 = in[u][4 * hh + 3];
                        const uint32_t t0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), t1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);
                        const uint32_t u0 = __builtin_amdgcn_perm(a3, a2, 0x05010400u), u1 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
                        o[hh][0] = __builtin_amdgcn_perm(u0, t0, 0x05040100u) ^ 0x80808080u;
                        o[hh][1] = __builtin_amdgcn_perm(u0, t0, 0x07060302u) ^ 0x80808080u;
                        o[hh][2] = __builtin_amdgcn_perm(u1, t1, 0x05040100u) ^ 0x80808080u;
                        o[hh][3] = __builtin_amdgcn_perm(u1, t1, 0x07060302u) ^ 0x80808080u;
                    }
                    uint8_t* dst = sb + (m4 >> 3) * 65536 + c * 256 + (m4 & 7) * 32;
                    *reinterpret_cast<uint4*>(dst) = make_uint4(o[0][0], o[1][0], o[0][1], o[1][1]);
                    *reinterpret_cast<uint4*>(dst + 16) = make_uint4(o[0][2], o[1][2], o[0][3], o[1][3]);
                }
            }
        } else {
            const int64_t q0 = it->q[0], q1 = it->q[1], q2 = it->q[2], q3 = it->q[3];
            constexpr int NU = (256 * (M / 4) + 1023) / 1024;
            const int nunits = ((var & 4) || skip_item) ? 0 : 256 * (M / 4);
            constexpr int GB = NU <= 6 ? NU : (NU + 1) / 2;          // loads in flight per thread: at most 6 x 4 registers
            uint32_t in[GB][4];
#pragma unroll
            for (int g0 = 0; g0 < NU; g0 += GB) {
#pragma unroll
            for (int u = 0; u < GB; u++) {
                const int e = tid + (g0 + u) * 1024;
                const int ee = e < nunits ? e : 0;
                const int c = ee / (M / 4), m4 = ee - c * (M / 4);
                // non-temporal: a table row is read once per item (out of the Infinity Cache) — it should not displace the code lines the
                // sibling groups are about to re-read from this L2 (round 5, interleaved A/Bs: scan -1 to -2 % at every shard size from 12.5M to 100M vectors)
                in[u][0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(A.lut8 + (q0 * 256 + c) * M + m4 * 4));
                in[u][1] = np > 1 ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(A.lut8 + (q1 * 256 + c) * M + m4 * 4)) : 0u;
                in[u][2] = np > 2 ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(A.lut8 + (q2 * 256 + c) * M + m4 * 4)) : 0u;
                in[u][3] = np > 3 ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(A.lut8 + (q3 * 256 + c) * M + m4 * 4)) : 0u;
            }
#pragma unroll
            for (int u = 0; u < GB; u++) {
                const int e = tid + (g0 + u) * 1024;
                if (e >= nunits) continue;
                const int c = e / (M / 4), m4 = e - c * (M / 4);
                const uint32_t t0 = __builtin_amdgcn_perm(in[u][1], in[u][0], 0x05010400u), t1 = __builtin_amdgcn_perm(in[u][1], in[u][0], 0x07030602u);
                const uint32_t u0 = __builtin_amdgcn_perm(in[u][3], in[u][2], 0x05010400u), u1 = __builtin_amdgcn_perm(in[u][3], in[u][2], 0x07030602u);
                uint4 o;
                o.x = __builtin_amdgcn_perm(u0, t0, 0x05040100u) ^ 0x80808080u;
                o.y = __builtin_amdgcn_perm(u0, t0, 0x07060302u) ^ 0x80808080u;
                o.z = __builtin_amdgcn_perm(u1, t1, 0x05040100u) ^ 0x80808080u;
                o.w = __builtin_amdgcn_perm(u1, t1, 0x07060302u) ^ 0x80808080u;
                const int m = m4 * 4;
                const int p = m < 64 * NF ? (m >> 6) : NF;
                const int slot = m < 64 * NF ? (m & 63) : (m - 64 * NF);
                *reinterpret_cast<uint4*>(sb + p * 65536 + c * 256 + slot * 4) = o;
                if (NQ) *reinterpret_cast<uint4*>(sb + c * 256 + 64 + slot * 4) = o;      // M = 16: second copy for the odd lane groups
            }
            }
        }
        // ---- next item: drawn JUST IN TIME (round 3).  The XCD's item order is (list, tile, group), so the workgroups that
        // draw the sibling groups of a list tile are the ones that come free one after the other: they start within a few
        // microseconds of each other and stream the same code lines, the followers out of the XCD's L2 (a line lives ~5 us
        // there at this fill rate).  Round 2 drew one item AHEAD, which bound an item to a workgroup ~70 us before it started
        // and scattered the siblings in time: L2 hit rate 21 %, 19.4 GB from the fabric for 9.6 GB of unique code bytes.
        // Wave 0 issues the draw three loop iterations before the end of its share, the record load one iteration before the
        // end: both latencies hide behind the last blocks of the scan.
        uint4 pre = make_uint4(0xffffffffu, 0, 0, 0);
        int i1 = 0x7fffffff;
        int dstate = 0;                       // wave 0: 0 = not drawn, 1 = draw in flight, 2 = record requested
        int nit0 = 0;                         // loop iterations of wave 0 for this item
        {
            int nb0 = skip_item ? 0 : (nblk - tb0 + 15) >> 4;
            if (nb0 > bpw) nb0 = bpw;
            if (nb0 < 0) nb0 = 0;
            nit0 = (nb0 + RD - 1) / RD;
        }
        // static columns up to sequence number dyn_from, the tail from the LDS counter (which starts there): the waves of a SIMD do
        // not run at one speed — the arbiter favours the oldest — and with a fully static split the slowest wave finished its share
        // 9 us after the fastest (49 vs 58 us per item, measured); drawing EVERY chunk dynamically cost more in LDS atomics than the
        // balance returned, so only the last dyn_rows rows of a tile are drawn
        // (bit 4 of pq_pace: every chunk dynamic — measured: all waves then finish together at 60 us, later than the slowest did)
        const bool dyn_on = ((pace_arg >> 4) & 1) != 0;
        const int dyn_rows = dyn_on ? 0x7fff : (int)((pace_arg >> 12) & 15);      // bits 12-15 of pq_pace (0 = static throughout)
        int dyn_from = 16 * (nit0 - dyn_rows);                                      // nit0 rows in all, the last one possibly partial
        if (dyn_from < 32) dyn_from = 32;
        if (dyn_rows == 0) dyn_from = 0x7fffffff;
        const bool prio_rot = !((pace_arg >> 4) & 4);      // diagnostics: bit 6 of pq_pace switches the priority rotation off
#ifdef RSX_MEASURE
        if (tid == 0 && item < 65536) {
            g_rot_trace[4 * item + 0] = wall_clock64();
            g_rot_trace[4 * item + 2] = ((uint64_t)blockIdx.x << 32) | (uint32_t)nit0;
            g_rot_trace[4 * item + 3] = ((uint64_t)(uint32_t)f0 << 32) | (uint32_t)f1;
        }
#endif
        if (w == 0) {
            uint32_t i0v = 0;
            if (family) {
                sib_landed();                 // the progress words requested before the staging (the wave's table loads are older: no extra wait)
                if (join_on) {
                    // most advanced sibling still on its first pass (bit 30 marks a wrapped one, 0 = not started, 0x7fffffff = done);
                    // its word is a few microseconds old by the time this workgroup scans: start `ahead` rows further on
                    const uint32_t best = sib_max8_first_pass(sib_a);
                    int ahead = (pace_arg >> 8) & 15; if (ahead == 0) ahead = 2;
                    if (best != 0u && (int)best - 1 + ahead + 2 < nit0 - 1) i0v = best - 1u + (uint32_t)ahead;
                }
            }
            if (lane == 0) { lds_wr32(i0_a, i0v); lds_wr32(chunk_a, (uint32_t)dyn_from); }
        }
        // ---- the lane's share of the item record: accumulator init, score parameters of the query it owns (n < 4)
        const PQRotItem* itq = &islot[buf * G + rq];        // the record of this lane's query column
        const int cinit = FILTER ? ((NQ || n < 4 * G) ? itq->cinit[nq4] : -(1 << 30)) : 0;
        const v4i Ci = {cinit, cinit, cinit, cinit};
        // (the score parameters of the lane's query are read from the item record in LDS where a survivor is scored — rare — instead
        //  of living in ~9 VGPRs through the gather loop: the M = 16 form runs two workgroups per CU on 64 VGPRs)
        const float p_dis0_u = FILTER ? 0.0f : it->dis0[nq4], p_scale_u = FILTER ? 0.0f : it->scale[nq4], p_bias_u = FILTER ? 0.0f : it->bias[nq4];
        const int64_t p_off_u = FILTER ? 0 : it->off[nq4];
        const uint32_t qstart = lcur;         // the lane's log position at the start of this item
        __syncthreads();    // #2: table staged
#ifdef RSX_MEASURE
        const uint64_t t_scan0 = wall_clock64();
#endif

        // ---- scan.  The tile is a grid of CHUNKS: chunk (row r, column c) = the RD blocks tb0 + c + 16 (r RD + dd), i.e. the
        // blocks column c of the 16 would scan in loop iteration r of a static round-robin.  Rows 0 .. R-1 are full, the last row
        // has cl <= 16 non-empty chunks.  Chunks are taken in SEQUENCE order n = 0, 1, ...: the full rows circularly from row r0
        // (the join position), the partial row last; column = n % 16.  Wave w takes the sequence numbers w, 16 + w, 32 + w, ...
        // (static split), or — dyn_on — its first two statically and every further one from an LDS counter, one loop iteration
        // ahead of its use.  Each wave keeps RD blocks in flight: while chunk A is scanned its register slots are refilled
        // with chunk B's blocks.
        int nb0 = ((var & 2) || skip_item) ? 0 : (nblk - tb0 + 15) >> 4;           // blocks of column 0 (the longest column)
        if (nb0 > bpw) nb0 = bpw;
        if (nb0 < 0) nb0 = 0;
        const int nrow = (nb0 + RD - 1) / RD;                                // = nit0
        const int R = nrow > 0 ? nrow - 1 : 0;                                     // full rows
        int cl = nrow > 0 ? nblk - tb0 - 16 * (R * RD) : 0;                     // non-empty chunks of the last row
        if (cl > 16) cl = 16;
        if (cl < 0) cl = 0;
        const int nch = 16 * R + cl;
        int r0 = join_on ? (int)__builtin_amdgcn_readfirstlane((int)lds_rd32_volatile(i0_a)) : 0;
        if (r0 >= R) r0 = 0;
        const int so_oob = nblk * BB;                                        // past the descriptor's end: reads zeros
        constexpr int BK_OOB = 0x3fffff00;                                   // "no such chunk" as a block index (any comparison against the tile's end fails)
        const int bk_end = tb0 + 16 * bpw;                                   // first block past this tile
        // Round 4: chunks are tracked as BLOCK indices (their byte offsets used to be divided by the block size again for every
        // block — ~16 scalar instructions per block, and the counters say the loop issues ONE instruction per 4 clk and SIMD whatever
        // its kind: SQ_INSTS_{VALU,LDS,SALU,VMEM} = 1.23e9 per 4.77e9 SIMD cycles, profiles/r03z_pmc_sq_counters.md)
        // sequence number -> first block of the chunk (BK_OOB: no such chunk), its row, and whether it is on the second pass
        auto row_of = [&](int nseq, bool& wrapped) -> int {          // row of a valid sequence number, and whether it is on the second pass
            int r = nseq >> 4;
            if (r < R) { r += r0; wrapped = r >= R; if (wrapped) r -= R; } else { r = R; wrapped = r0 > 0; }
            return r;
        };
        auto chunk_of = [&](int nseq) -> int {
            if (nseq >= nch) return BK_OOB;
            bool wr_;
            return tb0 + (nseq & 15) + 16 * (row_of(nseq, wr_) * RD);
        };
        int nA = w, nB = 16 + w, nC = 0x7fffffff;
        int bkA = chunk_of(nA), bkB = chunk_of(nB);
#pragma unroll
        for (int dd = 0; dd < RD; dd++) {     // chunk A's blocks
            // (a chunk's RD blocks lie 16 apart in ONE column of the tile: with fewer than RD blocks per wave and tile — M = 16 at small
            //  tiles — the later ones would belong to the next tile, which another item scans)
            const int so = (bkA + 16 * dd >= bk_end) ? so_oob : (bkA + 16 * dd) * BB;
#pragma unroll
            for (int p = 0; p < NL; p++) ca[dd][p] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo16 + p * 1024, so, 0);
            if (NH) cb[dd] = __builtin_amdgcn_raw_buffer_load_b64(rs, vo8 + NF * 1024, so, 0);
            // keep the issue order dd = 0 .. RD-1: the loop waits for slot dd with `vmcnt(loads of the RD-1 younger slots)`, and one
            // s_waitcnt serves both the loop entry and the back edge — with the scheduler's order (slot 0 LAST) the entry needs
            // vmcnt(0), so every iteration drained all RD refills it had just issued (prefetch distance one block instead of RD)
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll 1
        for (int k = 0; nA < nch; k++) {
            // the chunk after next: drawn now, needed at the end of this iteration
            if (nB + 16 < dyn_from) nC = nB + 16;
            else if (lane == 0) nC = (int)__hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(sb + chunk_a), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int b0 = bkA;                                              // first block of chunk A
            const int bk_next = bkB;
            if (prio_rot) {
                // the four waves of a SIMD (w, w + 4, w + 8, w + 12) take turns at the top issue priority, one loop iteration each
                switch ((k + (w >> 2)) & 3) {
                    case 0: __builtin_amdgcn_s_setprio(3); break;
                    case 1: __builtin_amdgcn_s_setprio(2); break;
                    case 2: __builtin_amdgcn_s_setprio(1); break;
                    default: __builtin_amdgcn_s_setprio(0); break;
                }
            }
            if (w == 0) {
                // progress word for joining siblings: the row this workgroup is on.  The siblings share this XCD's L2, so L2-level
                // visibility is all that is needed — a plain store (the vector L1 is write-through) and readers that bypass their
                // L1 (sc0).  A device-scope store (sc1) writes through to the fabric and its late acknowledgement holds the wave's
                // in-order vmcnt queue, i.e. every code load behind it (measured: the scan 1.45x slower with one per iteration).
                if (family) {
                    bool wrapA; const int rowA = row_of(nA, wrapA);
                    if (lane == 0)
                        __hip_atomic_store(&prog[item], (uint32_t)(rowA + 1) | (wrapA ? 0x40000000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (dstate == 1 && nA >= nch - 16) {
                    i1 = resolve_draw();
                    if (lane < 11 * G && i1 != 0x7fffffff) pre = reinterpret_cast<const uint4*>(&items[(size_t)i1 * G])[lane];
                    dstate = 2;
                }
                if (dstate == 0 && nA >= nch - 48) { if (lane == 0) drawn = atomicAdd(ctr, 1u); dstate = 1; }
            }
#pragma unroll
            for (int dd = 0; dd < RD; dd++) {
                const int b = b0 + 16 * dd;
                uint32_t gv[NG];
                // addresses: (plane << 16) | (code << 8) | rotation byte — one v_perm each
                if constexpr (G == 2) {
                    const uint32_t cw[4] = {ca[dd][0].x, ca[dd][0].y, ca[dd][0].z, ca[dd][0].w};
#pragma unroll
                    for (int s = 0; s < 16; s++)
                        gv[s] = __builtin_amdgcn_perm(cw[s >> 2], R1[s / 3], 0x0c030000u | ((uint32_t)(4 + (s & 3)) << 8) | (uint32_t)(s % 3));
                } else if (NF >= 1 || NQ) {
                    const uint32_t cw[4] = {ca[dd][0].x, ca[dd][0].y, ca[dd][0].z, ca[dd][0].w};
#pragma unroll
                    for (int s = 0; s < 16; s++)
                        gv[s] = __builtin_amdgcn_perm(cw[s >> 2], R0[s >> 2], 0x0c0c0000u | ((uint32_t)(4 + (s & 3)) << 8) | (uint32_t)(s & 3));
                }
                if (NF >= 2) {
                    const uint32_t cw[4] = {ca[dd][NF - 1].x, ca[dd][NF - 1].y, ca[dd][NF - 1].z, ca[dd][NF - 1].w};
#pragma unroll
                    for (int s = 0; s < 16; s++)
                        gv[16 + s] = __builtin_amdgcn_perm(cw[s >> 2], R1[s / 3], 0x0c030000u | ((uint32_t)(4 + (s & 3)) << 8) | (uint32_t)(s % 3));
                }
                if (NH) {
                    const uint32_t cw[2] = {cb[dd].x, cb[dd].y};
#pragma unroll
                    for (int s = 0; s < 8; s++) {
                        if (NF == 0)
                            gv[s] = __builtin_amdgcn_perm(cw[s >> 2], R0[s >> 2], 0x0c0c0000u | ((uint32_t)(4 + (s & 3)) << 8) | (uint32_t)(s & 3));
                        else
                            gv[16 * NF + s] = __builtin_amdgcn_perm(cw[s >> 2], R1[s / 3], 0x0c030000u | ((uint32_t)(4 + (s & 3)) << 8) | (uint32_t)(s % 3));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                {   // the slot's code registers are dead: refill them in place
                    const int so = (bk_next + 16 * dd >= bk_end) ? so_oob : (bk_next + 16 * dd) * BB;
#pragma unroll
                    for (int p = 0; p < NL; p++) ca[dd][p] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo16 + p * 1024, so, 0);
                    if (NH) cb[dd] = __builtin_amdgcn_raw_buffer_load_b64(rs, vo8 + NF * 1024, so, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                v4i C = Ci;
                if constexpr (G == 2) {
                    rot_v2u g8[NG];
#pragma unroll
                    for (int s = 0; s < NG; s++) g8[s] = lds_rd64(gv[s]);
#pragma unroll
                    for (int t = 0; t < NG / 2; t++) {
                        const v4i Av = {(int)g8[2 * t].x, (int)g8[2 * t].y, (int)g8[2 * t + 1].x, (int)g8[2 * t + 1].y};
                        C = __builtin_amdgcn_mfma_i32_16x16x64_i8(Av, Bm, C, 0, 0, 0);
                    }
                } else {
#ifdef RSX_MEASURE
                // cost split (tools/ builds only): 16 = no table look-ups (the addresses stand in for the data), 32 = no MFMAs (one VALU
                // add per quad instead), 64 = no address formation (the code words stand in for the addresses, masked into the table)
                if (var & 64) {
#pragma unroll
                    for (int s = 0; s < NG; s++) gv[s] = (NF >= 1 || NQ ? ca[dd][0].x : cb[dd].x) & 0xfffcu;
                }
                if (!(var & 16)) {
#pragma unroll
                    for (int s = 0; s < NG; s++) gv[s] = lds_rd32(gv[s]);
                }
                if (var & 32) {
#pragma unroll
                    for (int t = 0; t < NG / 4; t++) C[t & 3] += (int)(gv[4 * t] ^ gv[4 * t + 1] ^ gv[4 * t + 2] ^ gv[4 * t + 3]) >> 31;
                } else {
#pragma unroll
                    for (int t = 0; t < NG / 4; t++) {
                        const v4i Av = {(int)gv[4 * t], (int)gv[4 * t + 1], (int)gv[4 * t + 2], (int)gv[4 * t + 3]};
                        C = __builtin_amdgcn_mfma_i32_16x16x64_i8(Av, Bm, C, 0, 0, 0);
                    }
                }
#else
#pragma unroll
                for (int s = 0; s < NG; s++) gv[s] = lds_rd32(gv[s]);
#pragma unroll
                for (int t = 0; t < NG / 4; t++) {
                    const v4i Av = {(int)gv[4 * t], (int)gv[4 * t + 1], (int)gv[4 * t + 2], (int)gv[4 * t + 3]};
                    C = __builtin_amdgcn_mfma_i32_16x16x64_i8(Av, Bm, C, 0, 0, 0);
                }
#endif
                }
                // C[r] (lanes n < 4) = cinit + sum over m of (u8 - 128) for vector 4 g + r of the block and query n
                if (FILTER) {
                    if (__builtin_amdgcn_ballot_w64((C[0] & C[1] & C[2] & C[3]) >= 0) && !(var & 1)) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const bool cnd = C[r] >= 0;
                            if (__builtin_amdgcn_ballot_w64(cnd)) {
                                const float p_dis0 = itq->dis0[nq4], p_scale = itq->scale[nq4], p_bias = itq->bias[nq4];
                                const int64_t p_off = itq->off[nq4];
                                const uint64_t p_tau = itq->tau[nq4];
                                const int64_t pos = NQ ? ((int64_t)b << 6) + 16 * (n >> 2) + 4 * g + r : ((int64_t)b << 4) + 4 * g + r;
                                const float sc = p_dis0 + __fmaf_rn(p_scale, (float)(C[r] - cinit + 128 * M), p_bias);
                                const uint64_t key = (cnd && pos < len && b - tb0 < 16 * bpw) ? make_key(sc, (uint32_t)p_off + (uint32_t)pos) : 0ull;
                                const bool pass = key > p_tau;
                                const uint64_t mq = __builtin_amdgcn_ballot_w64(pass) & QM;      // this step's survivors of MY query
                                if (pass) {
                                    const uint32_t slot = lcur + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
                                    if (slot < (uint32_t)log_cap) mylog[slot] = key;   // beyond: counted, dropped -> the query is re-run exactly
                                }
                                lcur += (uint32_t)__builtin_popcountll(mq);
                            }
                        }
                    }
                } else {
                    if ((NQ ? (n & 3) < np : n < np) && b < nblk && b - tb0 < 16 * bpw) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int64_t pos = NQ ? ((int64_t)b << 6) + 16 * (n >> 2) + 4 * g + r : ((int64_t)b << 4) + 4 * g + r;
                            const float sc = p_dis0_u + __fmaf_rn(p_scale_u, (float)(C[r] + 128 * M), p_bias_u);
                            a.temp[p_off_u + pos] = (pos < len) ? sc : -__builtin_inff();
                        }
                    }
                }
            }
            nA = nB; bkA = bkB;
            nB = __builtin_amdgcn_readfirstlane(nC);
            bkB = chunk_of(nB);
        }
        // ---- item epilogue: the wave's four run descriptors leave with one store (lanes 0..3 own queries 0..3): first key of the
        // run in the log pool, keys stored, bit 31 = the log was full and keys were dropped; wave 0 parks the next record (or the
        // end marker) and draws the index of the item after it
        __builtin_amdgcn_s_setprio(0);
        if (FILTER && lane < 4 * G) {      // lane = 4 record + slot = its own query column n (g = 0)
            const uint32_t c0 = qstart < (uint32_t)log_cap ? qstart : (uint32_t)log_cap, c1 = lcur < (uint32_t)log_cap ? lcur : (uint32_t)log_cap;
            seg_desc[(((size_t)item * G + (size_t)(lane >> 2)) * 16 + w) * 4 + (lane & 3)] = make_uint2((uint32_t)(mylog_i * (size_t)log_cap) + c0, (c1 - c0) | ((lcur > (uint32_t)log_cap && lcur > qstart) ? 0x80000000u : 0u));   // only a run that itself lost keys is flagged
        }
#ifdef RSX_MEASURE
        if (lane == 0 && item < 16384) g_rot_wave[16 * item + w] = (uint32_t)(wall_clock64() - t_scan0);
        if (tid == 0 && item < 65536) g_rot_trace[4 * item + 1] = wall_clock64();
#endif
        if (w == 0 && lane == 0 && join_on)       // done: no sibling joins this item any more
            __hip_atomic_store(&prog[item], 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (w == 0) {
            if (dstate == 0) { if (lane == 0) drawn = atomicAdd(ctr, 1u); dstate = 1; }     // empty / pruned / one-iteration items
            if (dstate == 1) {
                i1 = resolve_draw();
                if (lane < 11 * G && i1 != 0x7fffffff) pre = reinterpret_cast<const uint4*>(&items[(size_t)i1 * G])[lane];
            }
            if (lane < 11 * G) reinterpret_cast<uint4*>(&islot[(buf ^ 1) * G])[lane] = pre;
            if (lane == 0) islot[(buf ^ 1) * G].pad0 = i1;
        }
    }
}

// ---------------------------------------------------------------------------------------
// M = 16, filtered scan: SIXTEEN queries per pass over a list tile (round 4) — k_pq_scan_rot16.
// The reference's shipped IVF-PQ point (ric/conf/ivf_pq.yaml:64-78: M 16, 8192 lists, 512 probes) has ~64 probing queries per
// list and batch: sixteen 4-query groups, each of which re-read the list's codes in k_pq_scan_rot<0,0,true,1> — 27.7 GB through
// the L2s per batch for 1.6 GB of codes, 6.7 TB/s at 4.1 ms: bound by that traffic, not by the CUs (two workgroups per CU changed
// nothing: profiles/r04_scan_experiments.md).  Here a work item carries FOUR records = up to 16 queries (k_pq_rot_items, ngq = 4)
// and every 64-vector code block is loaded once and looked up in four tables: a table row is 256 bytes and a 4-query group needs
// 128 of them (its 16 entries twice: lane groups of even / odd g use one copy each), so a 64 KiB plane holds TWO groups — bytes
// 0-127 map to banks 0-31, bytes 128-255 to banks 32-63, every gather stays conflict-free — and two planes (128 KiB) hold four.
// Survivors, logs and run descriptors are per RECORD, exactly as k_pq_scan_rot leaves them: the compaction / gather kernels and
// everything behind them are unchanged.  Simpler than the general kernel on purpose: static block columns, no circular join (a
// list has 4 sibling items instead of 16), items drawn one ahead.
// ---------------------------------------------------------------------------------------
constexpr int R16_G = 4;
__global__ __launch_bounds__(1024) void k_pq_scan_rot16(PQScan8Args A, const PQRotItem* __restrict__ items, uint64_t* __restrict__ log_keys,
                                                        uint2* __restrict__ seg_desc, uint32_t* xcd_ctr, int log_cap, int bpw) {
    constexpr int M = 16, BB = 1024, RD = 4, G = R16_G;
    constexpr int TAB = 2 * 65536;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) uint32_t rot16_s[];
    uint8_t* sb = reinterpret_cast<uint8_t*>(rot16_s);
    PQRotItem* islot = reinterpret_cast<PQRotItem*>(sb + TAB);                  // [2][G] current / next item's records
    const PQScanArgs& a = A.b;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15, n = lane & 15, nq4 = n & 3;
    // ---- items: XCD b % 8 owns a contiguous range of the list-major item order, its workgroups draw from one counter, an exhausted
    // range steals from the next XCD's (as k_pq_scan_rot)
    const int ti = *A.total_items;
    const int per_xcd = (ti + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    int xlo = xcd * per_xcd;
    int xhi = xlo + per_xcd; if (xhi > ti) xhi = ti;
    if (xlo >= xhi) return;
    uint32_t* ctr = xcd_ctr + xcd * 32;
    int cx = xcd, hops = 0;
    unsigned drawn = 0;
    auto resolve_draw = [&]() -> int {
        for (;;) {
            const int i2 = xlo + (int)__builtin_amdgcn_readfirstlane(drawn);
            if (i2 < xhi) return i2;
            if (hops >= 7) return 0x7fffffff;
            hops++;
            cx = (cx + 1) & 7;
            xlo = cx * per_xcd;
            xhi = xlo + per_xcd; if (xhi > ti) xhi = ti;
            ctr = xcd_ctr + cx * 32;
            if (xlo >= xhi) { drawn = 0u; xlo = 0; xhi = 0; continue; }
            if (lane == 0) drawn = atomicAdd(ctr, 1u);
        }
    };
    // ---- per-lane constants: rotation bytes (lane (g, i) = vector 16 g + i reaches sub-quantiser (i + s) & 15 at step s, copy g & 1
    // of the code's 16 entries), for the two row halves (+128) and the two planes (plane 1: three rotation bytes + the plane byte)
    uint32_t R00[4], R01[4], R10[6], R11[6];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        uint32_t v = 0;
#pragma unroll
        for (int bb = 0; bb < 4; bb++) v |= (uint32_t)(64 * (g & 1) + 4 * ((i + r * 4 + bb) & 15)) << (8 * bb);
        R00[r] = v; R01[r] = v | 0x80808080u;
    }
#pragma unroll
    for (int r = 0; r < 6; r++) {
        uint32_t v = 0x01000000u;     // byte 3 = plane 1 -> address bit 16
#pragma unroll
        for (int bb = 0; bb < 3; bb++) { const int s2 = r * 3 + bb; if (s2 < 16) v |= (uint32_t)(64 * (g & 1) + 4 * ((i + s2) & 15)) << (8 * bb); }
        R10[r] = v; R11[r] = v | 0x00808080u;
    }
    // B one-hot: column n = (vector group n >> 2, query n & 3) takes byte n & 3 of K group n >> 2
    const int bsel = ((n >> 2) == g) ? (1 << (8 * (n & 3))) : 0;
    const v4i Bm = {bsel, bsel, bsel, bsel};
    const int vo16 = lane * 16;
    const uint64_t QM = 0x1111111111111111ull << (n & 3);                // every lane with n & 3 == my query slot
    const size_t mylog_0 = ((size_t)blockIdx.x * 16 + (size_t)w) * (4 * G) + (size_t)nq4;      // + 4 gq: the log of (record gq, slot nq4) of this wave
    uint32_t lcur[G];
#pragma unroll
    for (int gq = 0; gq < G; gq++) lcur[gq] = 0u;
    const auto load_records = [&](int it_) -> uint4 {       // lanes 0 .. 11 G - 1: the item's G records, 16 bytes per lane
        uint4 r0 = make_uint4(0xffffffffu, 0, 0, 0);        // l = -1: end marker
        if (lane < 11 * G && it_ != 0x7fffffff) r0 = reinterpret_cast<const uint4*>(&items[(size_t)it_ * G])[lane];
        return r0;
    };
    int item = 0;
    if (w == 0) {
        if (lane == 0) drawn = atomicAdd(ctr, 1u);
        item = resolve_draw();
        const uint4 r0 = load_records(item);
        if (lane < 11 * G) reinterpret_cast<uint4*>(&islot[0])[lane] = r0;
        if (lane == 0) islot[0].pad0 = item;
    }
    int buf = 0;
#pragma unroll 1
    for (;; buf ^= 1) {
        __syncthreads();    // #1: every wave has left the previous item's scan (tables free), the records are in LDS
        const PQRotItem* it0 = &islot[buf * G];
        const int item_l = __builtin_amdgcn_readfirstlane(it0->l);
        if (item_l == -1) break;
        item = __builtin_amdgcn_readfirstlane(it0->pad0);
        const int64_t len = ((int64_t)__builtin_amdgcn_readfirstlane((int)(it0->len >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)it0->len);
        const int64_t base_row = ((int64_t)__builtin_amdgcn_readfirstlane((int)(it0->base_row >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)it0->base_row);
        const int nblk = (int)((len + 63) >> 6);
        const int tb0 = __builtin_amdgcn_readfirstlane(it0->tile) * (16 * bpw);
        int npg[G];
#pragma unroll
        for (int gq = 0; gq < G; gq++) npg[gq] = __builtin_amdgcn_readfirstlane(islot[buf * G + gq].np) & 15;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.codes + (base_row >> 6) * (int64_t)BB), 0, nblk * BB, 0x00020000);
        const int so_oob = nblk * BB;
        // my blocks: tb0 + w, tb0 + w + 16, ... inside the tile and the list
        int bend = tb0 + 16 * bpw; if (bend > nblk) bend = nblk;
        int nmine = (bend - (tb0 + w) + 15) >> 4; if (nmine < 0) nmine = 0;
        // the first code blocks travel while the tables are staged
        v4u ca[RD];
#pragma unroll
        for (int dd = 0; dd < RD; dd++) {
            ca[dd] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo16, dd < nmine ? (tb0 + w + 16 * dd) * BB : so_oob, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (w == 0 && lane == 0) drawn = atomicAdd(ctr, 1u);      // the next item's index: resolved after the staging
        // ---- stage the tables, a plane (two groups) at a time: thread e -> (code e >> 4 + 64 pass, half (e >> 3) & 1, copy (e >> 2) & 1,
        // four sub-quantisers e & 3): 16 consecutive lanes write the 16 different 16-byte bank slots of ONE row — no bank conflict
        // (the 4-query form wrote 4 slots of every row: SQ_LDS_BANK_CONFLICT 4.5 % of the kernel's LDS cycles)
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            const int m4 = tid & 3, cp = (tid >> 2) & 1, hf = (tid >> 3) & 1;
            const PQRotItem* itg = &islot[buf * G + 2 * pl + hf];
            const int npq = itg->np & 15;
            const int64_t q0 = itg->q[0], q1 = itg->q[1], q2 = itg->q[2], q3 = itg->q[3];
            uint32_t in[4][4];
#pragma unroll
            for (int ps = 0; ps < 4; ps++) {
                const int c = ps * 64 + (tid >> 4);
                in[ps][0] = npq > 0 ? *reinterpret_cast<const uint32_t*>(A.lut8 + (q0 * 256 + c) * M + m4 * 4) : 0u;
                in[ps][1] = npq > 1 ? *reinterpret_cast<const uint32_t*>(A.lut8 + (q1 * 256 + c) * M + m4 * 4) : 0u;
                in[ps][2] = npq > 2 ? *reinterpret_cast<const uint32_t*>(A.lut8 + (q2 * 256 + c) * M + m4 * 4) : 0u;
                in[ps][3] = npq > 3 ? *reinterpret_cast<const uint32_t*>(A.lut8 + (q3 * 256 + c) * M + m4 * 4) : 0u;
            }
#pragma unroll
            for (int ps = 0; ps < 4; ps++) {
                const int c = ps * 64 + (tid >> 4);
                const uint32_t t0 = __builtin_amdgcn_perm(in[ps][1], in[ps][0], 0x05010400u), t1 = __builtin_amdgcn_perm(in[ps][1], in[ps][0], 0x07030602u);
                const uint32_t u0 = __builtin_amdgcn_perm(in[ps][3], in[ps][2], 0x05010400u), u1 = __builtin_amdgcn_perm(in[ps][3], in[ps][2], 0x07030602u);
                uint4 o;
                o.x = __builtin_amdgcn_perm(u0, t0, 0x05040100u) ^ 0x80808080u;
                o.y = __builtin_amdgcn_perm(u0, t0, 0x07060302u) ^ 0x80808080u;
                o.z = __builtin_amdgcn_perm(u1, t1, 0x05040100u) ^ 0x80808080u;
                o.w = __builtin_amdgcn_perm(u1, t1, 0x07060302u) ^ 0x80808080u;
                *reinterpret_cast<uint4*>(sb + pl * 65536 + c * 256 + hf * 128 + cp * 64 + m4 * 16) = o;
            }
        }
        // ---- the next item's records: requested now (the draw has returned), parked after the scan
        uint4 pre = make_uint4(0xffffffffu, 0, 0, 0);
        int i1 = 0x7fffffff;
        if (w == 0) { i1 = resolve_draw(); pre = load_records(i1); }
        int cin[G];
        uint32_t qstart[G];
#pragma unroll
        for (int gq = 0; gq < G; gq++) { cin[gq] = islot[buf * G + gq].cinit[nq4]; qstart[gq] = lcur[gq]; }
        __syncthreads();    // #2: tables staged
        // ---- scan: RD code blocks in flight per wave; a block's 16 bytes per lane become 16 look-ups in each live group's table
#pragma unroll 1
        for (int j = 0; j < nmine; j += RD) {
#pragma unroll
            for (int dd = 0; dd < RD; dd++) {
                const int b = tb0 + w + 16 * (j + dd);
                const uint32_t cw[4] = {ca[dd].x, ca[dd].y, ca[dd].z, ca[dd].w};
                __builtin_amdgcn_sched_barrier(0);
                ca[dd] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo16, j + dd + RD < nmine ? (b + 16 * RD) * BB : so_oob, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (j + dd >= nmine) continue;          // wave-uniform
#pragma unroll
                for (int gq = 0; gq < G; gq++) {
                    if (npg[gq] == 0) continue;          // wave-uniform: no query in this record
                    uint32_t gv[16];
                    if (gq < 2) {
#pragma unroll
                        for (int s2 = 0; s2 < 16; s2++)
                            gv[s2] = __builtin_amdgcn_perm(cw[s2 >> 2], gq == 0 ? R00[s2 >> 2] : R01[s2 >> 2], 0x0c0c0000u | ((uint32_t)(4 + (s2 & 3)) << 8) | (uint32_t)(s2 & 3));
                    } else {
#pragma unroll
                        for (int s2 = 0; s2 < 16; s2++)
                            gv[s2] = __builtin_amdgcn_perm(cw[s2 >> 2], gq == 2 ? R10[s2 / 3] : R11[s2 / 3], 0x0c030000u | ((uint32_t)(4 + (s2 & 3)) << 8) | (uint32_t)(s2 % 3));
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 16; s2++) gv[s2] = lds_rd32(gv[s2]);
                    const int cinit = cin[gq];
                    v4i C = {cinit, cinit, cinit, cinit};
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const v4i Av = {(int)gv[4 * t], (int)gv[4 * t + 1], (int)gv[4 * t + 2], (int)gv[4 * t + 3]};
                        C = __builtin_amdgcn_mfma_i32_16x16x64_i8(Av, Bm, C, 0, 0, 0);
                    }
                    // C[r] = cinit + sum over m of (u8 - 128) for vector 16 (n >> 2) + 4 g + r of the block and query n & 3 of record gq
                    if (__builtin_amdgcn_ballot_w64((C[0] & C[1] & C[2] & C[3]) >= 0)) {
                        const PQRotItem* itg = &islot[buf * G + gq];
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const bool cnd = C[r] >= 0;
                            if (__builtin_amdgcn_ballot_w64(cnd)) {
                                const float p_dis0 = itg->dis0[nq4], p_scale = itg->scale[nq4], p_bias = itg->bias[nq4];
                                const int64_t p_off = itg->off[nq4];
                                const uint64_t p_tau = itg->tau[nq4];
                                const int64_t pos = ((int64_t)b << 6) + 16 * (n >> 2) + 4 * g + r;
                                const float sc = p_dis0 + __fmaf_rn(p_scale, (float)(C[r] - cinit + 128 * M), p_bias);
                                const uint64_t key = (cnd && pos < len) ? make_key(sc, (uint32_t)p_off + (uint32_t)pos) : 0ull;
                                const bool pass = key > p_tau;
                                const uint64_t mq = __builtin_amdgcn_ballot_w64(pass) & QM;      // this step's survivors of MY query
                                if (pass) {
                                    const uint32_t slot = lcur[gq] + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
                                    if (slot < (uint32_t)log_cap) log_keys[(mylog_0 + 4 * gq) * (size_t)log_cap + slot] = key;   // beyond: counted, dropped -> exact re-run
                                }
                                lcur[gq] += (uint32_t)__builtin_popcountll(mq);
                            }
                        }
                    }
                }
            }
        }
        // ---- item epilogue: the wave's 4 G run descriptors (lane 4 gq + k owns record gq, query slot k; its lcur / qstart are those
        // of its own n & 3 = k), then wave 0 parks the next item's records
        if (lane < 4 * G) {
            const int gq = lane >> 2;
            uint32_t l1 = lcur[0], l0 = qstart[0];
#pragma unroll
            for (int x = 1; x < G; x++) if (gq == x) { l1 = lcur[x]; l0 = qstart[x]; }
            const uint32_t c0 = l0 < (uint32_t)log_cap ? l0 : (uint32_t)log_cap, c1 = l1 < (uint32_t)log_cap ? l1 : (uint32_t)log_cap;
            seg_desc[(((size_t)item * G + gq) * 16 + w) * 4 + (lane & 3)] =
                make_uint2((uint32_t)((mylog_0 - nq4 + lane) * (size_t)log_cap) + c0, (c1 - c0) | ((l1 > (uint32_t)log_cap && l1 > l0) ? 0x80000000u : 0u));
        }
        if (w == 0) {
            if (lane < 11 * G) reinterpret_cast<uint4*>(&islot[(buf ^ 1) * G])[lane] = pre;
            if (lane == 0) islot[(buf ^ 1) * G].pad0 = i1;
        }
    }
}


// ---------------------------------------------------------------------------------------
// Sliced layout, FOUR queries per gather — k_pq_scan_sl4 (round 6).  The eight-query scan pays where lists are long and probed by
// many queries; a handful of queries (one per list: single-query latency, batch 64) or short lists (one rank of an 8-GPU run: 3 k
// vectors per list) prefer the rotated scan's shape: all slices' tables resident, ONE pass per block.  This is that shape on the
// sliced layout: the four queries' dword table keeps 256-byte rows — slices 2 p and 2 p + 1 share plane p (bytes 0-127 / 128-255:
// bank = m % 32 either way, every ds_read_b32 gather conflict-free) — M KiB in all; per 32-vector block a wave loads its M / 32
// slices (16 bytes per lane each), forms 16 addresses per slice (one v_perm each), gathers, and the one-hot B operand routes
// K groups 0, 1 (vector i) to columns 0-3 and K groups 2, 3 (vector 16 + i) to columns 4-7: M / 8 MFMAs per block.  One 4-query
// record per item; items one ahead, static block columns — its batches have no sibling groups to keep in step.
// ---------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(1024) void k_pq_scan_sl4(PQScan8Args A, const PQRotItem* __restrict__ items, uint64_t* __restrict__ log_keys,
                                                      uint2* __restrict__ seg_desc, uint32_t* xcd_ctr, int log_cap, int tile_blocks) {
    constexpr int M = 32 * NS;
    constexpr int BB = 32 * M;
    constexpr int NPL = (NS + 1) / 2;
    constexpr int TAB = NPL * 65536;
    constexpr int RD = 2;                      // 32-vector blocks (NS KiB each) in flight per wave
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) uint32_t sl4_s[];
    uint8_t* sb = reinterpret_cast<uint8_t*>(sl4_s);
    PQRotItem* islot = reinterpret_cast<PQRotItem*>(sb + TAB);                  // [2] current / next item record
    const PQScanArgs& a = A.b;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15, n = lane & 15, nq4 = n & 3;
    const int ti = *A.total_items;
    const int per_xcd = (ti + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    int xlo = xcd * per_xcd;
    int xhi = xlo + per_xcd; if (xhi > ti) xhi = ti;
    if (xlo >= xhi) return;
    uint32_t* ctr = xcd_ctr + xcd * 32;
    int cx = xcd, hops = 0;
    unsigned drawn = 0;
    auto resolve_draw = [&]() -> int {
        for (;;) {
            const int i2 = xlo + (int)__builtin_amdgcn_readfirstlane(drawn);
            if (i2 < xhi) return i2;
            if (hops >= 7) return 0x7fffffff;
            hops++;
            cx = (cx + 1) & 7;
            xlo = cx * per_xcd;
            xhi = xlo + per_xcd; if (xhi > ti) xhi = ti;
            ctr = xcd_ctr + cx * 32;
            if (xlo >= xhi) { drawn = 0u; xlo = 0; xhi = 0; continue; }
            if (lane == 0) drawn = atomicAdd(ctr, 1u);
        }
    };
    // rotation bytes: dword slot 16 (g & 1) + ((i + s) & 15) of the row half; plane and row half are ORed in per slice
    uint32_t R[6];
#pragma unroll
    for (int r = 0; r < 6; r++) {
        uint32_t v = 0;
#pragma unroll
        for (int bb = 0; bb < 3; bb++) { const int s2 = r * 3 + bb; if (s2 < 16) v |= (uint32_t)(64 * (g & 1) + 4 * ((i + s2) & 15)) << (8 * bb); }
        R[r] = v;
    }
    const int bsel = (n < 8 && (n >> 2) == (g >> 1)) ? (1 << (8 * (n & 3))) : 0;
    const v4i Bm = {bsel, bsel, bsel, bsel};
    const int vo16 = lane * 16;
    const uint64_t QM = n < 8 ? (0x0011001100110011ull << nq4) : 0ull;       // the eight lanes (4 g x 2 vector halves) of query n & 3
    const size_t mylog_i = ((size_t)blockIdx.x * 16 + (size_t)w) * 4 + (size_t)nq4;
    uint64_t* const mylog = log_keys + mylog_i * (size_t)log_cap;
    uint32_t lcur = 0;
    const auto load_record = [&](int it_) -> uint4 {
        uint4 r0 = make_uint4(0xffffffffu, 0, 0, 0);
        if (lane < 11 && it_ != 0x7fffffff) r0 = reinterpret_cast<const uint4*>(&items[it_])[lane];
        return r0;
    };
    int item = 0;
    if (w == 0) {
        if (lane == 0) drawn = atomicAdd(ctr, 1u);
        item = resolve_draw();
        const uint4 r0 = load_record(item);
        if (lane < 11) reinterpret_cast<uint4*>(&islot[0])[lane] = r0;
        if (lane == 0) islot[0].pad0 = item;
    }
    int buf = 0;
#pragma unroll 1
    for (;; buf ^= 1) {
        __syncthreads();    // #1: every wave has left the previous item's scan (tables free), the record is in LDS
        const PQRotItem* it = &islot[buf];
        const int item_l = __builtin_amdgcn_readfirstlane(it->l);
        if (item_l == -1) break;
        item = __builtin_amdgcn_readfirstlane(it->pad0);
        const int np = __builtin_amdgcn_readfirstlane(it->np) & 15;
        const int64_t len = ((int64_t)__builtin_amdgcn_readfirstlane((int)(it->len >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)it->len);
        const int64_t base_row = ((int64_t)__builtin_amdgcn_readfirstlane((int)(it->base_row >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)it->base_row);
        const int nblk = (int)((len + 63) >> 6) << 1;
        const int tb0 = __builtin_amdgcn_readfirstlane(it->tile) * tile_blocks;
        int bend = tb0 + tile_blocks; if (bend > nblk) bend = nblk;
        const int so_oob = ((nblk + PQ_SLICED_GB - 1) / PQ_SLICED_GB) * (PQ_SLICED_GB * BB);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.codes + (base_row >> 5) * (int64_t)BB), 0, so_oob, 0x00020000);
        int nmine = bend > tb0 + w ? (bend - (tb0 + w) + 15) >> 4 : 0;
        auto blk_off = [&](int j) -> int {     // byte offset of slice 0 of my j-th block
            const int b = tb0 + w + 16 * j;
            return j < nmine ? (b / PQ_SLICED_GB) * (PQ_SLICED_GB * BB) + (b % PQ_SLICED_GB) * 1024 : so_oob;
        };
        v4u ca[RD][NS];
#pragma unroll
        for (int dd = 0; dd < RD; dd++) {
            const int so = blk_off(dd);
#pragma unroll
            for (int sl = 0; sl < NS; sl++) ca[dd][sl] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo16, so == so_oob ? so_oob : so + sl * (PQ_SLICED_GB * 1024), 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (w == 0 && lane == 0) drawn = atomicAdd(ctr, 1u);      // the next item's index: resolved after the staging
        // ---- stage the four queries' table: unit = (code, 4 consecutive m) -> 4 dwords (byte k = query k as int8); lut8 is [q][slice][code][32]
        {
            int qq[4];
#pragma unroll
            for (int k = 0; k < 4; k++) qq[k] = __builtin_amdgcn_readfirstlane(it->q[k]);
            constexpr int NU = 256 * (M / 4) / 1024;
            uint32_t in[NU][4];
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int e = tid + u * 1024;
                const int c = e / (M / 4), m4 = e - c * (M / 4);
                const uint32_t eo = (uint32_t)pq_lut8_index(0, c, m4 * 4, M, 2);
#pragma unroll
                for (int k = 0; k < 4; k++)
                    in[u][k] = k < np ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(A.lut8 + (int64_t)qq[k] * (256 * M) + eo)) : 0u;
            }
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int e = tid + u * 1024;
                const int c = e / (M / 4), m4 = e - c * (M / 4);
                const uint32_t t0 = __builtin_amdgcn_perm(in[u][1], in[u][0], 0x05010400u), t1 = __builtin_amdgcn_perm(in[u][1], in[u][0], 0x07030602u);
                const uint32_t u0 = __builtin_amdgcn_perm(in[u][3], in[u][2], 0x05010400u), u1 = __builtin_amdgcn_perm(in[u][3], in[u][2], 0x07030602u);
                uint4 o;
                o.x = __builtin_amdgcn_perm(u0, t0, 0x05040100u) ^ 0x80808080u;
                o.y = __builtin_amdgcn_perm(u0, t0, 0x07060302u) ^ 0x80808080u;
                o.z = __builtin_amdgcn_perm(u1, t1, 0x05040100u) ^ 0x80808080u;
                o.w = __builtin_amdgcn_perm(u1, t1, 0x07060302u) ^ 0x80808080u;
                const int m = m4 * 4;
                *reinterpret_cast<uint4*>(sb + (m >> 6) * 65536 + c * 256 + (m & 63) * 4) = o;       // slice (m >> 5) & 1 = the row half
            }
        }
        uint4 pre = make_uint4(0xffffffffu, 0, 0, 0);
        int i1 = 0x7fffffff;
        if (w == 0) { i1 = resolve_draw(); pre = load_record(i1); }
        const int cinit = n < 8 ? it->cinit[nq4] : -(1 << 30);
        const uint32_t qstart = lcur;
        __syncthreads();    // #2: tables staged
#pragma unroll 1
        for (int j0 = 0; j0 < nmine; j0 += RD) {
#pragma unroll
            for (int dd = 0; dd < RD; dd++) {
                const int j = j0 + dd;
                const int b = tb0 + w + 16 * j;
                // slice by slice: 16 addresses (one v_perm each), the slot's code registers refilled with the same slice of the block RD
                // ahead, 16 gathers, 4 MFMAs — the sums of all slices accumulate in C
                v4i C = {cinit, cinit, cinit, cinit};
                const int so_n = blk_off(j + RD);
#pragma unroll
                for (int sl = 0; sl < NS; sl++) {
                    uint32_t gv[16];
                    {
                        const uint32_t cw[4] = {ca[dd][sl].x, ca[dd][sl].y, ca[dd][sl].z, ca[dd][sl].w};
                        const uint32_t orv = ((uint32_t)(sl >> 1) << 24) | ((sl & 1) ? 0x00808080u : 0u);
#pragma unroll
                        for (int s2 = 0; s2 < 16; s2++)
                            gv[s2] = __builtin_amdgcn_perm(cw[s2 >> 2], R[s2 / 3] | orv, 0x0c030000u | ((uint32_t)(4 + (s2 & 3)) << 8) | (uint32_t)(s2 % 3));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    ca[dd][sl] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo16, so_n == so_oob ? so_oob : so_n + sl * (PQ_SLICED_GB * 1024), 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j < nmine) {          // wave-uniform
#pragma unroll
                        for (int s2 = 0; s2 < 16; s2++) gv[s2] = lds_rd32(gv[s2]);
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const v4i Av = {(int)gv[4 * t], (int)gv[4 * t + 1], (int)gv[4 * t + 2], (int)gv[4 * t + 3]};
                            C = __builtin_amdgcn_mfma_i32_16x16x64_i8(Av, Bm, C, 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (j >= nmine) continue;          // wave-uniform
                // C[r] (lanes n < 8) = cinit + sum over m of (u8 - 128) for vector 16 (n >> 2) + 4 g + r of the block and query n & 3
                if (__builtin_amdgcn_ballot_w64((C[0] & C[1] & C[2] & C[3]) >= 0)) {
#pragma unroll 1
                    for (int r = 0; r < 4; r++) {
                        const int Cr = r == 0 ? C[0] : r == 1 ? C[1] : r == 2 ? C[2] : C[3];
                        const bool cnd = Cr >= 0;
                        if (__builtin_amdgcn_ballot_w64(cnd)) {
                            const float p_dis0 = it->dis0[nq4], p_scale = it->scale[nq4], p_bias = it->bias[nq4];
                            const int64_t p_off = it->off[nq4];
                            const uint64_t p_tau = it->tau[nq4];
                            const uint32_t pos = ((uint32_t)b << 5) + (uint32_t)(16 * ((n >> 2) & 1) + 4 * g + r);
                            const float sc = p_dis0 + __fmaf_rn(p_scale, (float)(Cr - cinit + 128 * M), p_bias);
                            const uint64_t key = (cnd && pos < (uint32_t)len) ? make_key(sc, (uint32_t)p_off + pos) : 0ull;
                            const bool pass_ = key > p_tau;
                            const uint64_t mq = __builtin_amdgcn_ballot_w64(pass_) & QM;
                            if (pass_) {
                                const uint32_t slot_k = lcur + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
                                if (slot_k < (uint32_t)log_cap) mylog[slot_k] = key;
                            }
                            lcur += (uint32_t)__builtin_popcountll(mq);
                        }
                    }
                }
            }
        }
        if (lane < 4) {
            const uint32_t c0 = qstart < (uint32_t)log_cap ? qstart : (uint32_t)log_cap, c1 = lcur < (uint32_t)log_cap ? lcur : (uint32_t)log_cap;
            seg_desc[((size_t)item * 16 + w) * 4 + lane] =
                make_uint2((uint32_t)(mylog_i * (size_t)log_cap) + c0, (c1 - c0) | ((lcur > (uint32_t)log_cap && lcur > qstart) ? 0x80000000u : 0u));
        }
        if (w == 0) {
            if (lane < 11) reinterpret_cast<uint4*>(&islot[buf ^ 1])[lane] = pre;
            if (lane == 0) islot[buf ^ 1].pad0 = i1;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Sliced layout (PQ_SLICED, rsx_internal.h), filtered scan: EIGHT queries per table gather at M = 96 — k_pq_scan_sl8 (round 6).
// The 8-byte-entry table of eight queries is M x 2 KiB: 192 KiB at M = 96, more than a CU's 160 KiB.  The sliced layout stores a
// 32-vector block as M / 32 slices of 1 KiB (32 sub-quantisers x 32 vectors each), so a pass over ONE slice of a run of blocks needs
// only that slice's table: 64 KiB ([code][m & 31] x 8 B, 256-byte rows).  Two table slots live in LDS.  A work item's tile is cut
// into SUB-TILES of 16 waves x NB blocks; per sub-tile a wave passes over its NB blocks once per slice and PARKS the blocks'
// partial sums in registers between the passes (one MFMA tile = 32 vectors x 8 queries: lane groups g < 2 hold vector i, g >= 2
// vector 16 + i, and the one-hot B operand routes K group g to columns 8 (g >> 1) + query — all 16 columns used; after one or two
// passes a sum fits 16 bits, so two blocks share four VGPRs and NB = 16 blocks cost 32).
// Slices are visited in ZIG-ZAG order (0, 1, 2 | 2, 1, 0 | ...): slice 1 keeps slot 1 for the whole item, slices 0 and 2 alternate
// in slot 0, which is re-staged once per sub-tile BEHIND the middle pass: every wave requests its share of the slice's table (two
// units of eight dwords, one at a time) behind early blocks of the pass and writes it a few blocks later, once an LDS counter says that
// all waves have left the first pass (slot 0 is free); the third pass waits on a second counter (all shares written).  No s_barrier
// inside an item.
// What bounds it (profiles/r06_sliced_scan.md): the L2-miss traffic — codes + table slices + sibling re-reads, 12.1 GB per launch at
// the rate this chip streams 256 private streams (5.0 - 5.7 TB/s); the CU side alone is ~1.5 ms.
// Per (32 vectors, slice, 8 queries): one 16-byte code load per lane, 16 v_perm, 16 ds_read_b64 (conflict-free: the 32 lanes of a
// half-wave read 32 different 8-byte slots), 8 v_mfma_i32_16x16x64_i8 — per (vector, query, sub-quantiser) half the look-up
// instructions of the 4-query form (k_pq_scan_rot), and a list probed by 5 .. 8 queries is passed over once instead of twice.
// Work items, thresholds (accumulators start at -threshold: C >= 0 <=> survivor), survivor logs and run descriptors are those of
// k_pq_scan_rot<.., G = 2>: two 4-query records per item, everything downstream is shared.
// ---------------------------------------------------------------------------------------
template <int NS, int NB, int RD>
__global__ __launch_bounds__(1024) void k_pq_scan_sl8(PQScan8Args A, const PQRotItem* __restrict__ items, uint64_t* __restrict__ log_keys,
                                                      uint2* __restrict__ seg_desc, uint32_t* xcd_ctr, int log_cap, int tile_blocks) {
    static_assert(NS == 3, "zig-zag schedule written for three slices (M = 96)");
    static_assert((NS * NB) % RD == 0 && RD <= NB, "the prefetch slot of a step is a compile-time function of (pass, block)");
    static_assert(NB % 2 == 0, "two blocks' partial sums share a register");
    constexpr int M = 32 * NS, G = 2;
    constexpr int BB = 32 * M;                 // bytes per 32-vector block
    constexpr int TAB = 2 * 65536;             // two table slots
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) uint32_t sl8_s[];
    uint8_t* sb = reinterpret_cast<uint8_t*>(sl8_s);
    PQRotItem* islot = reinterpret_cast<PQRotItem*>(sb + TAB);                  // [2][G] current / next item's records
    constexpr uint32_t cntA_a = (uint32_t)(TAB + 2 * G * 176);                 // LDS word: waves that have left the first pass, counted over the item's sub-tiles
    constexpr uint32_t cntS_a = cntA_a + 4;                                    // LDS word: waves that have written their share of the re-staged slot
    constexpr uint32_t cntB_a = cntA_a + 8;                                    // LDS word: waves that have written their share of the item's slice-1 table
    const PQScanArgs& a = A.b;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15, n = lane & 15, nq4 = n & 3;
    // ---- items: XCD b % 8 owns a contiguous range of the list-major item order, its workgroups draw from one counter, an exhausted
    // range steals from the next XCD's (as k_pq_scan_rot)
    const int ti = *A.total_items;
    const int per_xcd = (ti + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    int xlo = xcd * per_xcd;
    int xhi = xlo + per_xcd; if (xhi > ti) xhi = ti;
    if (xlo >= xhi) return;
    uint32_t* ctr = xcd_ctr + xcd * 32;
    int cx = xcd, hops = 0;
    unsigned drawn = 0;
    auto resolve_draw = [&]() -> int {
        for (;;) {
            const int i2 = xlo + (int)__builtin_amdgcn_readfirstlane(drawn);
            if (i2 < xhi) return i2;
            if (hops >= 7) return 0x7fffffff;
            hops++;
            cx = (cx + 1) & 7;
            xlo = cx * per_xcd;
            xhi = xlo + per_xcd; if (xhi > ti) xhi = ti;
            ctr = xcd_ctr + cx * 32;
            if (xlo >= xhi) { drawn = 0u; xlo = 0; xhi = 0; continue; }
            if (lane == 0) drawn = atomicAdd(ctr, 1u);
        }
    };
    // ---- per-lane constants.  Rotation bytes: lane (g, i) reads, at step s, the 8-byte slot 16 (g & 1) + ((i + s) & 15) of its code's
    // row; three per register, the fourth byte = the table slot of the pass (0 here, 1 in Rb) -> address bit 16
    uint32_t Ra[6], Rb[6];
#pragma unroll
    for (int r = 0; r < 6; r++) {
        uint32_t v = 0;
#pragma unroll
        for (int bb = 0; bb < 3; bb++) { const int s2 = r * 3 + bb; if (s2 < 16) v |= (uint32_t)(128 * (g & 1) + 8 * ((i + s2) & 15)) << (8 * bb); }
        Ra[r] = v; Rb[r] = v | 0x01000000u;
    }
    // B one-hot: an A operand = two 8-byte gathers (dwords 0, 2: queries 0-3; 1, 3: queries 4-7) of vector i (K groups 0, 1) or 16 + i
    // (K groups 2, 3): column n = 8 (vector half) + query takes byte query & 3 of the dwords with j & 1 == query >> 2 of its half's K groups
    const bool mine = (n >> 3) == (g >> 1);
    const int bsel_lo = (mine && (n & 7) < 4) ? (1 << (8 * (n & 3))) : 0;
    const int bsel_hi = (mine && (n & 7) >= 4) ? (1 << (8 * (n & 3))) : 0;
    const v4i Bm = {bsel_lo, bsel_hi, bsel_lo, bsel_hi};
    const int vo16 = lane * 16;
    const int qn = n & 7;                                                     // my query column: record qn >> 2, slot qn & 3
    const int rq = qn >> 2;
    const uint64_t QM = 0x0101010101010101ull << qn;                          // the eight lanes (4 g x 2 vector halves) of my query
    const size_t mylog_i = ((size_t)blockIdx.x * 16 + (size_t)w) * (4 * G) + (size_t)qn;
    uint64_t* const mylog = log_keys + mylog_i * (size_t)log_cap;
    uint32_t lcur = 0;
    const auto load_records = [&](int it_) -> uint4 {
        uint4 r0 = make_uint4(0xffffffffu, 0, 0, 0);        // l = -1: end marker
        if (lane < 11 * G && it_ != 0x7fffffff) r0 = reinterpret_cast<const uint4*>(&items[(size_t)it_ * G])[lane];
        return r0;
    };
    int item = 0;
    if (w == 0) {
        if (lane == 0) drawn = atomicAdd(ctr, 1u);
        item = resolve_draw();
        const uint4 r0 = load_records(item);
        if (lane < 11 * G) reinterpret_cast<uint4*>(&islot[0])[lane] = r0;
        if (lane == 0) islot[0].pad0 = item;
    }
    int buf = 0;
#pragma unroll 1
    for (;; buf ^= 1) {
        __syncthreads();    // #1: every wave has left the previous item's scan (slots free), the records are in LDS
        const PQRotItem* it0 = &islot[buf * G];
        const PQRotItem* it1 = it0 + 1;
        const int item_l = __builtin_amdgcn_readfirstlane(it0->l);
        if (item_l == -1) break;
        item = __builtin_amdgcn_readfirstlane(it0->pad0);
        const int64_t len = ((int64_t)__builtin_amdgcn_readfirstlane((int)(it0->len >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)it0->len);
        const int64_t base_row = ((int64_t)__builtin_amdgcn_readfirstlane((int)(it0->base_row >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)it0->base_row);
        const int nblk = (int)((len + 63) >> 6) << 1;                          // 32-vector blocks of the list, slab padding included
        const int tb0 = __builtin_amdgcn_readfirstlane(it0->tile) * tile_blocks;
        int bend = tb0 + tile_blocks; if (bend > nblk) bend = nblk;
        const int np0 = __builtin_amdgcn_readfirstlane(it0->np) & 15, np1 = __builtin_amdgcn_readfirstlane(it1->np) & 15;
        // (the list starts on a group boundary: block b of the list, slice s -> pq_sliced_off(b, s); the descriptor covers whole groups)
        const int so_oob = ((nblk + PQ_SLICED_GB - 1) / PQ_SLICED_GB) * (PQ_SLICED_GB * BB);      // past the descriptor's end: reads zeros
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.codes + (base_row >> 5) * (int64_t)BB), 0, so_oob, 0x00020000);
        const int nsub = bend > tb0 ? (bend - tb0 + 16 * NB - 1) / (16 * NB) : 0;
        int qq[8];                                                              // the eight queries (SGPRs: they live through the item's re-stagings)
#pragma unroll
        for (int k = 0; k < 8; k++) qq[k] = __builtin_amdgcn_readfirstlane(k < 4 ? it0->q[k] : it1->q[k - 4]);
        // ---- staging of one slice's table into a slot: unit = (code, 4 consecutive m of the slice) -> the eight queries' dwords ->
        // four 8-byte entries (bytes 0-3: record 0's queries, 4-7: record 1's) = 32 contiguous bytes of the code's row.  Two units per
        // thread; issue (16 loads in flight) and write-out are separate so that the loads can travel behind scan blocks.
        // (the tables of query slots WITHOUT a query are not fetched — three slots in ten are empty on the bench batch, and every slice
        //  of table crosses the fabric like a code line does; their accumulators start at -2^30, whatever the slot's table bytes are.
        //  Scalar base + 32-bit lane offset per load; the offset is made opaque so that the compiler does not keep sixteen 64-bit
        //  addresses alive — spilled — across the item's passes)
        // (behind scan blocks the two units travel ONE AT A TIME — eight registers in flight instead of sixteen: they pay for the third code
        //  load in flight per wave)
        uint32_t sin[8];
        auto stage_issue = [&](int sl, int u) {
            {
                uint32_t eoff = (uint32_t)((tid + u * 1024) * 4 + sl * 8192);       // lut8 is [q][slice][code][32]: thread e takes dword e of the slice's 8 KiB
                asm volatile("" : "+v"(eoff));
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    sin[k] = 0u;
                    if (k < 4 ? k < np0 : k - 4 < np1)        // wave-uniform
                        sin[k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(A.lut8 + (int64_t)qq[k] * (256 * M) + eoff));
                }
            }
        };
        auto stage_write = [&](int slot, int u) {
            {
                const int e = tid + u * 1024;
                const int c = e >> 3, m4 = e & 7;
                uint32_t o[2][4];
#pragma unroll
                for (int hh = 0; hh < 2; hh++) {
                    const uint32_t a0 = sin[4 * hh], a1 = sin[4 * hh + 1], a2 = sin[4 * hh + 2], a3 = sin[4 * hh + 3];
                    const uint32_t t0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), t1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);
                    const uint32_t u0 = __builtin_amdgcn_perm(a3, a2, 0x05010400u), u1 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
                    o[hh][0] = __builtin_amdgcn_perm(u0, t0, 0x05040100u) ^ 0x80808080u;
                    o[hh][1] = __builtin_amdgcn_perm(u0, t0, 0x07060302u) ^ 0x80808080u;
                    o[hh][2] = __builtin_amdgcn_perm(u1, t1, 0x05040100u) ^ 0x80808080u;
                    o[hh][3] = __builtin_amdgcn_perm(u1, t1, 0x07060302u) ^ 0x80808080u;
                }
                uint8_t* dst = sb + slot * 65536 + c * 256 + m4 * 32;
                *reinterpret_cast<uint4*>(dst) = make_uint4(o[0][0], o[1][0], o[0][1], o[1][1]);
                *reinterpret_cast<uint4*>(dst + 16) = make_uint4(o[0][2], o[1][2], o[0][3], o[1][3]);
            }
        };
        // ---- the code stream: step t = (sub-tile, pass, block j) in scan order; the load of step t + RD is issued when step t's
        // codes have become gather addresses.  Block of (sub-tile st, j): tb0 + 16 (st NB + j) + w; slice of pass p: zig-zag.
        auto slice_of = [&](int st, int p) -> int { return (st & 1) ? NS - 1 - p : p; };
        auto code_off = [&](int st, int p, int j) -> int {
            const int b = tb0 + 16 * (st * NB + j) + w;
            return (st < nsub && b < bend) ? (b / PQ_SLICED_GB) * (PQ_SLICED_GB * BB) + slice_of(st, p) * (PQ_SLICED_GB * 1024) + (b % PQ_SLICED_GB) * 1024 : so_oob;
        };
        v4u ca[RD];
#pragma unroll
        for (int dd = 0; dd < RD; dd++) {
            ca[dd] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo16, code_off(0, 0, dd), SL8_AUX);
            __builtin_amdgcn_sched_barrier(0);
        }
        // only slice 0 is staged ahead of the scan; slice 1 (slot 1, needed by the second pass) travels behind the first pass of the
        // first sub-tile, like the re-stagings (cntB)
        if (nsub > 0) { stage_issue(0, 0); stage_write(0, 0); stage_issue(0, 1); stage_write(0, 1); }
        // the next item is drawn JUST IN TIME (k_pq_scan_rot, round 3): the query groups of a list are adjacent in the item order, so the
        // workgroups that draw them are the ones that come free one after the other — they start within a couple of microseconds of each
        // other, walk the same sub-tiles in the same order at the same pace, and the second finds the code lines in the XCD's L2.  (Drawn
        // one item ahead, siblings started tens of microseconds apart: 14.6 GB fetched for 9.6 GB of codes.)  Wave 0 draws at the start
        // of the item's last pass and requests the records half-way through it.
        uint4 pre = make_uint4(0xffffffffu, 0, 0, 0);
        int i1 = 0x7fffffff;
        int dstate = 0;                       // wave 0: 0 = not drawn, 1 = draw in flight, 2 = records requested
        if (w == 0 && lane == 0) { lds_wr32(cntA_a, 0u); lds_wr32(cntS_a, 0u); lds_wr32(cntB_a, 0u); }       // the item's pass counters
        const PQRotItem* itq = &islot[buf * G + rq];
        const int cinit = itq->cinit[nq4];
        const uint32_t qstart = lcur;
        __syncthreads();    // #2: slice 0 staged, counters zero
        // ---- one pass of a wave over its NB blocks of sub-tile st.  P = 0: first pass (accumulators start at -threshold), P = 1:
        // middle pass — the wave's share of the third slice's table is requested behind block JL and written into slot 0 behind block
        // JW, once every wave has left the first pass (cntA); P = 2: last pass — waits until every wave has written its share (cntS),
        // then the sums are complete: survivors.  No barrier: a wave that is early waits on a counter, nobody else does.
        // Parked partial sums: after one or two passes a sum of (u8 - 128) over <= 64 sub-quantisers lies in [-8192, 8128] — 16 bits.  Blocks
        // 2 t and 2 t + 1 share the four registers accp[t] (low / high half): a sub-tile of 16 x NB blocks costs 2 NB VGPRs, which is what
        // sets the number of table re-stagings per item (64 KiB each through the fabric).  The threshold joins in the last pass.
        uint32_t accp[NB / 2][4];
        uint32_t xacc = 0;
        auto pass = [&](auto PC, int st) {
            constexpr int P = decltype(PC)::value;
            // unit 0 of the wave's share is requested behind block JL0 and written behind JW0, unit 1 behind JL1 / JW1
            constexpr int JL0 = 1 < NB ? 1 : 0, JW0 = NB >= 8 ? NB / 2 - 1 : (NB > 2 ? 2 : NB - 1), JL1 = JW0 + 1 < NB ? JW0 + 1 : NB - 1, JW1 = NB - 2 > JL1 ? NB - 2 : NB - 1;
            const int sl = slice_of(st, P);
            const int pn = P + 1 < NS ? P + 1 : 0, stn = P + 1 < NS ? st : st + 1;       // the pass after this one (prefetch across the boundary)
#if !(SL8_VAR & 16)
            if (P == 1 && st == 0) { while ((int)(lds_rd32_volatile(cntB_a) - 16u) < 0) __builtin_amdgcn_s_sleep(1); }
#endif
#if !(SL8_VAR & 17)     // (16 = no waiting on the pass counters — races, timing only)
            if (P == 2) { while ((int)(lds_rd32_volatile(cntS_a) - (uint32_t)(16 * (st + 1))) < 0) __builtin_amdgcn_s_sleep(1); }
#endif
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const int b = tb0 + 16 * (st * NB + j) + w;
#if SL8_VAR & 32        // (32 / 64 = the sixteen waves meet before every block / every fourth block: do they fetch better in lock-step?  timing experiment)
                __builtin_amdgcn_s_barrier();
#elif SL8_VAR & 64
                if (j % 4 == 0) __builtin_amdgcn_s_barrier();
#endif
                if (P == 2 && w == 0 && st == nsub - 1) {
                    if (j == 0 && dstate == 0) { if (lane == 0) drawn = atomicAdd(ctr, 1u); dstate = 1; }
                    if (j == NB / 2 && dstate == 1) { i1 = resolve_draw(); pre = load_records(i1); dstate = 2; }
                }
#if !(SL8_VAR & 8)      // (8 = no issue-priority rotation)
                // the four waves of a SIMD (w, w + 4, w + 8, w + 12) take turns at the top issue priority, SL8_PR blocks each: the arbiter
                // favours the oldest wave otherwise, and here every wave waits for the slowest once per sub-tile
                if (j % SL8_PR == 0) {
                    switch ((j / SL8_PR + P * (NB / SL8_PR) + st + (w >> 2)) & 3) {
                        case 0: __builtin_amdgcn_s_setprio(3); break;
                        case 1: __builtin_amdgcn_s_setprio(2); break;
                        case 2: __builtin_amdgcn_s_setprio(1); break;
                        default: __builtin_amdgcn_s_setprio(0); break;
                    }
                }
#endif
                uint32_t gv[16];
#if SL8_VAR & 2         // (2 = the code stream and the table re-staging alone: no look-ups, no sums)
                xacc ^= ca[(P * NB + j) % RD].x ^ ca[(P * NB + j) % RD].y ^ ca[(P * NB + j) % RD].z ^ ca[(P * NB + j) % RD].w;
#pragma unroll
                for (int s2 = 0; s2 < 16; s2++) gv[s2] = 0;
#else
                {
                    const uint32_t cw[4] = {ca[(P * NB + j) % RD].x, ca[(P * NB + j) % RD].y, ca[(P * NB + j) % RD].z, ca[(P * NB + j) % RD].w};
#pragma unroll
                    for (int s2 = 0; s2 < 16; s2++)
                        gv[s2] = __builtin_amdgcn_perm(cw[s2 >> 2], P == 1 ? Rb[s2 / 3] : Ra[s2 / 3], 0x0c030000u | ((uint32_t)(4 + (s2 & 3)) << 8) | (uint32_t)(s2 % 3));
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                ca[(P * NB + j) % RD] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo16, j + RD < NB ? code_off(st, P, j + RD) : code_off(stn, pn, j + RD - NB), SL8_AUX);
                __builtin_amdgcn_sched_barrier(0);
                if (b < bend && !(SL8_VAR & 2)) {          // wave-uniform: inside the tile and the list (the prefetch slot has been refilled either way)
                    v4i C = {0, 0, 0, 0};
                    if (P > 0) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int v = (j & 1) ? (int)accp[j >> 1][r] >> 16 : __builtin_amdgcn_sbfe((int)accp[j >> 1][r], 0, 16);
                            C[r] = P == 2 ? v + cinit : v;
                        }
                    }
                    // sixteen 8-byte gathers in four groups, two groups in flight: the next group is requested before this one's two MFMAs
                    rot_v2u ga[4], gb[4];
#pragma unroll
                    for (int s2 = 0; s2 < 4; s2++) ga[s2] = lds_rd64(gv[s2]);
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) {
                        if (q4 < 3) {
#pragma unroll
                            for (int s2 = 0; s2 < 4; s2++) { if (q4 & 1) ga[s2] = lds_rd64(gv[4 * (q4 + 1) + s2]); else gb[s2] = lds_rd64(gv[4 * (q4 + 1) + s2]); }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (q4 & 1) {
                            C = __builtin_amdgcn_mfma_i32_16x16x64_i8(v4i{(int)gb[0].x, (int)gb[0].y, (int)gb[1].x, (int)gb[1].y}, Bm, C, 0, 0, 0);
                            C = __builtin_amdgcn_mfma_i32_16x16x64_i8(v4i{(int)gb[2].x, (int)gb[2].y, (int)gb[3].x, (int)gb[3].y}, Bm, C, 0, 0, 0);
                        } else {
                            C = __builtin_amdgcn_mfma_i32_16x16x64_i8(v4i{(int)ga[0].x, (int)ga[0].y, (int)ga[1].x, (int)ga[1].y}, Bm, C, 0, 0, 0);
                            C = __builtin_amdgcn_mfma_i32_16x16x64_i8(v4i{(int)ga[2].x, (int)ga[2].y, (int)ga[3].x, (int)ga[3].y}, Bm, C, 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (P < 2) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            if (P == 0 && !(j & 1)) accp[j >> 1][r] = (uint32_t)C[r] & 0xffffu;
                            else if (j & 1) accp[j >> 1][r] = __builtin_amdgcn_perm((uint32_t)C[r], accp[j >> 1][r], 0x05040100u);      // {acc.b0, acc.b1, C.b0, C.b1}
                            else accp[j >> 1][r] = __builtin_amdgcn_perm((uint32_t)C[r], accp[j >> 1][r], 0x03020504u);               // {C.b0, C.b1, acc.b2, acc.b3}
                        }
                    }
                    else if (__builtin_amdgcn_ballot_w64((C[0] & C[1] & C[2] & C[3]) >= 0)) {
                        // C[r] = cinit + sum over all M sub-quantisers of (u8 - 128) for vector 16 (n >> 3) + 4 g + r of block b and query n & 7
                        int b5 = b << 5;
                        asm volatile("" : "+s"(b5));      // opaque: the positions of all NB x 4 rows would otherwise be computed ahead of the pass loop and spilled
                        // (rolled: this rare path is instantiated once per unrolled block of the last pass — unrolled four ways it was two
                        //  thirds of the kernel's code, and the scan loop has to stay inside the instruction cache)
#pragma unroll 1
                        for (int r = 0; r < 4; r++) {
                            const int Cr = r == 0 ? C[0] : r == 1 ? C[1] : r == 2 ? C[2] : C[3];
                            const bool cnd = Cr >= 0;
                            if (__builtin_amdgcn_ballot_w64(cnd)) {
                                const float p_dis0 = itq->dis0[nq4], p_scale = itq->scale[nq4], p_bias = itq->bias[nq4];
                                const int64_t p_off = itq->off[nq4];
                                const uint64_t p_tau = itq->tau[nq4];
                                const uint32_t pos = (uint32_t)b5 + (uint32_t)(16 * (n >> 3) + 4 * g + r);
                                const float sc = p_dis0 + __fmaf_rn(p_scale, (float)(Cr - cinit + 128 * M), p_bias);
                                const uint64_t key = (cnd && pos < (uint32_t)len) ? make_key(sc, (uint32_t)p_off + pos) : 0ull;
                                const bool pass_ = key > p_tau;
                                const uint64_t mq = __builtin_amdgcn_ballot_w64(pass_) & QM;      // this step's survivors of MY query
                                if (pass_) {
                                    const uint32_t slot_k = lcur + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
                                    if (slot_k < (uint32_t)log_cap) mylog[slot_k] = key;   // beyond: counted, dropped -> the query is re-run exactly
                                }
                                lcur += (uint32_t)__builtin_popcountll(mq);
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (P == 0 && st == 0) {
                    if (j == JL0) stage_issue(1, 0);
                    if (j == JW0) stage_write(1, 0);
                    if (j == JL1) stage_issue(1, 1);
                    if (j == JW1) stage_write(1, 1);
                }
#if !(SL8_VAR & 1)      // (cost-split builds of tools/build_variant.sh: 1 = no re-staging — wrong sums, timing only)
                if (P == 1) {
                    if (j == JL0) stage_issue(slice_of(st, 2), 0);
                    if (j == JW0) {
#if !(SL8_VAR & 16)
                        while ((int)(lds_rd32_volatile(cntA_a) - (uint32_t)(16 * (st + 1))) < 0) __builtin_amdgcn_s_sleep(1);
#endif
                        stage_write(0, 0);
                    }
                    if (j == JL1) stage_issue(slice_of(st, 2), 1);
                    if (j == JW1) stage_write(0, 1);
                }
#endif
            }
            if (P == 0 && st == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(sb + cntB_a), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (P == 0 && lane == 0) __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(sb + cntA_a), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (P == 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's table writes have left before it reports them
                if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(sb + cntS_a), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
#pragma unroll 1
        for (int st = 0; st < nsub; st++) {
            pass(std::integral_constant<int, 0>{}, st);
            pass(std::integral_constant<int, 1>{}, st);
            pass(std::integral_constant<int, 2>{}, st);
        }
        __builtin_amdgcn_s_setprio(0);
        if ((SL8_VAR & 2) && xacc == 0x12345678u) mylog[0] = xacc;
        // ---- item epilogue: the wave's 4 G run descriptors (lane = 4 record + slot = its own query column, g = 0, vector half 0), then
        // wave 0 parks the next item's records
        if (lane < 4 * G) {
            const uint32_t c0 = qstart < (uint32_t)log_cap ? qstart : (uint32_t)log_cap, c1 = lcur < (uint32_t)log_cap ? lcur : (uint32_t)log_cap;
            seg_desc[(((size_t)item * G + (size_t)(lane >> 2)) * 16 + w) * 4 + (lane & 3)] =
                make_uint2((uint32_t)(mylog_i * (size_t)log_cap) + c0, (c1 - c0) | ((lcur > (uint32_t)log_cap && lcur > qstart) ? 0x80000000u : 0u));
        }
        if (w == 0) {
            if (dstate == 0) { if (lane == 0) drawn = atomicAdd(ctr, 1u); dstate = 1; }     // empty items
            if (dstate == 1) { i1 = resolve_draw(); pre = load_records(i1); }
            if (lane < 11 * G) reinterpret_cast<uint4*>(&islot[(buf ^ 1) * G])[lane] = pre;
            if (lane == 0) islot[(buf ^ 1) * G].pad0 = i1;
        }
    }
}

// One wave per work item: append the item's (wave, query) survivor segments to the candidate rows of its queries — the only
// atomics of the filtered scan live here, one reservation per (item, query), in a kernel with thousands of independent waves.
__global__ __launch_bounds__(64 * ROT_CW) void k_pq_rot_compact(const PQRotItem* __restrict__ items, const int32_t* total_items, int ngq,
                                                       const uint64_t* __restrict__ log_keys, const uint2* __restrict__ seg_desc,
                                                       uint64_t* cand, unsigned long long* cand_cnt, int cand_cap) {
    const int item = blockIdx.x * ROT_CW + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;     // item = RECORD here (ngq records per work item)
    if (item >= *total_items * ngq) return;
    const int w = lane >> 2, k = lane & 3;
    const uint2 dsc = seg_desc[(size_t)item * 64 + lane];                  // lane = (wave, query) run of the item
    const uint32_t c = dsc.y & 0x7fffffffu;
    const uint64_t ovm = __builtin_amdgcn_ballot_w64((dsc.y >> 31) != 0u);        // runs that dropped keys (all 64 lanes vote)
    if (__builtin_amdgcn_ballot_w64(c != 0u) == 0ull && ovm == 0ull) return;
    // exclusive prefix over the 16 waves of the same query (lanes k, k + 4, ...), and the query's total
    uint32_t incl = c;
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) { const uint32_t y = __shfl_up(incl, off); if (lane >= off) incl += y; }
    const uint32_t total = __shfl(incl, 60 + k);
    const bool myover = (ovm & (0x1111111111111111ull << k)) != 0ull;              // ... any of them of MY query
    const int64_t q = items[item].q[k];
    unsigned long long base = 0;
    if (w == 0 && (total > 0 || myover)) {
        // a full log dropped keys: push the row's count past its capacity so that k_finalize flags the query
        base = atomicAdd(&cand_cnt[q * CCS], (unsigned long long)total + (myover ? (unsigned long long)cand_cap + 1ull : 0ull));
    }
    base = __shfl(base, k) + (incl - c);
    // copy-out.  Short runs — the usual case since the thresholds are tight: one or two survivors per (wave, query) of an item —
    // are moved by their own lane, all 64 runs in ONE round trip
    if (c != 0u && c <= 4u && base < (unsigned long long)cand_cap) {
        const unsigned long long room = (unsigned long long)cand_cap - base;
        const uint32_t ce = (unsigned long long)c < room ? c : (uint32_t)room;
        const uint64_t* src = log_keys + dsc.x;
        uint64_t kk[4];
#pragma unroll
        for (int e = 0; e < 4; e++) kk[e] = (uint32_t)e < ce ? src[e] : 0ull;
        uint64_t* dst = cand + q * cand_cap + base;
#pragma unroll
        for (int e = 0; e < 4; e++) if ((uint32_t)e < ce) dst[e] = kk[e];
    }
    // longer runs: the wave walks them, all lanes on one run at a time (coalesced 8-byte moves)
    uint64_t live = __builtin_amdgcn_ballot_w64(c > 4u);
    while (live) {
        const int sgm = __builtin_ctzll(live);
        live &= live - 1;
        const uint32_t cs = __builtin_amdgcn_readlane(c, sgm);
        const uint32_t st0 = __builtin_amdgcn_readlane(dsc.x, sgm);
        const unsigned long long bs = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(base >> 32), sgm) << 32) |
                                      (uint32_t)__builtin_amdgcn_readlane((int)base, sgm);
        const int64_t qs = items[item].q[sgm & 3];
        const uint64_t* src = log_keys + st0;
        // a row past its capacity is re-run exactly anyway (k_finalize flags it): nothing beyond cand_cap is moved
        if (bs >= (unsigned long long)cand_cap) continue;
        const uint32_t room = (uint32_t)((unsigned long long)cand_cap - bs);
        const uint32_t ce = cs < room ? cs : room;
        uint64_t* dst = cand + qs * cand_cap + bs;
        uint32_t e = lane;
        for (; e + 192 < ce; e += 256) {       // four independent loads in flight per lane
            const uint64_t k0 = src[e], k1 = src[e + 64], k2 = src[e + 128], k3 = src[e + 192];
            dst[e] = k0; dst[e + 64] = k1; dst[e + 128] = k2; dst[e + 192] = k3;
        }
        for (; e < ce; e += 64) dst[e] = src[e];
    }
}

// persistent workgroups of the scan on the current device: one per CU, a multiple of 8 (workgroup b serves XCD b % 8).  (Two per CU
// were measured for M = 16 — its 64 KiB table leaves room, 64 VGPRs per lane: 4.10 vs 4.01 ms at the reference's nprobe 512.  That
// scan is bound by the L2 -> CU traffic of its 16 query groups per list re-reading the codes, not by item turnover.)
int pq_scan_rot_max_wgs(int M) {
    static int ncu_of[64] = {};
    static DevOnce once;
    int& ncu = ncu_of[cur_device()];
    once.once([&] {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ncu = pr.multiProcessorCount;
    });
    if (ncu <= 0) ncu = 256;
    (void)M;
    return (ncu + 7) & ~7;
}

// G = records per work item (2: the 8-query form); A.max_items counts WORK ITEMS, the workspace holds G records each
template <int NF, int NH, bool FILTER, int NQ = 0, int G = 1>
static int launch_pq_scan_rot_t(const PQScan8Args& A, int bpw, void* desc_ws, int log_cap, hipStream_t st) {
    constexpr int M = 64 * NF + 32 * NH + 16 * NQ;
    // tables | 2 x G item records (176 B each) + pacing words | sibling progress [64]
    const size_t shm = (size_t)(G == 2 ? 2 : NF + NH + NQ) * 65536 + (size_t)((2 * G * 176 + 8 + 31) & ~31) + 256;
    static DevOnce once;
    static std::atomic<int> failed{0};
    once.once([&] {
        if (hipFuncSetAttribute((const void*)k_pq_scan_rot<NF, NH, FILTER, NQ, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) failed = 1;
    });
    if (failed) return -1;
    const int nwg = pq_scan_rot_max_wgs(M);
    const int64_t recs = (int64_t)A.max_items * G;
    PQRotItem* items = reinterpret_cast<PQRotItem*>(desc_ws);
    uint2* seg_desc = pq_scan_rot_ws_desc(desc_ws, recs);
    uint64_t* log_keys = pq_scan_rot_ws_keys(desc_ws, recs);
    uint32_t* xcd_ctr = pq_scan_rot_ws_ctr(desc_ws, recs, log_cap, nwg * G);
    uint32_t* prog = xcd_ctr + 256;
    { const int ll_ = A.nlist + 1 <= 12288 ? A.nlist + 1 : 0; hipLaunchKernelGGL((k_pq_rot_items<M, FILTER>), dim3((unsigned)((recs + 255) / 256)), dim3(256), (size_t)ll_ * 4, st, A, items, xcd_ctr, prog, G, ll_); }
    static const int var = measure_env("RSX_ROT_VARIANT", 0);
    // one persistent workgroup per CU; never more than the work items
    int64_t grid = nwg;
    if (grid > ((A.max_items + 7) & ~7)) grid = (A.max_items + 7) & ~7;
    hipLaunchKernelGGL((k_pq_scan_rot<NF, NH, FILTER, NQ, G>), dim3((unsigned)grid), dim3(1024), shm, st, A, items, log_keys, seg_desc, xcd_ctr, prog,
                       log_cap, bpw, A.pace, var);
    if (FILTER && !A.qitems)     // with qitems the runs are consumed in place by k_pq_gather_select
        hipLaunchKernelGGL(k_pq_rot_compact, dim3((unsigned)((recs + ROT_CW - 1) / ROT_CW)), dim3(64 * ROT_CW), 0, st, items, A.total_items, G, log_keys, seg_desc,
                           A.cand, A.cand_cnt, A.cand_cap);
    return 0;
}

// 4-query records per work item of the filtered scan: M = 16: sixteen queries (k_pq_scan_rot16); M = 64 with the 8-byte-entry table
// (q8): eight (k_pq_scan_rot<1, 0, true, 0, 2>)
int pq_scan_rot_ngq(int M, bool filtered, int q8) { return !filtered ? 1 : M == 16 ? R16_G : (M == 64 && q8) ? 2 : 1; }
static int launch_pq_scan_rot16(const PQScan8Args& A, int bpw, void* desc_ws, int log_cap, hipStream_t st) {
    constexpr int G = R16_G;
    const size_t shm = (size_t)2 * 65536 + (size_t)2 * G * 176 + 64;
    static DevOnce once;
    static std::atomic<int> failed{0};
    once.once([&] {
        if (hipFuncSetAttribute((const void*)k_pq_scan_rot16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) failed = 1;
    });
    if (failed) return -1;
    const int nwg = pq_scan_rot_max_wgs(16);
    const int64_t recs = (int64_t)A.max_items * G;
    PQRotItem* items = reinterpret_cast<PQRotItem*>(desc_ws);
    uint2* seg_desc = pq_scan_rot_ws_desc(desc_ws, recs);
    uint64_t* log_keys = pq_scan_rot_ws_keys(desc_ws, recs);
    uint32_t* xcd_ctr = pq_scan_rot_ws_ctr(desc_ws, recs, log_cap, nwg * G);
    uint32_t* prog = xcd_ctr + 256;
    { const int ll_ = A.nlist + 1 <= 12288 ? A.nlist + 1 : 0; hipLaunchKernelGGL((k_pq_rot_items<16, true>), dim3((unsigned)((recs + 255) / 256)), dim3(256), (size_t)ll_ * 4, st, A, items, xcd_ctr, prog, G, ll_); }
    int64_t grid = nwg;
    if (grid > ((A.max_items + 7) & ~7)) grid = (A.max_items + 7) & ~7;
    hipLaunchKernelGGL(k_pq_scan_rot16, dim3((unsigned)grid), dim3(1024), shm, st, A, items, log_keys, seg_desc, xcd_ctr, log_cap, bpw);
    if (!A.qitems)
        hipLaunchKernelGGL(k_pq_rot_compact, dim3((unsigned)((recs + ROT_CW - 1) / ROT_CW)), dim3(64 * ROT_CW), 0, st, items, A.total_items, G, log_keys, seg_desc,
                           A.cand, A.cand_cnt, A.cand_cap);
    return 0;
}

template <int NS, int NB, int RD>
static int launch_pq_scan_sl8_t(const PQScan8Args& A, int vpl, void* desc_ws, int log_cap, hipStream_t st) {
    constexpr int G = 2, M = 32 * NS;
    const size_t shm = (size_t)2 * 65536 + (size_t)2 * G * 176 + 64;
    static DevOnce once;
    static std::atomic<int> failed{0};
    once.once([&] {
        if (hipFuncSetAttribute((const void*)k_pq_scan_sl8<NS, NB, RD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) failed = 1;
    });
    if (failed) return -1;
    const int nwg = pq_scan_rot_max_wgs(M);
    const int64_t recs = (int64_t)A.max_items * G;
    PQRotItem* items = reinterpret_cast<PQRotItem*>(desc_ws);
    uint2* seg_desc = pq_scan_rot_ws_desc(desc_ws, recs);
    uint64_t* log_keys = pq_scan_rot_ws_keys(desc_ws, recs);
    uint32_t* xcd_ctr = pq_scan_rot_ws_ctr(desc_ws, recs, log_cap, nwg * G);
    uint32_t* prog = xcd_ctr + 256;
    { const int ll_ = A.nlist + 1 <= 12288 ? A.nlist + 1 : 0; hipLaunchKernelGGL((k_pq_rot_items<M, true>), dim3((unsigned)((recs + 255) / 256)), dim3(256), (size_t)ll_ * 4, st, A, items, xcd_ctr, prog, G, ll_); }
    int64_t grid = nwg;
    if (grid > ((A.max_items + 7) & ~7)) grid = (A.max_items + 7) & ~7;
    hipLaunchKernelGGL((k_pq_scan_sl8<NS, NB, RD>), dim3((unsigned)grid), dim3(1024), shm, st, A, items, log_keys, seg_desc, xcd_ctr, log_cap, 32 * vpl);
    if (!A.qitems)
        hipLaunchKernelGGL(k_pq_rot_compact, dim3((unsigned)((recs + ROT_CW - 1) / ROT_CW)), dim3(64 * ROT_CW), 0, st, items, A.total_items, G, log_keys, seg_desc,
                           A.cand, A.cand_cnt, A.cand_cap);
    return 0;
}

template <int NS>
static int launch_pq_scan_sl4_t(const PQScan8Args& A, int vpl, void* desc_ws, int log_cap, hipStream_t st) {
    constexpr int M = 32 * NS;
    const size_t shm = (size_t)((NS + 1) / 2) * 65536 + (size_t)2 * 176 + 64;
    static DevOnce once;
    static std::atomic<int> failed{0};
    once.once([&] {
        if (hipFuncSetAttribute((const void*)k_pq_scan_sl4<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) failed = 1;
    });
    if (failed) return -1;
    const int nwg = pq_scan_rot_max_wgs(M);
    PQRotItem* items = reinterpret_cast<PQRotItem*>(desc_ws);
    uint2* seg_desc = pq_scan_rot_ws_desc(desc_ws, A.max_items);
    uint64_t* log_keys = pq_scan_rot_ws_keys(desc_ws, A.max_items);
    uint32_t* xcd_ctr = pq_scan_rot_ws_ctr(desc_ws, A.max_items, log_cap, nwg);
    uint32_t* prog = xcd_ctr + 256;
    { const int ll_ = A.nlist + 1 <= 12288 ? A.nlist + 1 : 0; hipLaunchKernelGGL((k_pq_rot_items<M, true>), dim3((unsigned)((A.max_items + 255) / 256)), dim3(256), (size_t)ll_ * 4, st, A, items, xcd_ctr, prog, 1, ll_); }
    int64_t grid = nwg;
    if (grid > ((A.max_items + 7) & ~7)) grid = (A.max_items + 7) & ~7;
    hipLaunchKernelGGL((k_pq_scan_sl4<NS>), dim3((unsigned)grid), dim3(1024), shm, st, A, items, log_keys, seg_desc, xcd_ctr, log_cap, 32 * vpl);
    if (!A.qitems)
        hipLaunchKernelGGL(k_pq_rot_compact, dim3((unsigned)((A.max_items + ROT_CW - 1) / ROT_CW)), dim3(64 * ROT_CW), 0, st, items, A.total_items, 1, log_keys, seg_desc,
                           A.cand, A.cand_cnt, A.cand_cap);
    return 0;
}

// returns 0 on launch, -1 if this M has no rotated kernel.  tau_key == null: unfiltered (every score to a.temp).
int launch_pq_scan_rot(const PQScanArgs& a, const uint8_t* lut8t, const void* qparam, const int32_t* pairs_sorted,
                       const int32_t* pair_off, const int32_t* group_off, const int32_t* total_groups,
                       const int32_t* item_off, const int32_t* total_items, int nlist, int64_t max_items, int vpl,
                       const uint64_t* tau_key, int64_t tau_stride, uint64_t* cand, unsigned long long* cand_cnt,
                       int cand_cap, void* item_ws, int log_cap, int prune, int pace, const uint16_t* excl, int32_t* qitems,
                       int qitems_tmax, hipStream_t st, int q8) {
    if (!pq_rot_family(a.CB) || !item_ws || log_cap <= 0 || !pq_rot_applies(a.M) || a.M != a.Mpad || max_items <= 0 || max_items > 0x7fffff00) return -1;
    if (a.CB == PQ_SLICED && (!tau_key || !pq_sliced_applies(a.M))) return -1;      // the sliced layout has the filtered 8-query scan only
    PQScan8Args A;
    A.b = a; A.lut8 = lut8t; A.qp = (const PQQParam*)qparam;
    A.pairs_sorted = pairs_sorted; A.pair_off = pair_off; A.group_off = group_off; A.total_groups = total_groups;
    A.item_off = item_off; A.total_items = total_items;
    A.nlist = nlist; A.max_items = (int)max_items;
    A.tau_key = tau_key; A.tau_stride = tau_stride; A.cand = cand; A.cand_cnt = cand_cnt; A.cand_cap = cand_cap;
    A.prune = prune; A.pace = pace; A.excl = excl; A.qitems = qitems; A.qitems_tmax = qitems_tmax;
    const int bpw = 4 * vpl;   // tile = 16 waves x bpw blocks x 16 vectors = 1024 vpl vectors, as k_pq_scan8's
    const bool f = tau_key != nullptr;
    if (a.CB == PQ_SLICED) return q8 ? launch_pq_scan_sl8_t<3, SL8_NB, SL8_RD>(A, vpl, item_ws, log_cap, st) : launch_pq_scan_sl4_t<3>(A, vpl, item_ws, log_cap, st);
    switch (a.M) {
        case 16: return f ? launch_pq_scan_rot16(A, vpl, item_ws, log_cap, st) : launch_pq_scan_rot_t<0, 0, false, 1>(A, vpl, item_ws, log_cap, st);   // 64-vector blocks
        case 32: return f ? launch_pq_scan_rot_t<0, 1, true>(A, bpw, item_ws, log_cap, st) : launch_pq_scan_rot_t<0, 1, false>(A, bpw, item_ws, log_cap, st);
        case 64: return f ? (q8 ? launch_pq_scan_rot_t<1, 0, true, 0, 2>(A, bpw, item_ws, log_cap, st) : launch_pq_scan_rot_t<1, 0, true>(A, bpw, item_ws, log_cap, st))
                          : launch_pq_scan_rot_t<1, 0, false>(A, bpw, item_ws, log_cap, st);
        case 96: return f ? launch_pq_scan_rot_t<1, 1, true>(A, bpw, item_ws, log_cap, st) : launch_pq_scan_rot_t<1, 1, false>(A, bpw, item_ws, log_cap, st);
        case 128: return f ? launch_pq_scan_rot_t<2, 0, true>(A, bpw, item_ws, log_cap, st) : launch_pq_scan_rot_t<2, 0, false>(A, bpw, item_ws, log_cap, st);
        default: return -1;
    }
}

// ---------------------------------------------------------------------------------------
// Threshold pre-pass on the scan's own machinery (round 3): FOUR queries per workgroup share one table image in the scan's
// format (one dword = the four queries' bytes), each query's sample — the first pre_rows vectors of ITS closest list — is
// scanned by a group of four waves with the scan's conflict-free gathers and the MFMA adder, the 16-bit integer sums go to LDS,
// and the k-th largest of each sample gives the query's threshold exactly as k_pq_prepass (k_select.hip) derives it:
// tau = a_k - 2 eps.  k_pq_prepass scores the same rows with byte gathers on a per-query byte table (~3.5-way bank
// conflicts, one query per gather): 137 us per 1024 queries x 4096 rows against ~50 here.  No emission (small k only: the
// caller keeps k_pq_prepass when the sample has to cover the first scan tile).
// ---------------------------------------------------------------------------------------
#ifdef RSX_MEASURE
__device__ uint64_t g_pp4_trace[64 * 16 * 8];          // [workgroup < 64][wave][mark]
extern "C" int rsx_debug_pp4_trace(uint64_t* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp4_trace), sizeof(uint64_t) * 64 * 16 * 8) == hipSuccess ? 0 : -1; }
#define PP4_MARK(i) do { if (blockIdx.x < 64 && lane == 0) g_pp4_trace[(blockIdx.x * 16 + w) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PP4_MARK(i)
#endif
// BIG (round 4, the reference's n_docs = 1000 / 2000 on the headline index): samples of any size, spanning several lists.  The
// 16-bit sums are not kept — every (sum + 1) goes straight into a per-query histogram of (sum + 1) >> PP4_SH (16-bit counters,
// two per dword: a sample holds <= 32768 rows) and the threshold is taken at the LOWER EDGE of the bin that holds the k-th
// largest: a lower bound of the k-th largest sum (<= 7 of ~24 k integer units below it), hence still a valid a_k.  A sample that
// the closest list cannot fill continues in the next closest lists with k_pq_prepass's shifted sums (a lower bound of the
// vector's approximate score relative to the closest list's dis0).  k_pq_prepass scores such samples with byte gathers on a
// per-query byte table: 0.77 ms per 1024 queries x 16384 rows; this form shares the table image between four queries and uses
// the scan's conflict-free gathers and the MFMA adder.
constexpr int PP4_MAXSEG = 8;
// SL > 0 (round 6): the SLICED layout (PQ_SLICED) with SL slices: 32-vector blocks, a 1 KiB slice = 16 bytes per lane; the four-query
// table image keeps 256-byte rows — slices 2 p and 2 p + 1 share plane p (bytes 0-127 / 128-255: bank = m % 32 either way) — and an
// MFMA tile holds 32 vectors x 4 queries (K groups 0, 1: vector i -> columns 0-3; K groups 2, 3: vector 16 + i -> columns 4-7).
template <int NF, int NH, bool BIG = false, int SL = 0>
__global__ __launch_bounds__(1024) void k_pq_prepass4(PQPrepassArgs a, int64_t nq) {
    constexpr int M = SL ? 32 * SL : 64 * NF + 32 * NH;
    constexpr int PP4_SH = M > 96 ? 4 : 3;
    constexpr int NB = ((255 * M + 1) >> PP4_SH) + 1;        // bins of (sum + 1) >> PP4_SH
    constexpr int NBW = (NB + 1) / 2;                        // dwords of one query's histogram
    constexpr int NPH = SL ? (SL + 1) / 2 : NF + NH;
    constexpr int TAB = NPH * 65536;
    constexpr int NG = SL ? 16 * SL : M / 4;               // gathers per lane and block (SL: 32-vector blocks)
    constexpr int NR1 = SL ? 6 : NPH > 1 ? rot_nreg(1, NF >= 2 ? 16 : 8) : 0;
    constexpr int NR0 = SL ? 0 : NF >= 1 ? 4 : rot_nreg(0, 8);
    constexpr int BVEC = SL ? 32 : 16;                       // vectors per code block
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) uint32_t pp4_s[];
    uint8_t* sb = reinterpret_cast<uint8_t*>(pp4_s);
    uint16_t* sums = reinterpret_cast<uint16_t*>(sb + TAB);                               // [4][pre_rows]: integer sum + 1, 0 = no vector
    int32_t* hist = reinterpret_cast<int32_t*>(sb + TAB + (BIG ? (size_t)0 : (size_t)4 * a.pre_rows * 2));    // [4][256] | BIG: [4][NBW] packed 16-bit counters
    int32_t* ctl = hist + (BIG ? 4 * NBW : 1024);                                         // [4][8]
    int32_t* segs = ctl + 32;                                                             // BIG: [4][PP4_MAXSEG][4] list, rows, shift, -; then [4] counts
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15, n = lane & 15;
    const int grp = w >> 2, wq = w & 3;                  // query slot of this wave, wave within the slot
    PP4_MARK(0);
    const int64_t q0 = (int64_t)blockIdx.x * 4;
    const int np = (int)((nq - q0) < 4 ? (nq - q0) : 4);
    const int64_t q = q0 + (grp < np ? grp : 0);
    // ---- rotation bytes and the one-hot B operand, as in k_pq_scan_rot
    uint32_t R0[NR0 > 0 ? NR0 : 1], R1[NR1 > 0 ? NR1 : 1];
#pragma unroll
    for (int r = 0; r < NR0; r++) {
        uint32_t v = 0;
#pragma unroll
        for (int bb = 0; bb < 4; bb++) {
            const int s = r * 4 + bb;
            const uint32_t rot = NF >= 1 ? (uint32_t)(64 * g + 4 * ((i + s) & 15))
                                         : (uint32_t)(64 * (g & 1) + 4 * ((i + s + 8 * (g >> 1)) & 15));
            v |= rot << (8 * bb);
        }
        R0[r] = v;
    }
#pragma unroll
    for (int r = 0; r < NR1; r++) {
        uint32_t v = SL ? 0u : 0x01000000u;              // SL: plane / row-half bits are ORed in per slice
#pragma unroll
        for (int bb = 0; bb < 3; bb++) {
            const int s = r * 3 + bb;
            const uint32_t rot = SL ? (s < 16 ? (uint32_t)(64 * (g & 1) + 4 * ((i + s) & 15)) : 0u)       // dword slot 16 (g & 1) + ((i + s) & 15) of the row half
                               : NF >= 2 ? (uint32_t)(64 * g + 4 * ((i + s) & 15))
                                         : (uint32_t)(64 * (g & 1) + 4 * ((i + s + 8 * (g >> 1)) & 15));
            v |= (rot & 255u) << (8 * bb);
        }
        R1[r] = v;
    }
    const int bsel = SL ? ((n < 8 && (n >> 2) == (g >> 1)) ? (1 << (8 * (n & 3))) : 0) : (n < 4 ? (1 << (8 * n)) : 0);
    const v4i Bm = {bsel, bsel, bsel, bsel};
    // ---- the four tables, staged exactly like a scan item's (unit = (code, 4 consecutive m) -> 4 dwords, byte k = query k as int8)
    {
        const int64_t qa = q0, qb = q0 + (np > 1 ? 1 : 0), qc = q0 + (np > 2 ? 2 : 0), qd = q0 + (np > 3 ? 3 : 0);
        constexpr int NU = (256 * (M / 4) + 1023) / 1024;
        constexpr int nunits = 256 * (M / 4);
        uint32_t in[NU][4];
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int e = tid + u * 1024;
            const int ee = e < nunits ? e : 0;
            const int c = ee / (M / 4), m4 = ee - c * (M / 4);
            const int64_t eo = SL ? pq_lut8_index(0, c, m4 * 4, M, 2) : (int64_t)c * M + m4 * 4;      // within a query's 256 M table bytes
            in[u][0] = *reinterpret_cast<const uint32_t*>(a.lut8 + qa * 256 * M + eo);
            in[u][1] = np > 1 ? *reinterpret_cast<const uint32_t*>(a.lut8 + qb * 256 * M + eo) : 0u;
            in[u][2] = np > 2 ? *reinterpret_cast<const uint32_t*>(a.lut8 + qc * 256 * M + eo) : 0u;
            in[u][3] = np > 3 ? *reinterpret_cast<const uint32_t*>(a.lut8 + qd * 256 * M + eo) : 0u;
        }
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int e = tid + u * 1024;
            if (e >= nunits) continue;
            const int c = e / (M / 4), m4 = e - c * (M / 4);
            const uint32_t t0 = __builtin_amdgcn_perm(in[u][1], in[u][0], 0x05010400u), t1 = __builtin_amdgcn_perm(in[u][1], in[u][0], 0x07030602u);
            const uint32_t u0 = __builtin_amdgcn_perm(in[u][3], in[u][2], 0x05010400u), u1 = __builtin_amdgcn_perm(in[u][3], in[u][2], 0x07030602u);
            uint4 o;
            o.x = __builtin_amdgcn_perm(u0, t0, 0x05040100u) ^ 0x80808080u;
            o.y = __builtin_amdgcn_perm(u0, t0, 0x07060302u) ^ 0x80808080u;
            o.z = __builtin_amdgcn_perm(u1, t1, 0x05040100u) ^ 0x80808080u;
            o.w = __builtin_amdgcn_perm(u1, t1, 0x07060302u) ^ 0x80808080u;
            const int m = m4 * 4;
            const int p = SL ? (m >> 6) : m < 64 * NF ? (m >> 6) : NF;
            const int slot = SL ? (m & 63) : m < 64 * NF ? (m & 63) : (m - 64 * NF);       // SL: slice (m >> 5) & 1 = the row half
            *reinterpret_cast<uint4*>(sb + p * 65536 + c * 256 + slot * 4) = o;
        }
    }
    for (int t = tid; t < (BIG ? 4 * NBW : 1024) + 32; t += 1024) hist[t] = 0;
    // ---- my query's closest probed list that holds vectors here (wave-uniform)
    int32_t l = -1; int64_t len = 0; int j0 = 0;
    if (grp < np) {
        for (int j = 0; j < a.nprobe; j++) {
            const int32_t lj = a.probe_list[q * a.nprobe + j];
            if (lj >= 0 && a.list_len[lj] > 0) { l = lj; len = a.list_len[lj]; j0 = j; break; }
        }
    }
    l = __builtin_amdgcn_readfirstlane(l); j0 = __builtin_amdgcn_readfirstlane(j0);
    int nrows = (int)(len < a.pre_rows ? len : a.pre_rows);
    int nblk = __builtin_amdgcn_readfirstlane((nrows + BVEC - 1) / BVEC);
    const uint8_t* lp = a.codes + ((l >= 0 ? a.list_base[l] : 0) / BVEC) * (int64_t)(BVEC * M);
    uint16_t* mysum = sums + (size_t)grp * a.pre_rows;
    int nseg = 1;
    if (BIG) {
        // the slot's sample: the closest list first, then (k_pq_prepass's rule) the next closest ones until it holds min(pre_rows, 8 k) rows
        int32_t* sg = segs + grp * (PP4_MAXSEG * 4);
        if (wq == 0 && lane == 0) {
            int ns = 0, off = 0;
            if (l >= 0 && grp < np) {
                sg[0] = l; sg[1] = nrows; sg[2] = 0; ns = 1; off = nrows;
                const float scale = a.qparam[q * 4 + 0];
                const float dis0 = a.probe_dis0[q * a.nprobe + j0];
                const int want = a.pre_rows < 8 * a.k ? a.pre_rows : 8 * a.k;
                if (scale > 0.0f)
                    for (int j = j0 + 1; j < a.nprobe && ns < PP4_MAXSEG && off < want; j++) {
                        const int32_t lj = a.probe_list[q * a.nprobe + j];
                        if (lj < 0) continue;
                        const int64_t len_j = a.list_len[lj];
                        if (len_j <= 0) continue;
                        const float sh = floorf((a.probe_dis0[q * a.nprobe + j] - dis0) / scale) - 2.0f;
                        if (!(sh > -1.0e9f)) break;
                        const int rows = (int)(len_j < (int64_t)(a.pre_rows - off) ? len_j : (int64_t)(a.pre_rows - off));
                        sg[4 * ns + 0] = lj; sg[4 * ns + 1] = rows; sg[4 * ns + 2] = sh < 0.0f ? (int)sh : 0;
                        off += rows;
                        ns++;
                    }
            }
            segs[4 * PP4_MAXSEG * 4 + grp] = ns;
        }
    }
    PP4_MARK(1);
    __syncthreads();        // tables staged, histograms zero
    PP4_MARK(2);
    if (BIG) nseg = __builtin_amdgcn_readfirstlane(segs[4 * PP4_MAXSEG * 4 + grp]);
    int shift = 0;
    uint32_t* hq = reinterpret_cast<uint32_t*>(hist) + grp * NBW;
#pragma unroll 1
    for (int sgi = 0; sgi < nseg; sgi++) {
    if (BIG) {
        const int32_t* sg = segs + grp * (PP4_MAXSEG * 4) + 4 * sgi;
        const int32_t ls = __builtin_amdgcn_readfirstlane(sg[0]);
        nrows = __builtin_amdgcn_readfirstlane(sg[1]); shift = __builtin_amdgcn_readfirstlane(sg[2]);
        nblk = (nrows + BVEC - 1) / BVEC;
        lp = a.codes + (a.list_base[ls] / BVEC) * (int64_t)(BVEC * M);
    }
    // ---- scan: wave wq of the slot takes blocks wq, wq + 4, ...; PD blocks in flight per wave (register slots refilled in place
    // right after their codes have become gather addresses, as in k_pq_scan_rot; the prologue issues in slot order so that one
    // s_waitcnt serves the loop entry and the back edge)
    {
        constexpr int PD = SL ? 2 : 4;
        constexpr int NFx = SL ? SL : NF > 0 ? NF : 1;
        v4u ca[PD][NFx]; v2u cb[PD];
        auto fetch = [&](int b, v4u (&xa)[NFx], v2u& xb) {
            const uint8_t* bp = lp + (SL ? pq_sliced_off(b, 0, M) : (int64_t)b * (BVEC * M));      // SL: the list starts on a group boundary
#pragma unroll
            for (int p = 0; p < (SL ? SL : NF); p++) xa[p] = *reinterpret_cast<const v4u*>(bp + p * (SL ? PQ_SLICED_GB * 1024 : 1024) + lane * 16);
            if (NH && !SL) xb = *reinterpret_cast<const v2u*>(bp + NF * 1024 + lane * 8);
            __builtin_amdgcn_sched_barrier(0);
        };
        const int blast = nblk > 0 ? nblk - 1 : 0;
#pragma unroll
        for (int d = 0; d < PD; d++) cb[d] = v2u{0u, 0u};
        if (nblk > 0) {       // ONE conditional region: four separately guarded fetches would each count as "maybe not issued" in the waits
#pragma unroll
            for (int d = 0; d < PD; d++) fetch(wq + 4 * d < nblk ? wq + 4 * d : blast, ca[d], cb[d]);
        }
#pragma unroll 1
        for (int b0 = wq; b0 < nblk; b0 += 4 * PD) {
#pragma unroll
            for (int d = 0; d < PD; d++) {
                const int b = b0 + 4 * d;        // past the sample's end the slot holds the last block again: summed, not stored (a
                uint32_t gv[NG];                 // `break` here would make every slot's wait a vmcnt(0): the exit path has no younger loads)
                if constexpr (SL > 0) {
#pragma unroll
                    for (int sl = 0; sl < SL; sl++) {
                        const uint32_t cw[4] = {ca[d][sl].x, ca[d][sl].y, ca[d][sl].z, ca[d][sl].w};
                        const uint32_t orv = ((uint32_t)(sl >> 1) << 24) | ((sl & 1) ? 0x00808080u : 0u);        // plane sl >> 1, row half sl & 1
#pragma unroll
                        for (int s = 0; s < 16; s++)
                            gv[16 * sl + s] = __builtin_amdgcn_perm(cw[s >> 2], R1[s / 3] | orv, 0x0c030000u | ((uint32_t)(4 + (s & 3)) << 8) | (uint32_t)(s % 3));
                    }
                } else if (NF >= 1) {
                    const uint32_t cw[4] = {ca[d][0].x, ca[d][0].y, ca[d][0].z, ca[d][0].w};
#pragma unroll
                    for (int s = 0; s < 16; s++)
                        gv[s] = __builtin_amdgcn_perm(cw[s >> 2], R0[s >> 2], 0x0c0c0000u | ((uint32_t)(4 + (s & 3)) << 8) | (uint32_t)(s & 3));
                }
                if (NF >= 2 && !SL) {
                    const uint32_t cw[4] = {ca[d][NFx - 1].x, ca[d][NFx - 1].y, ca[d][NFx - 1].z, ca[d][NFx - 1].w};
#pragma unroll
                    for (int s = 0; s < 16; s++)
                        gv[16 + s] = __builtin_amdgcn_perm(cw[s >> 2], R1[s / 3], 0x0c030000u | ((uint32_t)(4 + (s & 3)) << 8) | (uint32_t)(s % 3));
                }
                if (NH && !SL) {
                    const uint32_t cw[2] = {cb[d].x, cb[d].y};
#pragma unroll
                    for (int s = 0; s < 8; s++) {
                        if (NF == 0)
                            gv[s] = __builtin_amdgcn_perm(cw[s >> 2], R0[s >> 2], 0x0c0c0000u | ((uint32_t)(4 + (s & 3)) << 8) | (uint32_t)(s & 3));
                        else
                            gv[16 * NF + s] = __builtin_amdgcn_perm(cw[s >> 2], R1[s / 3], 0x0c030000u | ((uint32_t)(4 + (s & 3)) << 8) | (uint32_t)(s % 3));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                fetch(b + 4 * PD < nblk ? b + 4 * PD : blast, ca[d], cb[d]);      // the slot's codes are dead: refill in place
#pragma unroll
                for (int s = 0; s < NG; s++) gv[s] = lds_rd32(gv[s]);
                v4i C = {0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < NG / 4; t++) {
                    const v4i Av = {(int)gv[4 * t], (int)gv[4 * t + 1], (int)gv[4 * t + 2], (int)gv[4 * t + 3]};
                    C = __builtin_amdgcn_mfma_i32_16x16x64_i8(Av, Bm, C, 0, 0, 0);
                }
                // lanes (g, n): C[r] = sum over m of (u8 - 128) for vector 4 g + r of the block and query n; my slot's query is n == grp
                // (SL: vector 16 (n >> 2) + 4 g + r and query n & 3, n < 8)
                if ((SL ? (n < 8 && (n & 3) == grp) : n == grp) && b < nblk) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int pos = SL ? b * 32 + 16 * (n >> 2) + 4 * g + r : b * 16 + 4 * g + r;
                        if (BIG) {
                            const int v = C[r] + 128 * M + 1 + shift;          // shifted sum + 1; < 1: below every score the bound could certify
                            if (pos < nrows && v >= 1) atomicAdd(&hq[(v >> PP4_SH) >> 1], 1u << (16 * ((v >> PP4_SH) & 1)));
                        } else {
                            mysum[pos] = pos < len ? (uint16_t)(C[r] + 128 * M + 1) : (uint16_t)0;
                        }
                    }
                }
            }
        }
    }
    }       // segments
    PP4_MARK(3);
    __syncthreads();
    PP4_MARK(4);
    // ---- k-th largest (sum + 1) of each slot's sample: two 8-bit radix rounds over the 16-bit values, the four slots side by side
    // with the same barriers.  ctl[slot]: [0] high byte, [1] remaining rank, [2] low byte / -1 = fewer than k vectors
    const int tg = (wq << 6) | lane;                     // thread within the slot
    int32_t* hs = hist + grp * 256;
    int32_t* cs = ctl + grp * 8;
    const int N = nblk * BVEC;
    auto find_bin = [&](int want) -> int {              // wave 0 of the slot: the bin holding the want-th largest; leaves the rank inside it in cs[1]
        const int h0 = hs[4 * lane], h1 = hs[4 * lane + 1], h2 = hs[4 * lane + 2], h3 = hs[4 * lane + 3];
        const int sum4 = h0 + h1 + h2 + h3;
        int suf = sum4;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_down(suf, off); if (lane + off < 64) suf += y; }
        const uint64_t mk = __builtin_amdgcn_ballot_w64(suf >= want);
        int res = -1;
        if (mk != 0ull) {
            const int L = 63 - __builtin_clzll((unsigned long long)mk);
            const int runL = __shfl(suf - sum4, L);
            const int a0 = __shfl(h0, L), a1 = __shfl(h1, L), a2 = __shfl(h2, L), a3 = __shfl(h3, L);
            const int hb[4] = {a0, a1, a2, a3};
            int run = runL;
            for (int bb = 3; bb >= 0; bb--) {
                if (run + hb[bb] >= want) { res = 4 * L + bb; if (lane == 0) cs[1] = want - run; break; }
                run += hb[bb];
            }
        }
        return res;
    };
    int kth = -1;                           // (k-th largest integer sum) + 1 — BIG: a lower bound of it; -1: fewer than k vectors in the sample
    if (BIG) {
        // the highest bin B whose suffix count reaches k: wave 0 of the slot, lane i owns bins [i per, (i + 1) per)
        if (wq == 0) {
            constexpr int per = (NB + 63) / 64;
            auto cnt_of = [&](int bin) -> int { return bin < NB ? (int)((hq[bin >> 1] >> (16 * (bin & 1))) & 0xffffu) : 0; };
            int mine = 0;
            for (int t = 0; t < per; t++) mine += cnt_of(lane * per + t);
            int suf = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_down(suf, off); if (lane + off < 64) suf += y; }
            const uint64_t mk = __builtin_amdgcn_ballot_w64(suf >= a.k);
            int B = -1;
            if (mk != 0ull) {
                const int L = 63 - __builtin_clzll((unsigned long long)mk);
                int run = __shfl(suf - mine, L);
                if (lane == L)
                    for (int t = per - 1; t >= 0; t--) { run += cnt_of(L * per + t); if (run >= a.k) { B = L * per + t; break; } }
                B = __shfl(B, L);
            }
            if (lane == 0) cs[0] = B;
        }
        __syncthreads();
        const int B = cs[0];
        if (B >= 0) { kth = B << PP4_SH; if (kth < 1) kth = 1; }       // the bin's lower edge (values are >= 1)
    } else {
        for (int t = tg; t < N; t += 256) { const int v = mysum[t]; if (v) atomicAdd(&hs[v >> 8], 1); }
        __syncthreads();
        if (wq == 0) { const int d = find_bin(a.k); if (lane == 0) cs[0] = d; }
        __syncthreads();
        const int dhi = cs[0], want2 = cs[1];
        __syncthreads();
        for (int t = tg; t < 256; t += 256) hs[t] = 0;
        __syncthreads();
        if (dhi >= 0) for (int t = tg; t < N; t += 256) { const int v = mysum[t]; if (v && (v >> 8) == dhi) atomicAdd(&hs[v & 255], 1); }
        __syncthreads();
        if (wq == 0) { const int e = dhi >= 0 ? find_bin(want2) : -1; if (lane == 0) cs[2] = e; }
        __syncthreads();
        const int dlo = cs[2];
        if (dhi >= 0 && dlo >= 0) kth = (dhi << 8) | dlo;
    }
    PP4_MARK(5);
    // ---- the threshold (k_pq_prepass's arithmetic), the empty merge state, the candidate counter
    if (grp < np) {
        uint64_t* o = a.state + q * a.KP;
        for (int t = tg; t < a.KP; t += 256) o[t] = 0ull;
        if (tg == 0) {
            uint64_t tau = 0ull;
            if (kth >= 0 && l >= 0) {
                const float scale = a.qparam[q * 4 + 0], bias = a.qparam[q * 4 + 1], eps = a.qparam[q * 4 + 2];
                const float dis0 = a.probe_dis0[q * a.nprobe + j0];
                const float a_k = dis0 + __fmaf_rn(scale, (float)(kth - 1), bias);
                float t = __fmaf_rn(-2.0002f, eps, a_k);
                t -= fabsf(t) * 4.8e-7f + 1e-30f;
                tau = make_key(t, 0xFFFFFFFFu);
            }
            a.tau[q] = tau;
            a.cand_cnt[q * CCS] = 0ull;
            if (a.excl) a.excl[q] = (uint16_t)0;
        }
    }
    PP4_MARK(6);
}

template <int NF, int NH, int SL = 0>
static void launch_pq_prepass4_t(const PQPrepassArgs& a, int64_t nq, hipStream_t st) {
    const size_t shm = (size_t)(SL ? (SL + 1) / 2 : NF + NH) * 65536 + (size_t)4 * a.pre_rows * 2 + (1024 + 32) * 4 + 64;
    static DevSize attr;
    attr.grow(shm, [&] { (void)hipFuncSetAttribute((const void*)k_pq_prepass4<NF, NH, false, SL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); });
    hipLaunchKernelGGL((k_pq_prepass4<NF, NH, false, SL>), dim3((unsigned)((nq + 3) / 4)), dim3(1024), shm, st, a, nq);
}
template <int NF, int NH, int SL = 0>
static void launch_pq_prepass4_big_t(const PQPrepassArgs& a, int64_t nq, hipStream_t st) {
    constexpr int M = SL ? 32 * SL : 64 * NF + 32 * NH;
    constexpr int NBW = ((((255 * M + 1) >> (M > 96 ? 4 : 3)) + 1) + 1) / 2;
    const size_t shm = (size_t)(SL ? (SL + 1) / 2 : NF + NH) * 65536 + (size_t)(4 * NBW + 32 + 4 * PP4_MAXSEG * 4 + 4) * 4 + 64;
    static DevOnce once;
    once.once([&] { (void)hipFuncSetAttribute((const void*)k_pq_prepass4<NF, NH, true, SL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); });
    hipLaunchKernelGGL((k_pq_prepass4<NF, NH, true, SL>), dim3((unsigned)((nq + 3) / 4)), dim3(1024), shm, st, a, nq);
}
// the BIG form (histogram of the sums, samples of any size over several lists): 0 on launch, -1 when it does not apply
int launch_pq_prepass4_big(const PQPrepassArgs& a, int64_t nq, hipStream_t st) {
    if (nq <= 0) return 0;
    if (!pq_rot_family(a.CB) || a.pre_rows <= 0 || a.pre_rows > 32768) return -1;        // 16-bit counters
    if (a.CB == PQ_SLICED) { if (a.Mpad != 96) return -1; launch_pq_prepass4_big_t<0, 0, 3>(a, nq, st); return 0; }
    switch (a.Mpad) {
        case 32: launch_pq_prepass4_big_t<0, 1>(a, nq, st); return 0;
        case 64: launch_pq_prepass4_big_t<1, 0>(a, nq, st); return 0;
        case 96: launch_pq_prepass4_big_t<1, 1>(a, nq, st); return 0;
        case 128: launch_pq_prepass4_big_t<2, 0>(a, nq, st); return 0;
        default: return -1;
    }
}
// sample rows the 4-query form can hold beside its tables (0: this M has no such kernel)
int pq_prepass4_max_rows(int M) {
    if (!pq_rot_applies(M) || M < 32) return 0;
    const int nph = (M >> 6) + ((M >> 5) & 1);
    const int64_t room = (int64_t)160 * 1024 - (int64_t)nph * 65536 - (1024 + 32) * 4 - 64 - 256;
    return room < 8 * 64 ? 0 : (int)(room / 8 / 64 * 64);
}
int launch_pq_prepass4(const PQPrepassArgs& a, int64_t nq, hipStream_t st) {
    if (nq <= 0) return 0;
    if (!pq_rot_family(a.CB) || a.pre_rows <= 0 || a.pre_rows % 64 || a.pre_rows > pq_prepass4_max_rows(a.Mpad)) return -1;
    if (a.CB == PQ_SLICED) { if (a.Mpad != 96) return -1; launch_pq_prepass4_t<0, 0, 3>(a, nq, st); return 0; }
    switch (a.Mpad) {
        case 32: launch_pq_prepass4_t<0, 1>(a, nq, st); return 0;
        case 64: launch_pq_prepass4_t<1, 0>(a, nq, st); return 0;
        case 96: launch_pq_prepass4_t<1, 1>(a, nq, st); return 0;
        case 128: launch_pq_prepass4_t<2, 0>(a, nq, st); return 0;
        default: return -1;
    }
}

// ---------------------------------------------------------------------------------------
// Exact scan of the rotated layout: workgroup = (query, probed list, chunk of slabs), the query's fp32 table in LDS
// (M KiB), one thread per vector, code bytes through pq_code_addr, sums in m order (= k_pq_scan = the oracle).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_pq_scan_rot_exact(PQScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float rot_lut_s[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
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
    {
        const float4* src = reinterpret_cast<const float4*>(a.lut + q * a.Mpad * 256);
        float4* dst = reinterpret_cast<float4*>(rot_lut_s);
        for (int e = tid; e < a.Mpad * 64; e += 1024) dst[e] = src[e];
    }
    __syncthreads();
    const int64_t row0 = a.list_base[l];
    for (int64_t s = s0 + w; s < s1; s += 16) {
        const int64_t pos = s * 64 + lane;
        float sum = 0.0f;
        for (int m0 = 0; m0 < a.M; m0 += 8) {
            uint32_t code[8];
#pragma unroll
            for (int t = 0; t < 8; t++) code[t] = a.codes[pq_code_addr(row0 + pos, m0 + t, a.Mpad, a.CB)];
#pragma unroll
            for (int t = 0; t < 8; t++) sum += rot_lut_s[(m0 + t) * 256 + code[t]];
        }
        out[pos] = (pos < len) ? dis0 + sum : -__builtin_inff();
    }
}

// ---------------------------------------------------------------------------------------
// IVF-PQ with METRIC_L2 (round 6; the reference only ever builds METRIC_INNER_PRODUCT, src/indicies/ivf_pq.py:147-153 — this is the
// other metric `north_star` names): squared distance to the decoded vector, ||(q - c_l) - r^||^2 = sum_m ||(q - c_l)_m - cb[m][code_m]||^2.
// The table depends on the (query, list) PAIR, so there is no shared-table fast scan: workgroup = (query, probed list, chunk of slabs)
// builds the pair's fp32 table in LDS (M x 256 fmaf chains of dsub terms: the arithmetic the CPU restatement under oracle/ fixes for this metric) and scans
// its chunk with one thread per vector, sums in m order.  Scores leave NEGATED (every selection downstream keeps the largest keys;
// k_finalize hands back the distance).  Any code layout (byte loads through pq_code_addr): correct first, not tuned — 2 x 256 x M x dsub
// flops of table build per (query, list, chunk) bound it, not the scan.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_pq_scan_l2(PQScanArgs a, const float* __restrict__ Q32, int ldq, const float* __restrict__ centroids,
                                                     const float* __restrict__ codebooks, int d, int dsub) {
    extern __shared__ __attribute__((aligned(16))) float l2_tab[];          // [Mpad][256], then the residual query [d]
    float* qr = l2_tab + (size_t)a.Mpad * 256;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
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
    float* out = a.temp + q * a.tstride + a.seg_start[q * (a.nprobe + 1) + j];
    for (int t = tid; t < d; t += 1024) qr[t] = __fsub_rn(Q32[q * ldq + t], centroids[(int64_t)l * d + t]);
    __syncthreads();
    for (int e = tid; e < a.Mpad * 256; e += 1024) {
        const int m = e >> 8, c = e & 255;
        float acc = 0.0f;
        if (m < a.M) {
            const float* cw = codebooks + ((int64_t)m * 256 + c) * dsub;
            for (int t = 0; t < dsub; t++) { const float df = __fsub_rn(qr[m * dsub + t], cw[t]); acc = __fmaf_rn(df, df, acc); }
        }
        l2_tab[e] = acc;
    }
    __syncthreads();
    const int64_t row0 = a.list_base[l];
    for (int64_t s = s0 + w; s < s1; s += 16) {
        const int64_t pos = s * 64 + lane;
        float sum = 0.0f;
        for (int m0 = 0; m0 < a.M; m0 += 4) {
            uint32_t code[4];
#pragma unroll
            for (int t = 0; t < 4; t++) code[t] = (m0 + t < a.M) ? a.codes[pq_code_addr(row0 + pos, m0 + t, a.Mpad, a.CB)] : 0u;
#pragma unroll
            for (int t = 0; t < 4; t++) if (m0 + t < a.M) sum += l2_tab[(m0 + t) * 256 + code[t]];
        }
        out[pos] = (pos < len) ? -sum : -__builtin_inff();
    }
}

int launch_pq_scan_l2(const PQScanArgs& a, const float* Q32, int ldq, const float* centroids, const float* codebooks, int d, int dsub, hipStream_t st) {
    const int64_t pairs = a.nq * a.nprobe;
    if (pairs <= 0 || a.max_chunks <= 0) return 0;
    const size_t shm = (size_t)a.Mpad * 1024 + (size_t)d * 4;
    if (shm > (size_t)160 * 1024) return -1;
    static DevSize attr;
    bool attr_ok = true;
    attr.grow(shm, [&] { attr_ok = hipFuncSetAttribute((const void*)k_pq_scan_l2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess; });
    if (!attr_ok) return -1;
    hipLaunchKernelGGL(k_pq_scan_l2, dim3((unsigned)pairs, (unsigned)a.max_chunks), dim3(1024), shm, st, a, Q32, ldq, centroids, codebooks, d, dsub);
    return 0;
}

int launch_pq_scan_rot_exact(const PQScanArgs& a, hipStream_t st) {
    const int64_t pairs = a.nq * a.nprobe;
    if (pairs <= 0 || a.max_chunks <= 0) return 0;
    if (!pq_rot_family(a.CB) || !pq_rot_applies(a.M)) return -1;
    const size_t shm = (size_t)a.Mpad * 1024;
    static DevSize attr;
    bool attr_ok = true;
    attr.grow(shm, [&] { attr_ok = hipFuncSetAttribute((const void*)k_pq_scan_rot_exact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess; });
    if (!attr_ok)
        return -1;
    hipLaunchKernelGGL(k_pq_scan_rot_exact, dim3((unsigned)pairs, (unsigned)a.max_chunks), dim3(1024), shm, st, a);
    return 0;
}

}  // namespace rsx
