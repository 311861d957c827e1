// k_pq_rot.hip — IVF-PQ scans of the ROTATED code layout (rsx_internal.h: CB = 0; M in {32, 64, 96, 128}).
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

constexpr int ROT_D = 4;      // 16-vector code blocks in flight per wave (16 M bytes each)
#ifndef ROT_CW
#define ROT_CW 4              // items (waves) per workgroup of k_pq_rot_compact
#endif

__device__ __forceinline__ uint32_t lds_rd32(uint32_t addr) {
    // raw LDS address: the kernels below declare no static LDS, so the dynamic segment starts at 0
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>((uintptr_t)addr);
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

template <int M, bool FILTER>
__global__ __launch_bounds__(256) void k_pq_rot_items(PQScan8Args A, PQRotItem* items, uint32_t* xcd_ctr) {
    const PQScanArgs& a = A.b;
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item < 8) xcd_ctr[item * 32] = 0u;     // the scan's per-XCD work counters (one per 128-byte line)
    const int ti = *A.total_items;
    if (item >= ti) return;
    int lo = 0, hi = A.nlist;   // largest l with item_off[l] <= item
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.item_off[mid] <= item) lo = mid; else hi = mid; }
    const int ng = A.group_off[lo + 1] - A.group_off[lo];
    const int r = item - A.item_off[lo];
    const int tile = r / ng, gi = r - tile * ng;
    const int cnt = A.pair_off[lo + 1] - A.pair_off[lo];
    const int pair0 = A.pair_off[lo] + 4 * gi;
    PQRotItem d;
    d.l = lo; d.tile = tile; d.np = (cnt - 4 * gi) > 4 ? 4 : (cnt - 4 * gi); d.pad0 = 0;
    d.len = a.list_len[lo]; d.base_row = a.list_base[lo];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int pi = A.pairs_sorted[pair0 + (k < d.np ? k : 0)];
        const int64_t q = pi / a.nprobe;
        const PQQParam p = A.qp[q];
        const int64_t col = a.seg_start[q * (a.nprobe + 1) + (pi - (int)q * a.nprobe)];
        const float dis0 = a.probe_dis0[pi];
        d.q[k] = (int32_t)q; d.dis0[k] = dis0; d.scale[k] = p.scale; d.bias[k] = p.bias;
        d.off[k] = FILTER ? col : q * a.tstride + col;
        const uint64_t tau = FILTER ? A.tau_key[q * A.tau_stride] : 0ull;
        d.tau[k] = tau;
        // threshold on the integer sum: the smallest S whose score dis0 + fma(scale, S, bias) (the expression the survivors are
        // scored with, monotone in S) reaches the threshold key's score.  The MFMA accumulates C = S - 128 M starting from
        // cinit = -(that threshold - 128 M): the block's result is then >= 0 exactly for the survivors.
        int thr = INT_MIN;
        if (k >= d.np) thr = INT_MAX;
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
    if (FILTER && A.prune && d.cinit[0] == -(1 << 30) && d.cinit[1] == -(1 << 30) && d.cinit[2] == -(1 << 30) && d.cinit[3] == -(1 << 30))
        d.l = -2;
    items[item] = d;
}

// ---------------------------------------------------------------------------------------
// The scan.  PERSISTENT workgroups (one per CU: the table owns the LDS): workgroup b serves XCD b % 8 and walks that XCD's
// contiguous range of the list-major item order with stride gridDim / 8, so the query groups of one list tile still run
// on one XCD close together in time (one HBM fetch per tile).  Per item: [barrier] stage the group's table, copy out the
// PREVIOUS item's survivor queues (their global reservations were issued before the staging and have landed by now),
// [barrier] scan.  Nothing at an item boundary waits on a dependent global load except the table rows themselves.
// ---------------------------------------------------------------------------------------
template <int NF, int NH, bool FILTER>
__global__ __launch_bounds__(1024) void k_pq_scan_rot(PQScan8Args A, const PQRotItem* __restrict__ items, uint64_t* __restrict__ seg_keys,
                                                      uint32_t* __restrict__ seg_cnt, uint32_t* xcd_ctr, int seg_cap, int bpw, int var) {
    constexpr int M = 64 * NF + 32 * NH;
    constexpr int NPH = NF + NH;               // phases = table planes
    constexpr int TAB = NPH * 65536;           // plane p at p * 64 KiB; row = code * 256; a half phase uses 128 B of the row
    constexpr int NG = M / 4;                  // gathers per lane per block
    constexpr int NR1 = NPH > 1 ? rot_nreg(1, NF >= 2 ? 16 : 8) : 0;
    constexpr int NR0 = NF >= 1 ? 4 : rot_nreg(0, 8);     // M = 32: the half phase IS plane 0
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) uint32_t rot_s[];
    uint8_t* sb = reinterpret_cast<uint8_t*>(rot_s);
    PQRotItem* islot = reinterpret_cast<PQRotItem*>(sb + TAB);                  // [2] current / next item record

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
    const int xlo = xcd * per_xcd;
    int xhi = xlo + per_xcd; if (xhi > ti) xhi = ti;
    if (xlo >= xhi) return;
    uint32_t* ctr = xcd_ctr + xcd * 32;
    unsigned drawn = 0;          // wave 0, lane 0: the counter value of the last draw (consumed one item later)
    // ---- per-lane constants, once per workgroup: rotation bytes, the one-hot B operand, the survivor-queue geometry
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
        uint32_t v = 0x01000000u;     // byte 3 = plane 1 -> address bit 16
#pragma unroll
        for (int bb = 0; bb < 3; bb++) {
            const int s = r * 3 + bb;
            const uint32_t rot = NF >= 2 ? (uint32_t)(64 * g + 4 * ((i + s) & 15))
                                         : (uint32_t)(64 * (g & 1) + 4 * ((i + s + 8 * (g >> 1)) & 15));
            v |= (rot & 255u) << (8 * bb);
        }
        R1[r] = v;
    }
    const int bsel = n < 4 ? (1 << (8 * n)) : 0;
    const v4i Bm = {bsel, bsel, bsel, bsel};
    const int vo16 = lane * 16, vo8 = lane * 8;
    // Survivors leave the scan with PLAIN stores: every (item, wave, query) owns a segment of seg_cap keys in HBM, the slot of
    // a survivor is the wave's running count for its query plus its rank among this step's survivors of the same query
    // (ballot + mbcnt), and the count goes out with one store per wave and item.  k_pq_rot_compact appends the segments to
    // the per-query candidate rows afterwards.  Nothing here returns a value: an LDS ds_add_rtn costs ~300 clk of the
    // whole CU's LDS pipe, and a returning global atomic sits in the wave's in-order vmcnt queue in front of the next item's
    // table loads (measured: 0.6 ms of a 3.8 ms scan for 1.9 M survivors).
    const uint64_t QM = n < 4 ? (0x0001000100010001ull << n) : 0ull;   // the four lanes that own query n

    int item = 0;
    if (w == 0) {
        if (lane == 0) { drawn = atomicAdd(ctr, 1u); }
        item = xlo + (int)__builtin_amdgcn_readfirstlane(drawn);
        if (lane == 0) drawn = atomicAdd(ctr, 1u);
        uint4 r0 = make_uint4(0xffffffffu, 0, 0, 0);       // l = -1: end marker
        if (lane < 11 && item < xhi) r0 = reinterpret_cast<const uint4*>(&items[item])[lane];
        if (lane < 11) reinterpret_cast<uint4*>(&islot[0])[lane] = r0;
        if (lane == 0) islot[0].pad0 = item;
    }
    int buf = 0;
#pragma unroll 1
    for (;; buf ^= 1) {
        __syncthreads();    // #1: every wave has left the previous item's scan (table free), the item record is in LDS
        const PQRotItem* it = &islot[buf];
        const int item_l = __builtin_amdgcn_readfirstlane(it->l);
        if (item_l == -1) break;
        item = __builtin_amdgcn_readfirstlane(it->pad0);
        const bool skip_item = item_l < -1;         // pruned: no survivors possible, only the bookkeeping below runs
        const int np = __builtin_amdgcn_readfirstlane(it->np);
        const int64_t len = ((int64_t)__builtin_amdgcn_readfirstlane((int)(it->len >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)it->len);
        const int64_t base_row = ((int64_t)__builtin_amdgcn_readfirstlane((int)(it->base_row >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)it->base_row);
        const int nblk = (int)(((len + 63) >> 6) << 2);     // 16-vector blocks of the list, slab padding included
        const int tb0 = __builtin_amdgcn_readfirstlane(it->tile) * (16 * bpw);
        const uint8_t* lp = a.codes + (base_row >> 4) * (int64_t)(16 * M);
        // ---- code loads: buffer instructions on a descriptor of THIS list's blocks (base + size in SGPRs, the block's byte
        // offset in an SGPR, lane * 16 in one constant VGPR): no address VALU, and a block past the list's end reads zeros
        // instead of needing a clamp (its sums are garbage that the pos < len test of the survivor path drops)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)lp, 0, nblk * 16 * M, 0x00020000);
        v4u ca[ROT_D][NF > 0 ? NF : 1]; v2u cb[ROT_D];
#pragma unroll
        for (int dd = 0; dd < ROT_D; dd++) {     // the first ROT_D blocks of this wave: in flight during the table staging
            const int so = (tb0 + w + 16 * dd) * (16 * M);
#pragma unroll
            for (int p = 0; p < NF; p++) ca[dd][p] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo16 + p * 1024, so, 0);
            if (NH) cb[dd] = __builtin_amdgcn_raw_buffer_load_b64(rs, vo8 + NF * 1024, so, 0);
        }

        // ---- stage the group's table: work unit = (code, 4 consecutive m) -> 4 dwords (one per m: byte k = query k, as int8 = u8 - 128)
        {
            const int64_t q0 = it->q[0], q1 = it->q[1], q2 = it->q[2], q3 = it->q[3];
            for (int e = tid; e < ((var & 4) || skip_item ? 0 : 256 * (M / 4)); e += 1024) {
                const int c = e / (M / 4), m4 = e - c * (M / 4);
                uint32_t in[4];
                in[0] = *reinterpret_cast<const uint32_t*>(A.lut8 + (q0 * 256 + c) * M + m4 * 4);
                in[1] = np > 1 ? *reinterpret_cast<const uint32_t*>(A.lut8 + (q1 * 256 + c) * M + m4 * 4) : 0u;
                in[2] = np > 2 ? *reinterpret_cast<const uint32_t*>(A.lut8 + (q2 * 256 + c) * M + m4 * 4) : 0u;
                in[3] = np > 3 ? *reinterpret_cast<const uint32_t*>(A.lut8 + (q3 * 256 + c) * M + m4 * 4) : 0u;
                const uint32_t t0 = __builtin_amdgcn_perm(in[1], in[0], 0x05010400u), t1 = __builtin_amdgcn_perm(in[1], in[0], 0x07030602u);
                const uint32_t u0 = __builtin_amdgcn_perm(in[3], in[2], 0x05010400u), u1 = __builtin_amdgcn_perm(in[3], in[2], 0x07030602u);
                uint4 o;
                o.x = __builtin_amdgcn_perm(u0, t0, 0x05040100u) ^ 0x80808080u;
                o.y = __builtin_amdgcn_perm(u0, t0, 0x07060302u) ^ 0x80808080u;
                o.z = __builtin_amdgcn_perm(u1, t1, 0x05040100u) ^ 0x80808080u;
                o.w = __builtin_amdgcn_perm(u1, t1, 0x07060302u) ^ 0x80808080u;
                const int m = m4 * 4;
                const int p = m < 64 * NF ? (m >> 6) : NF;
                const int slot = m < 64 * NF ? (m & 63) : (m - 64 * NF);
                *reinterpret_cast<uint4*>(sb + p * 65536 + c * 256 + slot * 4) = o;
            }
        }
        uint64_t* myseg = seg_keys + (((size_t)item * 16 + w) * 4 + nq4) * seg_cap;
        // ---- next item's record: requested now by wave 0 (its index was drawn during the previous item: the counter's answer
        // has had a barrier wait and a table staging to arrive), parked in LDS after the scan
        uint4 pre = make_uint4(0xffffffffu, 0, 0, 0);
        int i1 = 0;
        if (w == 0) {
            i1 = xlo + (int)__builtin_amdgcn_readfirstlane(drawn);
            if (lane < 11 && i1 < xhi) pre = reinterpret_cast<const uint4*>(&items[i1])[lane];
        }
        // ---- the lane's share of the item record: accumulator init, score parameters of the query it owns (n < 4)
        const int cinit = FILTER ? (n < 4 ? it->cinit[nq4] : -(1 << 30)) : 0;
        const v4i Ci = {cinit, cinit, cinit, cinit};
        const float p_dis0 = it->dis0[nq4], p_scale = it->scale[nq4], p_bias = it->bias[nq4];
        const int64_t p_off = it->off[nq4];
        const uint64_t p_tau = it->tau[nq4];
        uint32_t qcnt = 0;                    // survivors of the lane's query so far (equal in the four lanes of a query)
        __syncthreads();    // #2: table staged

        // ---- scan: blocks tb0 + w + 16 j, j < bpw, ROT_D of them in flight
#pragma unroll 1
        for (int j0 = 0; j0 < ((var & 2) || skip_item ? 0 : bpw); j0 += ROT_D) {
            if (tb0 + w + 16 * j0 >= nblk) break;
#pragma unroll
            for (int dd = 0; dd < ROT_D; dd++) {
                const int b = tb0 + w + 16 * (j0 + dd);
                uint32_t gv[NG];
                // addresses: (plane << 16) | (code << 8) | rotation byte — one v_perm each
                if (NF >= 1) {
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
                    const int so = (b + 16 * ROT_D) * (16 * M);
#pragma unroll
                    for (int p = 0; p < NF; p++) ca[dd][p] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo16 + p * 1024, so, 0);
                    if (NH) cb[dd] = __builtin_amdgcn_raw_buffer_load_b64(rs, vo8 + NF * 1024, so, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < NG; s++) gv[s] = lds_rd32(gv[s]);
                v4i C = Ci;
#pragma unroll
                for (int t = 0; t < NG / 4; t++) {
                    const v4i Av = {(int)gv[4 * t], (int)gv[4 * t + 1], (int)gv[4 * t + 2], (int)gv[4 * t + 3]};
                    C = __builtin_amdgcn_mfma_i32_16x16x64_i8(Av, Bm, C, 0, 0, 0);
                }
                // C[r] (lanes n < 4) = cinit + sum over m of (u8 - 128) for vector 4 g + r of the block and query n
                if (FILTER) {
                    if (__builtin_amdgcn_ballot_w64((C[0] & C[1] & C[2] & C[3]) >= 0) && !(var & 1)) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const bool cnd = C[r] >= 0;
                            if (__builtin_amdgcn_ballot_w64(cnd)) {
                                const int64_t pos = ((int64_t)b << 4) + 4 * g + r;
                                const float sc = p_dis0 + __fmaf_rn(p_scale, (float)(C[r] - cinit + 128 * M), p_bias);
                                const uint64_t key = (cnd && pos < len) ? make_key(sc, (uint32_t)p_off + (uint32_t)pos) : 0ull;
                                const bool pass = key > p_tau;
                                const uint64_t mq = __builtin_amdgcn_ballot_w64(pass) & QM;      // this step's survivors of MY query
                                if (pass) {
                                    const uint32_t slot = qcnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
                                    if (slot < (uint32_t)seg_cap) myseg[slot] = key;   // beyond: counted, dropped -> the query is re-run exactly
                                }
                                qcnt += (uint32_t)__builtin_popcountll(mq);
                            }
                        }
                    }
                } else {
                    if (n < np && b < nblk) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int64_t pos = ((int64_t)b << 4) + 4 * g + r;
                            const float sc = p_dis0 + __fmaf_rn(p_scale, (float)(C[r] + 128 * M), p_bias);
                            a.temp[p_off + pos] = (pos < len) ? sc : -__builtin_inff();
                        }
                    }
                }
            }
        }
        // ---- item epilogue: the wave's four survivor counts leave with one store (lanes 0..3 own queries 0..3); wave 0 parks
        // the next record (or the end marker) and draws the index of the item after it
        if (FILTER && lane < 4) seg_cnt[((size_t)item * 16 + w) * 4 + lane] = qcnt;
        if (w == 0) {
            if (lane < 11) reinterpret_cast<uint4*>(&islot[buf ^ 1])[lane] = pre;
            if (lane == 0) { islot[buf ^ 1].pad0 = i1; drawn = atomicAdd(ctr, 1u); }
        }
    }
}

// One wave per work item: append the item's (wave, query) survivor segments to the candidate rows of its queries — the only
// atomics of the filtered scan live here, one reservation per (item, query), in a kernel with thousands of independent waves.
__global__ __launch_bounds__(64 * ROT_CW) void k_pq_rot_compact(const PQRotItem* __restrict__ items, const int32_t* total_items,
                                                       const uint64_t* __restrict__ seg_keys, const uint32_t* __restrict__ seg_cnt,
                                                       int seg_cap, uint64_t* cand, unsigned long long* cand_cnt, int cand_cap) {
    const int item = blockIdx.x * ROT_CW + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (item >= *total_items) return;
    const int w = lane >> 2, k = lane & 3;
    const uint32_t c0 = seg_cnt[(size_t)item * 64 + lane];                 // lane = (wave, query) segment
    if (__builtin_amdgcn_ballot_w64(c0 != 0u) == 0ull) return;
    const uint32_t c = c0 < (uint32_t)seg_cap ? c0 : (uint32_t)seg_cap;
    // exclusive prefix over the 16 waves of the same query (lanes k, k + 4, ...), and the query's total
    uint32_t incl = c;
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) { const uint32_t y = __shfl_up(incl, off); if (lane >= off) incl += y; }
    const uint32_t total = __shfl(incl, 60 + k);
    const uint64_t ovm = __builtin_amdgcn_ballot_w64(c0 > (uint32_t)seg_cap);     // segments that dropped keys (all 64 lanes vote)
    const bool myover = (ovm & (0x1111111111111111ull << k)) != 0ull;              // ... any of them of MY query
    const int64_t q = items[item].q[k];
    unsigned long long base = 0;
    if (w == 0 && (total > 0 || myover)) {
        // an overflowing segment dropped keys: push the row's count past its capacity so that k_finalize flags the query
        base = atomicAdd(&cand_cnt[q * CCS], (unsigned long long)total + (myover ? (unsigned long long)cand_cap + 1ull : 0ull));
    }
    base = __shfl(base, k) + (incl - c);
    // copy-out: the wave walks the (at most 64) non-empty segments, all lanes on one segment at a time (coalesced 8-byte moves)
    uint64_t live = __builtin_amdgcn_ballot_w64(c != 0u);
    while (live) {
        const int sgm = __builtin_ctzll(live);
        live &= live - 1;
        const uint32_t cs = __builtin_amdgcn_readlane(c, sgm);
        const unsigned long long bs = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(base >> 32), sgm) << 32) |
                                      (uint32_t)__builtin_amdgcn_readlane((int)base, sgm);
        const int64_t qs = items[item].q[sgm & 3];
        const uint64_t* src = seg_keys + ((size_t)item * 64 + sgm) * seg_cap;
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

template <int NF, int NH, bool FILTER>
static int launch_pq_scan_rot_t(const PQScan8Args& A, int bpw, void* desc_ws, int seg_cap, hipStream_t st) {
    constexpr int M = 64 * NF + 32 * NH;
    const size_t shm = (size_t)(NF + NH) * 65536 + 2 * 192;
    static DevOnce once;
    static int ncu_of[64] = {};
    int& ncu = ncu_of[cur_device()];
    static std::atomic<int> failed{0};
    once.once([&] {
        int dev = 0; hipDeviceProp_t pr;
        if (hipFuncSetAttribute((const void*)k_pq_scan_rot<NF, NH, FILTER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) { failed = 1; return; }
        ncu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    });
    if (failed) return -1;
    if (ncu <= 0) ncu = 256;
    PQRotItem* items = reinterpret_cast<PQRotItem*>(desc_ws);
    uint32_t* seg_cnt = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(desc_ws) + (size_t)(A.max_items + 8) * 176);
    uint64_t* seg_keys = reinterpret_cast<uint64_t*>(reinterpret_cast<uint8_t*>(seg_cnt) + (size_t)(A.max_items + 8) * 256);
    uint32_t* xcd_ctr = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(seg_keys) + (size_t)(A.max_items + 8) * 64 * seg_cap * 8);
    hipLaunchKernelGGL((k_pq_rot_items<M, FILTER>), dim3((unsigned)((A.max_items + 255) / 256)), dim3(256), 0, st, A, items, xcd_ctr);
    static int var = -1;
    if (var < 0) { const char* e = getenv("RSX_ROT_VARIANT"); var = e ? atoi(e) : 0; }
    // one persistent workgroup per CU (a multiple of 8: workgroup b serves XCD b % 8); never more than the work items
    int64_t grid = (ncu + 7) & ~7;
    if (grid > ((A.max_items + 7) & ~7)) grid = (A.max_items + 7) & ~7;
    hipLaunchKernelGGL((k_pq_scan_rot<NF, NH, FILTER>), dim3((unsigned)grid), dim3(1024), shm, st, A, items, seg_keys, seg_cnt, xcd_ctr,
                       seg_cap, bpw, var);
    if (FILTER)
        hipLaunchKernelGGL(k_pq_rot_compact, dim3((unsigned)((A.max_items + ROT_CW - 1) / ROT_CW)), dim3(64 * ROT_CW), 0, st, items, A.total_items, seg_keys, seg_cnt, seg_cap,
                           A.cand, A.cand_cnt, A.cand_cap);
    return 0;
}

// returns 0 on launch, -1 if this M has no rotated kernel.  tau_key == null: unfiltered (every score to a.temp).
int launch_pq_scan_rot(const PQScanArgs& a, const uint8_t* lut8t, const void* qparam, const int32_t* pairs_sorted,
                       const int32_t* pair_off, const int32_t* group_off, const int32_t* total_groups,
                       const int32_t* item_off, const int32_t* total_items, int nlist, int64_t max_items, int vpl,
                       const uint64_t* tau_key, int64_t tau_stride, uint64_t* cand, unsigned long long* cand_cnt,
                       int cand_cap, void* item_ws, int seg_cap, int prune, hipStream_t st) {
    if (a.CB != 0 || !item_ws || !pq_rot_applies(a.M) || a.M != a.Mpad || max_items <= 0 || max_items > 0x7fffff00) return -1;
    PQScan8Args A;
    A.b = a; A.lut8 = lut8t; A.qp = (const PQQParam*)qparam;
    A.pairs_sorted = pairs_sorted; A.pair_off = pair_off; A.group_off = group_off; A.total_groups = total_groups;
    A.item_off = item_off; A.total_items = total_items;
    A.nlist = nlist; A.max_items = (int)max_items;
    A.tau_key = tau_key; A.tau_stride = tau_stride; A.cand = cand; A.cand_cnt = cand_cnt; A.cand_cap = cand_cap;
    A.prune = prune;
    const int bpw = 4 * vpl;   // tile = 16 waves x bpw blocks x 16 vectors = 1024 vpl vectors, as k_pq_scan8's
    const bool f = tau_key != nullptr;
    switch (a.M) {
        case 32: return f ? launch_pq_scan_rot_t<0, 1, true>(A, bpw, item_ws, seg_cap, st) : launch_pq_scan_rot_t<0, 1, false>(A, bpw, item_ws, seg_cap, st);
        case 64: return f ? launch_pq_scan_rot_t<1, 0, true>(A, bpw, item_ws, seg_cap, st) : launch_pq_scan_rot_t<1, 0, false>(A, bpw, item_ws, seg_cap, st);
        case 96: return f ? launch_pq_scan_rot_t<1, 1, true>(A, bpw, item_ws, seg_cap, st) : launch_pq_scan_rot_t<1, 1, false>(A, bpw, item_ws, seg_cap, st);
        case 128: return f ? launch_pq_scan_rot_t<2, 0, true>(A, bpw, item_ws, seg_cap, st) : launch_pq_scan_rot_t<2, 0, false>(A, bpw, item_ws, seg_cap, st);
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
            for (int t = 0; t < 8; t++) code[t] = a.codes[pq_code_addr(row0 + pos, m0 + t, a.Mpad, 0)];
#pragma unroll
            for (int t = 0; t < 8; t++) sum += rot_lut_s[(m0 + t) * 256 + code[t]];
        }
        out[pos] = (pos < len) ? dis0 + sum : -__builtin_inff();
    }
}

int launch_pq_scan_rot_exact(const PQScanArgs& a, hipStream_t st) {
    const int64_t pairs = a.nq * a.nprobe;
    if (pairs <= 0 || a.max_chunks <= 0) return 0;
    if (a.CB != 0 || !pq_rot_applies(a.M)) return -1;
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
