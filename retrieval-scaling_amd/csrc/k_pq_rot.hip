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

constexpr int ROT_LQ = 62;    // keys per (wave, query) queued in LDS before the wave's one global reservation per query
constexpr int ROT_D = 4;      // 16-vector code blocks in flight per wave (16 M bytes each)

__device__ __forceinline__ uint32_t lds_rd32(uint32_t addr) {
    // raw LDS address: the kernels below declare no static LDS, so the dynamic segment starts at 0
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>((uintptr_t)addr);
}

// registers of rotation bytes a phase needs: plane 0 packs 4 per dword, planes >= 1 pack 3 + the plane byte
__host__ __device__ constexpr int rot_nreg(int plane, int steps) { return plane == 0 ? steps / 4 : (steps + 2) / 3; }

__global__ __launch_bounds__(256) void k_pq_item_desc(const int32_t* item_off, const int32_t* group_off, const int32_t* pair_off,
                                                      const int32_t* total_items, const int64_t* list_base, const int64_t* list_len,
                                                      int nlist, int G, int64_t grid, PQItemDesc* desc) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= grid) return;
    PQItemDesc d; d.l = -1; d.tile = 0; d.pair0 = 0; d.np = 0; d.len = 0; d.base_row = 0;
    const int ti = *total_items;
    const int per_xcd = (ti + 7) >> 3;
    const int ix = (int)(b >> 3);
    const int item = (int)(b & 7) * per_xcd + ix;       // the contiguous item range of XCD b % 8 (see pq_decode_item)
    if (ix < per_xcd && item < ti) {
        int lo = 0, hi = nlist;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (item_off[mid] <= item) lo = mid; else hi = mid; }
        const int ng = group_off[lo + 1] - group_off[lo];
        const int r = item - item_off[lo];
        const int tile = r / ng, gi = r - tile * ng;
        const int cnt = pair_off[lo + 1] - pair_off[lo];
        d.l = lo; d.tile = tile; d.pair0 = pair_off[lo] + G * gi;
        d.np = (cnt - G * gi) > G ? G : (cnt - G * gi);
        d.len = list_len[lo]; d.base_row = list_base[lo];
    }
    desc[b] = d;
}
void launch_pq_item_desc(const int32_t* item_off, const int32_t* group_off, const int32_t* pair_off, const int32_t* total_items,
                         const int64_t* list_base, const int64_t* list_len, int nlist, int group_size, int64_t grid,
                         PQItemDesc* desc, hipStream_t st) {
    if (grid <= 0) return;
    hipLaunchKernelGGL(k_pq_item_desc, dim3((unsigned)((grid + 255) / 256)), dim3(256), 0, st, item_off, group_off, pair_off,
                       total_items, list_base, list_len, nlist, group_size, grid, desc);
}

template <int NF, int NH, bool FILTER>
__global__ __launch_bounds__(1024) void k_pq_scan_rot(PQScan8Args A, int bpw, int var) {
    constexpr int M = 64 * NF + 32 * NH;
    constexpr int NPH = NF + NH;               // phases = table planes
    constexpr int TAB = NPH * 65536;           // plane p at p * 64 KiB; row = code * 256; a half phase uses 128 B of the row
    constexpr int NG = M / 4;                  // gathers per lane per block
    constexpr int NRF = 4;                                  // plane-0 full phase
    constexpr int NR1 = NPH > 1 ? rot_nreg(1, NF >= 2 ? 16 : 8) : 0;
    constexpr int NR0 = NF >= 1 ? NRF : rot_nreg(0, 8);     // M = 32: the half phase IS plane 0
    extern __shared__ __attribute__((aligned(16))) uint32_t rot_s[];
    uint8_t* sb = reinterpret_cast<uint8_t*>(rot_s);
    float* prm_f = reinterpret_cast<float*>(sb + TAB);              // [4][4]: dis0, scale, bias, -
    int64_t* prm_o = reinterpret_cast<int64_t*>(prm_f + 16);        // [4][2]: column (or temp offset) | query
    uint64_t* prm_t = reinterpret_cast<uint64_t*>(prm_o + 8);       // [4]: threshold key
    int32_t* prm_c = reinterpret_cast<int32_t*>(prm_t + 4);         // [4]: - threshold on the MFMA sum (accumulator init)
    uint64_t* lq_key = reinterpret_cast<uint64_t*>(prm_c + 12);     // [16 waves][4][ROT_LQ]: wave-private survivor queues

    const PQScanArgs& a = A.b;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15, n = lane & 15;
    const PQItemDesc it = A.item_desc[blockIdx.x];      // uniform 32-byte load: the whole work item
    if (it.l < 0) return;
    const int np = it.np, pair0 = it.pair0;
    const int64_t len = it.len;
    const int nblk = (int)(((len + 63) >> 6) << 2);     // 16-vector blocks of the list, slab padding included
    const int tb0 = it.tile * (16 * bpw);
    if (tb0 >= nblk) return;
    const uint8_t* lp = a.codes + (it.base_row >> 4) * (int64_t)(16 * M);

    // ---- code loads: buffer instructions on a descriptor of THIS list's blocks (base + size in SGPRs, the block's byte
    // offset in an SGPR, lane * 16 in one constant VGPR): no address VALU at all, and a block past the list's end reads
    // zeros instead of needing a clamp (its sums are garbage that the pos < len test of the survivor path drops)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)lp, 0, nblk * 16 * M, 0x00020000);
    const int vo16 = lane * 16, vo8 = lane * 8;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    v4u ca[ROT_D][NF > 0 ? NF : 1]; v2u cb[ROT_D];
#pragma unroll
    for (int dd = 0; dd < ROT_D; dd++) {     // the first ROT_D blocks of this wave are requested before the table staging
        const int so = (tb0 + w + 16 * dd) * (16 * M);
#pragma unroll
        for (int p = 0; p < NF; p++) ca[dd][p] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo16 + p * 1024, so, 0);
        if (NH) cb[dd] = __builtin_amdgcn_raw_buffer_load_b64(rs, vo8 + NF * 1024, so, 0);
    }

    // ---- stage the group's table: item = (code, 4 consecutive m) -> 4 dwords (one per m: byte k = query k, as int8 = u8 - 128)
    int64_t qq[4];
#pragma unroll
    for (int k = 0; k < 4; k++) qq[k] = A.pairs_sorted[pair0 + (k < np ? k : 0)] / a.nprobe;
    // var: MEASUREMENT-ONLY switches (RSX_ROT_VARIANT; wrong results): 1 = survivors dropped, 2 = no main loop, 4 = no table staging
    for (int e = tid; e < ((var & 4) ? 0 : 256 * (M / 4)); e += 1024) {
        const int c = e / (M / 4), m4 = e - c * (M / 4);
        uint32_t in[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            in[k] = (k < np) ? *reinterpret_cast<const uint32_t*>(A.lut8 + (qq[k] * 256 + c) * M + m4 * 4) : 0u;
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
    if (tid < 4) {
        const int k = tid;
        const int pi = A.pairs_sorted[pair0 + (k < np ? k : 0)];
        const int64_t q = pi / a.nprobe;
        const PQQParam p = A.qp[q];
        const int64_t col = a.seg_start[q * (a.nprobe + 1) + (pi - (int)q * a.nprobe)];
        const float dis0 = a.probe_dis0[pi];
        prm_f[k * 4 + 0] = dis0; prm_f[k * 4 + 1] = p.scale; prm_f[k * 4 + 2] = p.bias; prm_f[k * 4 + 3] = 0.0f;
        prm_o[k * 2 + 0] = FILTER ? col : q * a.tstride + col;
        prm_o[k * 2 + 1] = q;
        const uint64_t tau = FILTER ? A.tau_key[q * A.tau_stride] : 0ull;
        prm_t[k] = tau;
        // threshold on the integer sum: the smallest S whose score dis0 + fma(scale, S, bias) (the expression the
        // survivors are scored with, monotone in S) reaches the threshold key's score; C = S - 128 M is what the MFMA holds
        int thr = INT_MIN;
        if (k >= np) thr = INT_MAX;
        else if (FILTER && tau != 0ull) {
            const float ts = key_score(tau);
            if (!(dis0 + __fmaf_rn(p.scale, (float)(255 * M), p.bias) >= ts)) thr = INT_MAX;
            else {
                int lo = 0, hi = 255 * M;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (dis0 + __fmaf_rn(p.scale, (float)mid, p.bias) >= ts) hi = mid; else lo = mid + 1;
                }
                thr = lo - 128 * M;
            }
        }
        // stored as the MFMA accumulator's INITIAL value: the block's result is then >= 0 exactly for the survivors
        prm_c[k] = thr == INT_MAX ? -(1 << 30) : thr == INT_MIN ? (1 << 30) : -thr;
    }

    // ---- per-lane constants: rotation bytes, the one-hot B operand
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
    __syncthreads();
    const int cinit = FILTER ? (n < 4 ? prm_c[n] : -(1 << 30)) : 0;
    const v4i Ci = {cinit, cinit, cinit, cinit};
    // the survivor path's per-query constants ride in registers of the lanes that own the query (n = lane & 15 < 4)
    const int nq4 = n & 3;
    const float p_dis0 = prm_f[nq4 * 4], p_scale = prm_f[nq4 * 4 + 1], p_bias = prm_f[nq4 * 4 + 2];
    const uint32_t p_col = (uint32_t)prm_o[nq4 * 2];
    const uint64_t p_tau = prm_t[nq4];
    // survivor queues are PRIVATE to the wave: the slot of a survivor is the wave's running count for its query plus its rank
    // among this step's survivors of the same query (ballot + mbcnt) — no atomics (an LDS ds_add_rtn costs ~300 clk of the
    // whole CU's LDS pipe here: measured 0.7 ms of a 3.8 ms scan for 1.9 M survivors)
    const uint64_t QM = n < 4 ? (0x0001000100010001ull << n) : 0ull;   // the four lanes that own query n
    uint32_t qcnt = 0;                                                 // survivors of the lane's query so far (wave-uniform per query)
    uint64_t* myq = lq_key + ((size_t)w * 4 + nq4) * ROT_LQ;

    unsigned long long dbg_hit_clk = 0, dbg_hit_blocks = 0, dbg_blocks = 0;
    const unsigned long long dbg_t0 = (var & 32) ? __builtin_amdgcn_s_memtime() : 0;
    // ---- main loop: blocks tb0 + w + 16 j, j < bpw, ROT_D of them in flight
#pragma unroll 1
    for (int j0 = 0; j0 < ((var & 2) ? 0 : bpw); j0 += ROT_D) {
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
                if (var & 32) dbg_blocks++;
                if (__builtin_amdgcn_ballot_w64((C[0] & C[1] & C[2] & C[3]) >= 0) && !(var & 1)) {
                    const unsigned long long dbg_h0 = (var & 32) ? __builtin_amdgcn_s_memtime() : 0;
                    if (var & 32) dbg_hit_blocks++;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const bool cnd = C[r] >= 0;
                        if (__builtin_amdgcn_ballot_w64(cnd)) {
                            const int64_t pos = ((int64_t)b << 4) + 4 * g + r;
                            const float sc = p_dis0 + __fmaf_rn(p_scale, (float)(C[r] - cinit + 128 * M), p_bias);
                            const uint64_t key = (cnd && pos < len) ? make_key(sc, p_col + (uint32_t)pos) : 0ull;
                            const bool pass = key > p_tau;
                            const uint64_t mq = __builtin_amdgcn_ballot_w64(pass) & QM;      // this step's survivors of MY query
                            if (pass) {
                                const uint32_t slot = qcnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
                                if (slot < (uint32_t)ROT_LQ) myq[slot] = key;
                                else {   // the wave's queue is full (a dense stretch of the query's closest list): straight to HBM
                                    const int64_t q = prm_o[n * 2 + 1];
                                    const unsigned long long s2 = atomicAdd(&A.cand_cnt[q], 1ull);
                                    if (s2 < (unsigned long long)A.cand_cap) A.cand[q * A.cand_cap + s2] = key;
                                }
                            }
                            qcnt += (uint32_t)__builtin_popcountll(mq);
                        }
                    }
                    if (var & 32) dbg_hit_clk += __builtin_amdgcn_s_memtime() - dbg_h0;
                }
            } else {
                if (n < np && b < nblk) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int64_t pos = ((int64_t)b << 4) + 4 * g + r;
                        const float sc = prm_f[n * 4] + __fmaf_rn(prm_f[n * 4 + 1], (float)(C[r] + 128 * M), prm_f[n * 4 + 2]);
                        a.temp[prm_o[n * 2] + pos] = (pos < len) ? sc : -__builtin_inff();
                    }
                }
            }
        }
    }
    if (FILTER && (var & 32) && lane == 0) {
        unsigned long long* dbg = A.cand_cnt + A.b.nq;     // [8] diagnostics behind the per-query counters
        atomicAdd(&dbg[0], dbg_hit_blocks); atomicAdd(&dbg[1], dbg_hit_clk);
        atomicAdd(&dbg[2], __builtin_amdgcn_s_memtime() - dbg_t0); atomicAdd(&dbg[3], dbg_blocks);
    }
    if (FILTER) {
        // ---- every wave appends its own queues: ONE reservation per (wave, query) that has survivors, no barrier
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t c0 = __builtin_amdgcn_readlane(qcnt, k);           // lane k owns query k
            const uint32_t cq = c0 < (uint32_t)ROT_LQ ? c0 : (uint32_t)ROT_LQ;
            if (cq > 0) {
                const int64_t q = prm_o[k * 2 + 1];
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(&A.cand_cnt[q], (unsigned long long)cq);
                base = __shfl(base, 0);
                const uint64_t* src = lq_key + ((size_t)w * 4 + k) * ROT_LQ;
                for (uint32_t e = lane; e < cq; e += 64) {
                    const unsigned long long s2 = base + e;
                    if (s2 < (unsigned long long)A.cand_cap) A.cand[q * A.cand_cap + s2] = src[e];
                }
            }
        }
    }
}

template <int NF, int NH, bool FILTER>
static int launch_pq_scan_rot_t(PQScan8Args A, int bpw, void* desc_ws, hipStream_t st) {
    const size_t shm = (size_t)(NF + NH) * 65536 + 224 + (size_t)16 * 4 * ROT_LQ * 8;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)k_pq_scan_rot<NF, NH, FILTER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            return -1;
        attr = true;
    }
    dim3 grid((unsigned)pq_scan_rot_grid(A.max_items));
    A.item_desc = reinterpret_cast<const PQItemDesc*>(desc_ws);
    launch_pq_item_desc(A.item_off, A.group_off, A.pair_off, A.total_items, A.b.list_base, A.b.list_len, A.nlist, 4, grid.x,
                        reinterpret_cast<PQItemDesc*>(desc_ws), st);
    static int var = -1;
    if (var < 0) { const char* e = getenv("RSX_ROT_VARIANT"); var = e ? atoi(e) : 0; }
    hipLaunchKernelGGL((k_pq_scan_rot<NF, NH, FILTER>), grid, dim3(1024), shm, st, A, bpw, var);
    return 0;
}

// returns 0 on launch, -1 if this M has no rotated kernel.  tau_key == null: unfiltered (every score to a.temp).
int launch_pq_scan_rot(const PQScanArgs& a, const uint8_t* lut8t, const void* qparam, const int32_t* pairs_sorted,
                       const int32_t* pair_off, const int32_t* group_off, const int32_t* total_groups,
                       const int32_t* item_off, const int32_t* total_items, int nlist, int64_t max_items, int vpl,
                       const uint64_t* tau_key, int64_t tau_stride, uint64_t* cand, unsigned long long* cand_cnt,
                       int cand_cap, void* item_desc_ws, hipStream_t st) {
    if (a.CB != 0 || !item_desc_ws || !pq_rot_applies(a.M) || a.M != a.Mpad || max_items <= 0 || max_items > 0x7fffff00) return -1;
    PQScan8Args A;
    A.b = a; A.lut8 = lut8t; A.qp = (const PQQParam*)qparam;
    A.pairs_sorted = pairs_sorted; A.pair_off = pair_off; A.group_off = group_off; A.total_groups = total_groups;
    A.item_off = item_off; A.total_items = total_items;
    A.nlist = nlist; A.max_items = (int)max_items;
    A.tau_key = tau_key; A.tau_stride = tau_stride; A.cand = cand; A.cand_cnt = cand_cnt; A.cand_cap = cand_cap;
    const int bpw = 4 * vpl;   // tile = 16 waves x bpw blocks x 16 vectors = 1024 vpl vectors, as k_pq_scan8's
    const bool f = tau_key != nullptr;
    switch (a.M) {
        case 32: return f ? launch_pq_scan_rot_t<0, 1, true>(A, bpw, item_desc_ws, st) : launch_pq_scan_rot_t<0, 1, false>(A, bpw, item_desc_ws, st);
        case 64: return f ? launch_pq_scan_rot_t<1, 0, true>(A, bpw, item_desc_ws, st) : launch_pq_scan_rot_t<1, 0, false>(A, bpw, item_desc_ws, st);
        case 96: return f ? launch_pq_scan_rot_t<1, 1, true>(A, bpw, item_desc_ws, st) : launch_pq_scan_rot_t<1, 1, false>(A, bpw, item_desc_ws, st);
        case 128: return f ? launch_pq_scan_rot_t<2, 0, true>(A, bpw, item_desc_ws, st) : launch_pq_scan_rot_t<2, 0, false>(A, bpw, item_desc_ws, st);
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
    static size_t attr = 0;
    if (shm > attr) {
        if (hipFuncSetAttribute((const void*)k_pq_scan_rot_exact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return -1;
        attr = shm;
    }
    hipLaunchKernelGGL(k_pq_scan_rot_exact, dim3((unsigned)pairs, (unsigned)a.max_chunks), dim3(1024), shm, st, a);
    return 0;
}

}  // namespace rsx
