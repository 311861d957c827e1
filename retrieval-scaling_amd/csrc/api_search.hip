// api_search.hip — the search drivers: search_batch (one internal batch: coarse quantiser, probe selection, table build, threshold
// pre-pass, scan, selection, exact re-rank, certificate fallbacks) and search_impl (one rsx_search call).  Shared declarations: rsx_host.h.
#include <cstring>
#include "rsx_host.h"

// ---------------------------------------------------------------------------------------
// search
// ---------------------------------------------------------------------------------------
// profile = -1: events around the dominant scan launch ONLY (an event record is ~5 us of stream time: eight marks per batch are 1.5 % of the
// headline batch — bench.py times its `value` in this mode and takes the stage breakdown from a separate pass in mode 1)
struct StageTimer {
    rsx_index* h; bool on; bool scan_only; std::string prefix;
    hipEvent_t ev[64]; const char* name[64]; int n = 0;
    StageTimer(rsx_index* hh, const char* pre = "") : h(hh), on(hh->profile != 0), scan_only(hh->profile == -1), prefix(pre) {}
    void record(const char* nm) {
        (void)hipEventCreate(&ev[n]);
        (void)hipEventRecord(ev[n], h->st);
        name[n] = nm; n++;
    }
    void mark(const char* nm) {
        if (!on || n >= 64) return;
        if (scan_only && !(n == 1 && std::strcmp(nm, "scan") == 0)) return;
        record(nm);
    }
    void pre_scan() { if (on && scan_only && n == 0) record("prescan"); }      // right before the dominant scan's launch
    void finish() {
        if (!on || n == 0) return;
        (void)hipEventSynchronize(ev[n - 1]);
        for (int i = 1; i < n; i++) {
            float ms = 0; (void)hipEventElapsedTime(&ms, ev[i - 1], ev[i]);
            h->timing[prefix + name[i]] += ms;
        }
        float tot = 0; (void)hipEventElapsedTime(&tot, ev[0], ev[n - 1]);
        h->timing[prefix + "total"] += tot;
        for (int i = 0; i < n; i++) (void)hipEventDestroy(ev[i]);
        n = 0;
    }
};

static void kp_for(const rsx_index* h, int k, bool fast, int& KP, int& BUF) {
    int want;
    // fast IVF-PQ scan: K' = the candidates re-scored exactly per query.  With the round-3 threshold (k_pq_prepass: the sample's
    // k-th best approximate score minus 2 eps) the scan admits what the data needs — measured on the bench mixture ~130 keys
    // for k = 10, ~500 for k = 100, ~2300 for k = 1000, ~3400 for k = 2000 — and K' only has to hold them: 3k, at least
    // k + 118, at most 4096 (k_finalize sorts K' candidates in LDS).  A query with more candidates keeps its best K' by
    // approximate score and is still certified against the K'-th one (k_finalize) or re-run exactly.
    if (h->kind == KIND_IVFPQ && fast) want = h->pq_fast_kp > 0 ? std::max(k, h->pq_fast_kp) : std::min(4096, std::max(k + 118, 3 * k));
    else if (h->kind == KIND_IVFPQ) want = (k >= 512) ? k : k + 4;
    else want = k + std::max(8, k / 16);
    KP = std::max(16, pow2ceil(want));
    BUF = std::max(2 * KP, (h->kind == KIND_IVFPQ && fast) ? 512 : 256);
}

// top-k of `nrows` rows of fp32 scores (row r valid length: row_n or n_uniform) into state [nrows, KP]
static void select_rows(rsx_index* h, const float* scores, int64_t row_stride, const int64_t* row_n, int64_t row_n_stride,
                        int64_t n_max, uint32_t idx_base, int64_t nrows, int KP, int BUF, int k, uint64_t* state,
                        bool merge_state, unsigned long long* threshold_only_cnt = nullptr) {
    int64_t seg_len = std::max<int64_t>(4096, (int64_t)8 * BUF);
    seg_len = round_up(seg_len, 256);
    if (KP >= 1024 && n_max <= 131072) seg_len = round_up(n_max, 256);     // one segment: the radix selection (launch_select)
    int nseg = (int)std::max<int64_t>(1, (n_max + seg_len - 1) / seg_len);
    SelectArgs a{};
    a.in = scores; a.in_is_keys = 0; a.row_stride = row_stride;
    a.row_n = row_n; a.row_n_stride = row_n_stride; a.n_uniform = n_max;
    a.seg_len = seg_len; a.nseg = nseg; a.idx_base = idx_base;
    a.nrows = nrows; a.KP = KP; a.BUF = BUF; a.k = k;
    if (nseg == 1) {
        a.init = merge_state ? state : nullptr;
        a.out = state; a.out_row_stride = KP;
        if (threshold_only_cnt) { a.keep_last = 1; a.zero_cnt = threshold_only_cnt; }
        launch_select(a, h->st);
        return;
    }
    h->w_keys1.ensure((size_t)nrows * nseg * KP * 8);
    a.init = nullptr; a.out = h->w_keys1.as<uint64_t>(); a.out_row_stride = (int64_t)nseg * KP;
    if (nseg >= 4 && !merge_state) {
        // Phase A: segment 0 of every row (for IVF: the head of the closest list) alone; its k-th key is a
        // lower bound of the row's final k-th best, so (phase B) the other segments start from that
        // threshold and append almost nothing — no LDS sorts on the bulk of the row.
        SelectArgs a0 = a; a0.nseg = 1; a0.seg_base = 0;
        launch_select(a0, h->st);
        a.seg_base = 1;
        a.tau_ptr = h->w_keys1.as<uint64_t>() + (k - 1); a.tau_stride = (int64_t)nseg * KP;
        launch_select(a, h->st);
    } else {
        launch_select(a, h->st);
    }
    SelectArgs b{};
    b.in = h->w_keys1.p; b.in_is_keys = 1; b.row_stride = (int64_t)nseg * KP;
    b.row_n = nullptr; b.n_uniform = (int64_t)nseg * KP;
    b.seg_len = round_up((int64_t)nseg * KP, 256); b.nseg = 1; b.idx_base = 0;
    b.init = merge_state ? state : nullptr;
    b.out = state; b.out_row_stride = KP;
    b.nrows = nrows; b.KP = KP; b.BUF = BUF; b.k = k;
    if (threshold_only_cnt) { b.keep_last = 1; b.zero_cnt = threshold_only_cnt; }
    launch_select(b, h->st);
}

// Upper bound on the number of (list, tile, group) work items of a list-major scan without a host round
// trip: sum_l ceil(cnt_l/G)*tiles_l <= (nq * TQ)/G + sum_l tiles_l, TQ = tiles of the nprobe longest lists.
// (sum, max) of the nprobe largest values of ceil(len / unit) * scale over the lists (unit > 0), memoised per directory
// generation.  tag distinguishes the callers' (unit, scale) families.
static std::pair<int64_t, int64_t> top_probe_sum(rsx_index* h, int nprobe, int unit, int scale) {
    const auto key = std::make_tuple(nprobe, unit, scale);
    auto it = h->bound_cache.find(key);
    if (it != h->bound_cache.end() && it->second.first == h->dir_gen) return it->second.second;
    std::vector<int64_t> t((size_t)h->nlist);
    for (int l = 0; l < h->nlist; l++) t[(size_t)l] = (h->h_len[(size_t)l] + unit - 1) / unit * scale;
    const int np = std::min(nprobe, h->nlist);
    std::partial_sort(t.begin(), t.begin() + np, t.end(), std::greater<int64_t>());
    int64_t s = 0;
    for (int j = 0; j < np; j++) s += t[(size_t)j];
    const std::pair<int64_t, int64_t> r(s, np > 0 ? t[0] : 0);
    h->bound_cache[key] = std::make_pair(h->dir_gen, r);
    return r;
}
static int64_t max_scan_items(rsx_index* h, int64_t nq, int nprobe, int G, int tile_rows) {
    const auto key = std::make_tuple(-1, tile_rows, 0);       // all tiles of all lists
    int64_t all;
    auto it = h->bound_cache.find(key);
    if (it != h->bound_cache.end() && it->second.first == h->dir_gen) all = it->second.second.first;
    else {
        all = 0;
        for (int l = 0; l < h->nlist; l++) all += (h->h_len[(size_t)l] + tile_rows - 1) / tile_rows;
        h->bound_cache[key] = std::make_pair(h->dir_gen, std::make_pair(all, (int64_t)0));
    }
    const int64_t tq = top_probe_sum(h, nprobe, tile_rows, 1).first;
    return (nq * tq + G - 1) / G + all + 8;
}


// The IVF-PQ fast scan with in-kernel filtering and the one-launch threshold pre-pass never writes a score row (search_batch):
// such a search needs no [nq, sum of the nprobe longest lists] score buffer, and its internal batch is not bounded by one.
static void search_batch(rsx_index* h, int64_t nq, const void* dq, int dtype, int k, float* dD, int64_t* dI, bool allow_fast = true);

static bool pq_fast_applies(const rsx_index* h, int k, bool allow_fast) {      // the ONE definition of "this search takes the 8-bit fast scan"
    bool fast = allow_fast && h->kind == KIND_IVFPQ && h->metric == RSX_METRIC_INNER_PRODUCT && h->pq_fast != 0 && h->scan_kernel == 0 && (h->CB == 16 || pq_rot_family(h->CB)) && h->M * 255 < 65536;
    if (fast) { int KP, BUF; kp_for(h, k, true, KP, BUF); if (KP > 4096) fast = false; }
    // the sliced layout has ONE fast kernel: the filtered scan behind the one-launch pre-pass (every other setting takes the exact scan)
    if (fast && h->CB == PQ_SLICED && !(std::min(h->nprobe, h->nlist) > 1 && h->pq_filter != 0 && h->pq_prepass_fused != 0)) fast = false;
    return fast;
}
static bool pq_search_needs_score_rows(const rsx_index* h, int nprobe, int k) {
    return !(pq_fast_applies(h, k, true) && nprobe > 1 && h->pq_filter != 0 && h->pq_prepass_fused != 0);
}

// Keys a query's candidate row can hold (filtered IVF-PQ fast scan).  The threshold is valid by construction, so the
// row must hold every vector within 2 eps of the query's k-th best: ~700 keys at M = 96 / k = 10, but eps grows as the tables get
// coarser — at M = 16 (48 dimensions per 8-bit table entry) the measured mean is 1800 and the maximum 23 000 at k = 10.  An
// overflowing row sends its query to the exact re-run, so small M gets four times the room (8 B x nq x cap of HBM).
static int64_t pq_cand_cap(int k, int M) {
    int64_t cap = k > 512 ? 131072 : (k > 64 ? 65536 : 16384);
    if (M <= 32) cap = std::min<int64_t>(cap * 4, 262144);
    return cap;
}

// Queries whose certificate failed (h->w_uncertain, written by k_finalize) are re-run through the exact path of their index
// kind and their result rows replaced — rare, and what makes the fast paths EXACT rather than "almost always right".
// temp_bytes_per_query > 0 bounds the exact path's score buffer (Flat / IVF-Flat): the re-run proceeds in chunks.
static void rerun_uncertified(rsx_index* h, int64_t nq, const void* dq, int dtype, int k, float* dD, int64_t* dI,
                              size_t temp_bytes_per_query, const std::function<void()>& second_chance = nullptr) {
    std::vector<int32_t> bad_v;
    const int32_t* bad;
    const bool coarse_live = h->coarse_flags_live;
    h->coarse_flags_live = false;
    auto read_flags = [&]() {
        // the fast coarse quantiser's per-query flags (k_coarse_pick: 2 = its candidate row overflowed or a score was not finite) sit behind the
        // certificate's [nq, 2 nq): one copy fetches both, and a flagged query goes to the exact re-run whatever the certificate said
        // (the flag words live in page-locked host memory mapped into the device: the kernels wrote them across the bus, no copy command)
        const size_t words = (size_t)nq * (coarse_live ? 2 : 1);
        int32_t* dst;
        if (h->w_uncertain.host_mapped) {
            HIPCHECK(hipStreamSynchronize(h->st));
            bad_v.assign(h->w_uncertain.as<int32_t>(), h->w_uncertain.as<int32_t>() + words);
            dst = bad_v.data();
        } else {
            if (nq <= 4096 && h->pin_flags.ensure(8192 * 4)) dst = h->pin_flags.as<int32_t>();
            else { bad_v.resize(words); dst = bad_v.data(); }
            HIPCHECK(hipMemcpyAsync(dst, h->w_uncertain.p, words * 4, hipMemcpyDeviceToHost, h->st));
            HIPCHECK(hipStreamSynchronize(h->st));
        }
        if (coarse_live)
            for (int64_t q = 0; q < nq; q++) if (dst[(size_t)(nq + q)]) { dst[(size_t)q] |= 2; h->timing["coarse_redo_queries"] += 1.0; }
        bad = dst;
    };
    read_flags();
    if (second_chance) {
        // flag 1 = the certificate could not clear the query although none of its candidates was dropped: every vector that
        // can matter is still in its candidate row — re-rank from a larger K' there before paying for an exact scan
        int64_t n1 = 0;
        for (int64_t q = 0; q < nq; q++) n1 += bad[(size_t)q] == 1;
        if (n1 > 0) {
            h->timing["second_chance_queries"] += (double)n1;
            second_chance();
            read_flags();
        }
    }
    const int d = h->d;
    const size_t esz = dtype == RSX_F16 ? 2 : 4;
    std::vector<int64_t> badq;
    int64_t n_over = 0;
    for (int64_t q = 0; q < nq; q++) if (bad[(size_t)q]) {
        badq.push_back(q); n_over += (bad[(size_t)q] & 2) != 0;
        if (bad[(size_t)q] & 4) { h->timing["fallback_tie_queries"] += 1.0; h->timing["fallback_tie_max"] = std::max(h->timing["fallback_tie_max"], (double)(bad[(size_t)q] >> 8)); }
    }
    h->timing["fallback_overflow_queries"] += (double)n_over;     // of the fallbacks: candidate buffer / survivor segment overflows
    const int64_t nbad = (int64_t)badq.size();
    h->timing["fallback_queries"] += (double)nbad;
    h->timing["fast_queries"] += (double)nq;
    if (nbad == 0) return;
    int64_t chunk = nbad;
    if (temp_bytes_per_query > 0) chunk = std::max<int64_t>(1, std::min<int64_t>(nbad, (int64_t)(((size_t)2 << 30) / temp_bytes_per_query)));
    const size_t qrow = (size_t)d * esz;
    h->w_fbq.ensure((size_t)chunk * qrow);
    h->w_fbD.ensure((size_t)chunk * k * 4);
    h->w_fbI.ensure((size_t)chunk * k * 8);
    for (int64_t c0 = 0; c0 < nbad; c0 += chunk) {
        const int64_t nb = std::min(chunk, nbad - c0);
        // gather the uncertified queries into one contiguous batch, search it exactly, scatter the rows back
        for (int64_t i = 0; i < nb; i++)
            HIPCHECK(hipMemcpyAsync((char*)h->w_fbq.p + (size_t)i * qrow, (const char*)dq + (size_t)badq[(size_t)(c0 + i)] * qrow, qrow,
                                    hipMemcpyDeviceToDevice, h->st));
        search_batch(h, nb, h->w_fbq.p, dtype, k, h->w_fbD.as<float>(), h->w_fbI.as<int64_t>(), false);
        for (int64_t i = 0; i < nb; i++) {
            HIPCHECK(hipMemcpyAsync(dD + badq[(size_t)(c0 + i)] * k, h->w_fbD.as<float>() + i * k, (size_t)k * 4, hipMemcpyDeviceToDevice, h->st));
            HIPCHECK(hipMemcpyAsync(dI + badq[(size_t)(c0 + i)] * k, h->w_fbI.as<int64_t>() + i * k, (size_t)k * 8, hipMemcpyDeviceToDevice, h->st));
        }
    }
}

static void search_batch(rsx_index* h, int64_t nq, const void* dq, int dtype, int k, float* dD, int64_t* dI, bool allow_fast) {
    StageTimer tm(h, allow_fast ? "" : "fb_");
    const int d = h->d, ld = h->ld;
    // IVFPQ fast path: needs the 16-byte-granule layout, 16-bit integer sums, and K' <= 4096
    const bool rot = h->kind == KIND_IVFPQ && pq_rot_family(h->CB);      // block layouts (rotated, sliced): transposed tables, work-item scans
    const bool sliced = h->kind == KIND_IVFPQ && h->CB == PQ_SLICED;
    const bool fast = pq_fast_applies(h, k, allow_fast);
    int KP, BUF;
    kp_for(h, k, fast, KP, BUF);
    tm.mark("start");
    // queries: fp32 copy (exact re-rank, coarse quantiser, LUT) [nq, ld]; fp16 copy for the scans
    h->w_q32.ensure((size_t)nq * ld * 4);
    int64_t nq_pad = nq > 128 ? round_up(nq, 256) : 128;   // query tiles: 128 (k_flat_gemm) or 256 (k_flat_gemm2)
    const bool certify = h->kind != KIND_IVFPQ && allow_fast && h->flat_cert != 0;
    // fast coarse quantiser (round 6): fp16 MFMA scores of all lists + exact chains of the candidates; needs the re-run machinery behind it
    const int np0 = std::min(h->nprobe, h->nlist);
    const bool coarse_fast = h->kind != KIND_FLAT && allow_fast && h->coarse_fast != 0 && (h->kind == KIND_IVFPQ ? fast : certify) && nq >= 32 &&
                             np0 <= CP_MAXPROBE && h->nlist <= CP_MAXLIST && h->nlist > np0 && !h->h_centroids.empty() &&
                             coarse_pick_lds(h->nlist, d, np0) <= 150 * 1024;
    // ... whose fp16 operand is the caller's own batch when that is fp16, unpadded (d = ld) and whole tiles: the fp32 copy then rides in the
    // GEMM's launch (extra grid rows) instead of a conversion launch of its own
    // (not with the side-stream table build: it reads the fp32 copy before the GEMM has run)
    const bool q16_direct = h->kind == KIND_IVFPQ && coarse_fast && h->overlap == 0 && dtype == RSX_F16 && d == ld && nq % 128 == 0 && ((uintptr_t)dq & 15) == 0;
    if (q16_direct) {
    } else if (h->kind == KIND_IVFPQ && coarse_fast) {
        h->w_q16.ensure((size_t)round_up(nq, 128) * ld * 2);
        launch_convert_to_f32_f16(dq, dtype == RSX_F16, d, nq, d, h->w_q32.as<float>(), h->w_q16.as<__half>(), ld, round_up(nq, 128), h->st);
    } else
        launch_convert_to_f32(dq, dtype == RSX_F16, d, nq, d, h->w_q32.as<float>(), ld, h->st);
    if (h->kind != KIND_IVFPQ) {
        h->w_q16.ensure((size_t)nq_pad * ld * 2);
        h->w_flag.ensure(sizeof(int));
        HIPCHECK(hipMemsetAsync(h->w_flag.p, 0, sizeof(int), h->st));
        launch_convert_to_f16(dq, dtype == RSX_F16, nq, d, h->w_q16.as<__half>(), ld, nq_pad, h->w_flag.as<int>(), h->st);
    }
    h->w_state.ensure((size_t)nq * KP * 8);
    uint64_t* state = h->w_state.as<uint64_t>();
    if (!q16_direct) tm.mark("convert");       // (nothing was launched for it otherwise: the widening rides in the coarse GEMM)
    // Round 4 (overlap = 1, no longer the default): the 8-bit tables depend on the queries only — their build starts here on the side
    // stream and runs beside the coarse quantiser and the probe selection; the per-query parameters (k_pq_qparam: they need the coarse
    // scores) follow on the main stream once both have finished.  Round 6 measured the three cross-stream joins of a batch at ~12 us
    // each (profiles/r06_fixed_cost.md): the default is one stream, with the table build, the parameters and the pair grouping in two launches.
    const bool pq_fused_lut = h->kind == KIND_IVFPQ && fast && pq_lut8_fused_lds(h->M, h->Mpad, h->dsub) <= 160 * 1024 - 64;
    const bool side_lut = pq_fused_lut && h->overlap != 0 && h->dsub == 8 && h->lut_tiled != 0 && nq >= 64;
    // the finalize-from-the-row kernel (k_pq_final_tab) will serve this batch: let the table builder store the fp32 tables for it
    // (M KiB per query, 100 MB at M = 96 / batch 1024) instead of every query's workgroup re-deriving its table from the 786 KB codebook
    const bool tab_expected = pq_fused_lut && rot && h->pq_final_tab != 0 && (h->pq_final_tab == 2 || KP >= 512 || h->dsub > 8) &&
                              std::min(h->nprobe, h->nlist) > 1 && h->pq_filter != 0 && h->pq_prepass_fused != 0 &&
                              pq_final_tab_capacity(h->M, h->CB, k) > 0;
    // the matrix-core table build will serve this batch and the fast coarse quantiser runs: its first pass rides in the probe-pick launch
    const bool lut_pass0_early = coarse_fast && pq_fused_lut && !side_lut && h->dsub == 8 && h->lut_tiled >= 2 && h->metric == RSX_METRIC_INNER_PRODUCT &&
                                 h->pq_lut_early != 0;
    float* lut32_out = nullptr;
    if (tab_expected) { h->w_lut.ensure((size_t)nq * h->Mpad * 256 * 4); lut32_out = h->w_lut.as<float>(); }
    // everything that can refuse this search is checked BEFORE work is forked onto the side stream ...
    if (h->kind != KIND_FLAT) {
        const int np_ = std::min(h->nprobe, h->nlist);
        if (np_ > 4096) RSX_THROW(RSX_ERR_UNSUPPORTED, "search: %d probed lists per query exceed this build's maximum of 4096", np_);
        const int pad_ = (h->kind == KIND_IVFPQ) ? 64 : 16;
        if (std::max<int64_t>(round_up(top_probe_sum(h, np_, pad_, pad_).first, 256), 256) >= ((int64_t)1 << 32))
            RSX_THROW(RSX_ERR_UNSUPPORTED, "probed lists exceed 2^32 vectors per query");
    }
    // ... and whatever still throws behind the fork (an allocation) leaves through this guard: the side stream is drained before
    // the caller can reuse or free the buffers its kernels read and write
    struct SideJoin {
        rsx_index* h; bool armed = false;
        ~SideJoin() { if (armed && h->st2) (void)hipStreamSynchronize(h->st2); }
    } side_join{h};
    if (side_lut) {
        ensure_side_stream(h);
        h->w_lut8.ensure((size_t)nq * h->Mpad * 256);
        h->w_qparam.ensure((size_t)nq * 16);
        h->w_lutws.ensure(pq_lut8_tiled_ws(nq, h->Mpad));
        HIPCHECK(hipEventRecord(h->ev_fork, h->st));
        HIPCHECK(hipStreamWaitEvent(h->st2, h->ev_fork, 0));
        launch_pq_lut8(nullptr, h->w_q32.as<float>(), ld, h->d_codebooks.as<float>(), h->dsub, nq, h->M, h->Mpad, nullptr, 0,
                       h->w_lut8.as<uint8_t>(), h->w_qparam.p, h->w_lutws.p, sliced ? 2 : rot ? 1 : 0, h->st2, 1, lut32_out);
        HIPCHECK(hipEventRecord(h->ev_lut, h->st2));
        side_join.armed = true;
    }

    FinalizeArgs fa{};
    fa.kind = h->kind; fa.metric = h->metric; fa.state = state; fa.KP = KP; fa.k = k; fa.nq = nq;
    // Flat / IVF-Flat re-score rows of 2 d bytes that sit anywhere in HBM: only the k + max(8, k / 16) best approximate candidates the
    // certificate needs, not the power of two the selection and the sort round them up to (round 5: k = 1000 re-read 2048 rows per query)
    // (only where the certificate runs — `certify`: with flat_cert = 0 there is no second pass, and the uncertified mode keeps re-ranking all
    //  K' candidates as it always did: ADVICE r5)
    if (h->kind != KIND_IVFPQ && allow_fast && certify) fa.KPv = std::min(KP, k + std::max(8, k / 16));
    fa.list_base = h->d_base.as<int64_t>();
    fa.ids = (h->kind == KIND_FLAT && !h->custom_ids) ? nullptr : h->ids.as<int64_t>();
    fa.Q32 = h->w_q32.as<float>(); fa.ldq = ld; fa.d = d;
    fa.X = h->data.p; fa.x_f16 = h->storage_f16; fa.ld = ld;
    fa.D = dD; fa.I = dI;
    if (certify) {
        // |approx - exact| of the fp16-MFMA scan for ANY stored vector (Cauchy-Schwarz on the per-element errors):
        //   fp32 accumulation of d exact products       (d + 2) 2^-24 |q| |x|
        //   fp32 rows rounded to fp16 inside the scan   2^-11 |q| |x|        (fp16 storage is lossless)
        //   fp32 queries rounded to fp16                2^-11 |q| |x|        (only when the batch held such a value: device flag)
        //   L2: the ranking score adds -|x|^2/2 (fp32)  (d + 4) 2^-24 |x|^2  (folded into the absolute term)
        const float u24 = 5.9604645e-8f, u11 = 4.8828125e-4f;
        const float xmax = sqrtf(h->max_norm2) * 1.0000002f;
        h->w_uncertain.ensure((size_t)nq * 8);
        fa.uncertain = h->w_uncertain.as<int32_t>();
        fa.cert_xmax = xmax;
        fa.cert_rel = ((float)d + 2.0f) * u24 * 1.01f + (h->storage_f16 ? 0.0f : u11 * 1.002f);
        fa.cert_rel_qlossy = u11 * 1.002f + u11 * u11;
        fa.cert_qflag = h->w_flag.as<int>();
        fa.cert_abs = sqrtf((float)d) * u24 + (h->metric == RSX_METRIC_L2 ? ((float)d + 4.0f) * u24 * xmax : 0.0f);
    }

    if (h->kind == KIND_FLAT) {
        const float* bias = nullptr;
        if (h->metric == RSX_METRIC_L2) {
            // ranking score = <q,x> - |x|^2/2 ; bias buffer holds -|x|^2/2 (derived from norms)
            h->w_misc.ensure((size_t)h->ntotal * 4);
            bias = h->w_misc.as<float>();
        }
        const int64_t N = h->ntotal;
        if (N == 0) {
            launch_fill_u64(state, nq * KP, 0, h->st);
        } else if (!allow_fast) {
            // exact mode (queries the certificate could not clear): fp64 scores of every row, rounded once = the canonical
            // scores themselves, then the ordinary selection
            const int64_t tstride = round_up(N, 16);
            h->w_temp.ensure((size_t)nq * tstride * 4);
            ExactScoreArgs ea{};
            ea.kind = KIND_FLAT; ea.metric = h->metric; ea.nq = nq; ea.Q32 = h->w_q32.as<float>(); ea.ldq = ld; ea.d = d;
            ea.X = h->data.p; ea.x_f16 = h->storage_f16; ea.ld = ld; ea.flat_n = N;
            ea.temp = h->w_temp.as<float>(); ea.tstride = tstride;
            launch_exact_scores(ea, h->st);
            tm.mark("scan");
            select_rows(h, h->w_temp.as<float>(), tstride, nullptr, 0, N, 0, nq, KP, BUF, KP, state, false);
            tm.mark("select");
        } else if (nq <= 32) {
            // small batch: stream the database once per group of 16 queries (list-scan kernel)
            int64_t tstride = round_up(N, 16);
            h->w_temp.ensure((size_t)nq * tstride * 4);
            ListScanArgs a{};
            a.Q16 = h->w_q16.as<__half>(); a.ld = ld; a.X = h->data.p; a.x_f16 = h->storage_f16; a.bias = bias;
            a.list_base = h->d_base.as<int64_t>(); a.list_len = h->d_len.as<int64_t>();
            a.flat_mode = 1; a.flat_n = N; a.nq = (int)nq; a.nprobe = 1; a.nlist = 1;
            a.temp = h->w_temp.as<float>(); a.tstride = tstride;
            a.chunk_rows = 1024;
            if (list_scan2_chunk_rows(h->storage_f16, ld) > 0 && round_up(N, 16) / list_scan2_chunk_rows(h->storage_f16, ld) < 65535)
                a.chunk_rows = list_scan2_chunk_rows(h->storage_f16, ld);     // LDS-DMA streaming kernel
            a.max_groups = (int)((nq + 15) / 16);
            a.max_chunks = (int)((round_up(N, 16) + a.chunk_rows - 1) / a.chunk_rows);
            if (a.max_chunks > 65535) { a.chunk_rows = (int)round_up((round_up(N, 16) + 65534) / 65535, 64); a.max_chunks = (int)((round_up(N, 16) + a.chunk_rows - 1) / a.chunk_rows); }
            launch_list_scan(a, h->st);
            tm.mark("scan");
            // threshold = the KP-th approximate key (not the k-th): the exact re-rank needs the true top-KP
            select_rows(h, h->w_temp.as<float>(), tstride, nullptr, 0, N, 0, nq, KP, BUF, KP, state, false);
            tm.mark("select");
        } else {
            const int64_t CH = 65536;
            h->w_temp.ensure((size_t)nq_pad * CH * 4);
            launch_fill_u64(state, nq * KP, 0, h->st);
            // chunk 0 through the score buffer: its top-K' gives every query a running threshold
            int64_t done_rows = 0;
            auto chunk_pass = [&](int64_t v0, int64_t vend) {
                int64_t nv = std::min<int64_t>(CH, vend - v0);
                launch_flat_gemm(h->w_q16.as<__half>(), (int)nq_pad, h->data.p, h->storage_f16, v0, nv, ld, bias,
                                 h->w_temp.as<float>(), CH, h->st);
                select_rows(h, h->w_temp.as<float>(), CH, nullptr, 0, nv, (uint32_t)v0, nq, KP, BUF, KP, state, true);
            };
            // Threshold phase: the K'-th key of the first rows is a threshold for everything behind them, so a filtered pass over
            // the rows [a, b) keeps ~K' (b - a) / a keys per query.  One 65536-row chunk is right for k = 10 (K' = 32: 5 k keys at
            // 10M rows).  For the reference's n_docs = 1000 (K' = 2048) one chunk let 310 k keys through, overflowed every candidate row
            // and fell back to 153 chunk passes (550 ms per batch, round 4); 160 K' rows and ONE filtered launch over the rest still
            // emitted 62 k keys per query — 64 M atomically placed keys, the filtered GEMM 27.8 instead of 17.4 ms — behind five
            // chunk selections of 0.93 ms.  Now: flat_pre_mult x K' rows through the score buffer (default 16, in units: below), then the rest
            // in STAGES of geometrically growing row ranges, each one filtered launch + one selection that tightens the threshold for
            // the next: S stages of ratio r = (N / first)^(1/S) emit ~S K' (r - 1) keys.  Measured at 10M x 768, batch 1024
            // (profiles/r04_flat_staged_filter.md): k = 1000 36.5 -> 23.3 ms (S = 5), k = 10 18.9 -> 17.2 ms (S = 2: even 5 k keys
            // per query cost the single filtered launch 1.6 ms), k = 100 17.8 ms.
            // (round 6: for small K' the threshold phase and the stage boundaries count in UNITS of a quarter chunk — at K' = 32 the selection
            //  over a full 65536-column chunk cost 0.43 ms of a 16.9 ms batch, four times what a 16384-row threshold phase needs; the
            //  stages pass ~1.5 k keys per query instead of ~700, which the queued epilogue does not notice.  Larger K': half a chunk and 16 K' rows —
            //  k = 1000: scan0 1.32 -> 0.62 ms, one stage more, 20.07 -> 19.68 ms per batch (profiles/r06_large_k_flat.md).  flat_pre_unit: rows, 0 = default)
            const int64_t U = h->flat_pre_unit > 0 ? std::min<int64_t>(CH, round_up(h->flat_pre_unit, 256)) : (KP <= 64 ? CH / 4 : CH / 2);
            const int64_t nchunks = (N + U - 1) / U;          // ... in units
            const int64_t n0 = std::min<int64_t>(nchunks, std::max<int64_t>(1, ((int64_t)KP * std::max(1, h->flat_pre_mult) + U - 1) / U));
            for (int64_t c = 0; c < n0; c++) chunk_pass(c * U, std::min<int64_t>(N, (c + 1) * U));
            done_rows = std::min<int64_t>(n0 * U, N);
            tm.mark("scan0");
            if (done_rows < N && h->flat_filter != 0) {
                int S = h->flat_stages;
                if (S <= 0) S = KP <= 64 ? 2 : std::min(6, std::max(1, (int)lround(log((double)nchunks / (double)n0) / log(3.0))));
                const double r = pow((double)nchunks / (double)n0, 1.0 / S);
                const int cap = KP <= 64 ? 32768 : 131072;
                h->w_cand.ensure((size_t)nq * cap * 8);
                // With the certificate behind it (the default) a stage does not wait for its counts: every stage owns a set of per-query
                // counters, the selection clamps an overflowed row to the buffer, and k_finalize flags a query any of whose stage rows
                // overflowed — it joins the exact re-run like a query the certificate could not clear.  Round 6: the count read-back +
                // host synchronisation per stage (five at the reference's n_docs = 1000) left the GPU idle between the stages.
                // flat_cert = 0 has no re-run behind it and keeps the per-stage check (an overflowed stage is redone chunk by chunk).
                const bool deferred = certify;
                const size_t cc_stride = (size_t)nq * CCS;
                h->w_candcnt.ensure(cc_stride * 8 * (deferred ? (size_t)S : 1));
                if (deferred) HIPCHECK(hipMemsetAsync(h->w_candcnt.p, 0, cc_stride * 8 * (size_t)S, h->st));
                std::vector<unsigned long long> cnts(deferred ? 0 : cc_stride);
                int stages_run = 0;
                for (int st_ = 0; st_ < S && done_rows < N; st_++) {
                    // stage boundaries on chunk multiples (the database tiles of the GEMM stay aligned)
                    int64_t endc = st_ == S - 1 ? nchunks : std::min<int64_t>(nchunks, std::max<int64_t>(done_rows / U + 1, (int64_t)llround((double)n0 * pow(r, st_ + 1))));
                    const int64_t end = std::min<int64_t>(N, endc * U);
                    unsigned long long* cc = h->w_candcnt.as<unsigned long long>() + (deferred ? cc_stride * (size_t)st_ : 0);
                    // ONE GEMM launch over the stage's rows whose epilogue keeps only keys beating the running K'-th key
                    if (!deferred) HIPCHECK(hipMemsetAsync(cc, 0, cc_stride * 8, h->st));
                    launch_flat_gemm_filter(h->w_q16.as<__half>(), (int)nq_pad, (int)nq, h->data.p, h->storage_f16, done_rows,
                                            end - done_rows, ld, bias, state + (KP - 1), KP, h->w_cand.as<uint64_t>(), cc, cap, h->st);
                    tm.mark("scan");
                    bool filtered_ok = true;
                    if (!deferred) {
                        HIPCHECK(hipMemcpyAsync(cnts.data(), cc, cc_stride * 8, hipMemcpyDeviceToHost, h->st));
                        HIPCHECK(hipStreamSynchronize(h->st));
                        for (int64_t qi = 0; qi < nq; qi++) if (cnts[(size_t)qi * CCS] > (unsigned long long)cap) { filtered_ok = false; break; }
                    }
                    if (filtered_ok) {
                        SelectArgs b{};
                        b.in = h->w_cand.p; b.in_is_keys = 1; b.row_stride = cap;
                        b.row_n = reinterpret_cast<const int64_t*>(cc); b.row_n_stride = CCS; b.n_uniform = cap;
                        b.seg_len = cap; b.nseg = 1; b.idx_base = 0;
                        b.init = state; b.out = state; b.out_row_stride = KP;
                        b.nrows = nq; b.KP = KP; b.BUF = BUF; b.k = KP;
                        launch_select(b, h->st);
                        tm.mark("select");
                    } else {
                        h->timing["flat_filter_overflows"] += 1;   // adversarial order: redo this stage's rows chunk by chunk
                        for (int64_t v0 = done_rows; v0 < end; v0 += CH) chunk_pass(v0, end);
                        tm.mark("scan");
                    }
                    done_rows = end;
                    stages_run = st_ + 1;
                }
                if (deferred) { fa.cand_cnt = h->w_candcnt.as<unsigned long long>(); fa.cand_cap = cap; fa.cand_cnt_n = stages_run; fa.cand_cnt_stride = (int64_t)cc_stride; }
            }
            if (done_rows < N) {       // flat_filter = 0
                for (int64_t v0 = done_rows; v0 < N; v0 += CH) chunk_pass(v0, N);
                tm.mark("scan");
            }
        }
        launch_finalize(fa, h->st);
        tm.mark("finalize");
        tm.finish();
        if (certify && N > 0) rerun_uncertified(h, nq, dq, dtype, k, dD, dI, (size_t)round_up(N, 16) * 4);
        return;
    }

    // ---------------- IVF ----------------
    const int nlist = h->nlist;
    const int nprobe = std::min(h->nprobe, nlist);
    // 1. coarse quantiser (exact fp32) + top-nprobe
    const int nlp = (int)round_up(nlist, 4);   // row stride of the coarse scores: 16-byte aligned rows for k_select
    const int pad_to = (h->kind == KIND_IVFPQ) ? 64 : 16;
    h->w_probelist.ensure((size_t)nq * nprobe * 4);
    h->w_dis0.ensure((size_t)nq * nprobe * 4);
    h->w_segstart.ensure((size_t)nq * (nprobe + 1) * 8);
    if (coarse_fast) {
        const int64_t nqp = h->kind == KIND_IVFPQ ? round_up(nq, 128) : nq_pad;
        const int64_t nl128 = round_up(nlist, 128);
        if (h->cent16_gen != h->cent_gen || !h->cent16.p) {       // the centroids' fp16 copy + the norm bound, once per centroid set
            h->cent16.ensure((size_t)nl128 * ld * 2);
            launch_convert_to_f16(h->d_centroids.p, 0, nlist, d, h->cent16.as<__half>(), ld, nl128, nullptr, h->st);
            double mx = 0.0;
            for (int l = 0; l < nlist; l++) { double n2 = 0.0; for (int t = 0; t < d; t++) { const double v = h->h_centroids[(size_t)l * d + t]; n2 += v * v; } mx = std::max(mx, n2); }
            h->cent_cmax = (float)(std::sqrt(mx) * 1.000001);
            h->cent16_gen = h->cent_gen;
        }
        h->w_coarse.ensure((size_t)nqp * nlp * 4);
        if (q16_direct) launch_coarse_approx((const __half*)dq, nqp, h->cent16.as<__half>(), nlist, ld, h->w_coarse.as<float>(), nlp, h->st, h->w_q32.as<float>(), nq * (int64_t)ld);
        else launch_coarse_approx(h->w_q16.as<__half>(), nqp, h->cent16.as<__half>(), nlist, ld, h->w_coarse.as<float>(), nlp, h->st);
        tm.mark("coarse");
        h->w_uncertain.ensure((size_t)nq * 8);
        CoarsePickArgs cp{};
        cp.approx = h->w_coarse.as<float>(); cp.nlp = nlp; cp.nlist = nlist; cp.Q32 = h->w_q32.as<float>(); cp.ld = ld; cp.d = d;
        cp.C = h->d_centroids.as<float>(); cp.cmax = h->cent_cmax;
        cp.ef = (dtype == RSX_F16 ? 1.0f : 2.0f) * 4.8828125e-4f;        // centroids rounded to fp16; fp32 queries as well
        cp.nprobe = nprobe; cp.list_len = h->d_len.as<int64_t>(); cp.pad_to = pad_to;
        cp.probe_list = h->w_probelist.as<int32_t>(); cp.dis0 = h->w_dis0.as<float>(); cp.seg_start = h->w_segstart.as<int64_t>();
        cp.bad = h->w_uncertain.as<int32_t>() + nq;
        if (lut_pass0_early) {      // pass 0 of the table build needs the queries only: extra workgroups of this launch
            h->w_lutws.ensure(pq_lut8_tiled_ws(nq, h->Mpad));
            cp.lp0 = LutPass0Args{h->w_q32.as<float>(), ld, h->d_codebooks.as<float>(), h->M, h->Mpad, nq, reinterpret_cast<float*>(h->w_lutws.p),
                                  pq_lut_pass0_blocks(nq, h->Mpad)};
        }
        launch_coarse_pick(cp, nq, h->st);
        h->coarse_flags_live = true;
    } else {
        h->w_coarse.ensure((size_t)nq * nlp * 4);
        launch_gemm_exact_scores(h->w_q32.p, 0, nq, ld, h->d_centroids.as<float>(), nlist, d, h->w_coarse.as<float>(), nlp, h->st);
        tm.mark("coarse");
        int KPp = std::max(16, pow2ceil(nprobe));
        int BUFp = std::max(2 * KPp, 256);
        h->w_probekeys.ensure((size_t)nq * KPp * 8);
        select_rows(h, h->w_coarse.as<float>(), nlp, nullptr, 0, nlist, 0, nq, KPp, BUFp, nprobe, h->w_probekeys.as<uint64_t>(), false);
        // 2. probe set-up
        launch_probe_setup(h->w_probekeys.as<uint64_t>(), KPp, nq, nprobe, h->d_len.as<int64_t>(), pad_to,
                           h->w_probelist.as<int32_t>(), h->w_dis0.as<float>(), h->w_segstart.as<int64_t>(), h->st);
    }
    if (side_lut) HIPCHECK(hipEventRecord(h->ev_probe, h->st));
    tm.mark("select_probe");
    if (h->profile >= 2) {
        std::vector<int32_t> pl((size_t)nq * nprobe);
        HIPCHECK(hipMemcpyAsync(pl.data(), h->w_probelist.p, pl.size() * 4, hipMemcpyDeviceToHost, h->st));
        HIPCHECK(hipStreamSynchronize(h->st));
        double tot = 0;
        std::vector<int32_t> pc((size_t)nlist, 0);
        for (int32_t l : pl) if (l >= 0) { tot += (double)h->h_len[(size_t)l]; pc[(size_t)l]++; }
        h->timing["scanned_vectors"] += tot;
        // the same batch seen list-major: vectors of every list probed at least once (what HBM must deliver), and vectors x
        // groups of 4 probing queries (what the IVF-PQ fast scan gathers)
        // (queries per table gather of the scan this index takes: 8 for the sliced layout and the M = 64 eight-query form, 16 at M = 16, else 4)
        const bool sl8_ = sliced && (h->pq_q8 == 2 || (h->pq_q8 != 0 && nq * (int64_t)nprobe >= (int64_t)3 * nlist && h->ntotal >= (int64_t)4096 * nlist));
        const int gq = h->kind == KIND_IVFPQ && pq_rot_family(h->CB) ? (sliced ? (sl8_ ? 8 : 4) : 4 * pq_scan_rot_ngq(h->M, true, h->pq_q8)) : 4;
        double uniq = 0, grp = 0;
        for (int l = 0; l < nlist; l++)
            if (pc[(size_t)l]) { uniq += (double)h->h_len[(size_t)l]; grp += (double)h->h_len[(size_t)l] * ((pc[(size_t)l] + gq - 1) / gq); }
        h->timing["scanned_unique_vectors"] += uniq; h->timing["scanned_group_vectors"] += grp; h->timing["scan_group_queries"] = (double)gq;
        tm.mark("count");
    }
    // host-side bound on a query's row of the score buffer: the nprobe longest (padded) lists
    const auto padded = top_probe_sum(h, nprobe, pad_to, pad_to);    // the nprobe longest lists, padded: sum and maximum
    int64_t tmax = padded.first, maxlen = padded.second;
    tmax = std::max<int64_t>(round_up(tmax, 256), 256);
    // score rows [nq, tmax]: every path but the filtered IVF-PQ fast scan with the one-launch pre-pass fills (part of) them.  At the
    // reference's nprobe 512 a row is 35 MB: allocating it unconditionally used to cut a 1024-query batch into four internal
    // batches (round 4: 4x the fixed stages, a quarter of the queries per list group)
    if (allow_fast ? pq_search_needs_score_rows(h, nprobe, k) : true) h->w_temp.ensure((size_t)nq * tmax * 4);
    bool filtered = false;   // fast path with in-kernel candidate filtering (no full score buffer)
    bool flag_overflows = false;   // IVF-Flat filtered scan: a few candidate rows overflowed — k_finalize sends their queries to the exact re-run
    bool use_gather = false; int gs_tmax = 0; PQGatherArgs gs{};   // ... whose candidates are gathered and selected in one launch
    bool fused_pre_used = false;   // ... whose threshold came from the one-launch pre-pass (complete candidate rows: second chance)
    int cand_cap = 0;
    // the exact kernels gather fp32 table entries; the fast path builds the table in LDS (when it fits)
    const bool fused_lut = pq_fused_lut;

    if (h->kind == KIND_IVFPQ) {
        const bool pq_l2 = h->metric == RSX_METRIC_L2;      // per-(query, list) tables, built inside the scan (k_pq_scan_l2)
        if (!fused_lut && !pq_l2) {
            h->w_lut.ensure((size_t)nq * h->Mpad * 256 * 4);
            launch_pq_lut(h->w_q32.as<float>(), ld, nq, d, h->M, h->Mpad, h->d_codebooks.as<float>(), h->w_lut.as<float>(), h->st);
            tm.mark("lut");
        }
        PQScanArgs a{};
        a.codes = h->data.as<uint8_t>(); a.M = h->M; a.Mpad = h->Mpad; a.CB = h->CB;
        a.list_base = h->d_base.as<int64_t>(); a.list_len = h->d_len.as<int64_t>();
        a.lut = h->w_lut.as<float>(); a.probe_list = h->w_probelist.as<int32_t>(); a.probe_dis0 = h->w_dis0.as<float>();
        a.seg_start = h->w_segstart.as<int64_t>(); a.nq = nq; a.nprobe = nprobe;
        a.temp = h->w_temp.as<float>(); a.tstride = tmax;
        int64_t max_slabs = std::max<int64_t>(1, maxlen / 64);
        int64_t pairs = nq * nprobe;
        bool done = false;
        if (fast) {
            // 8-bit tables, 4 queries per LDS read; approximate scores, certified in k_finalize
            h->w_lut8.ensure((size_t)nq * h->Mpad * 256);
            h->w_qparam.ensure((size_t)nq * 16);
            h->w_uncertain.ensure((size_t)nq * 8);
            void* lut_ws = nullptr;
            if (fused_lut && h->dsub == 8 && h->lut_tiled != 0) { h->w_lutws.ensure(pq_lut8_tiled_ws(nq, h->Mpad)); lut_ws = h->w_lutws.p; }
            // the 8-bit tables + per-query parameters; pg: the (query, probe) pairs grouped by list in the same launches (matrix-core form only).
            // Called once the scan's tile size is known (the grouping needs it)
            bool tables_built = false;
            auto build_tables = [&](const PairGroupArgs* pg) {
                if (tables_built) return;
                tables_built = true;
                if (side_lut) HIPCHECK(hipStreamWaitEvent(h->st, h->ev_lut, 0));     // the tables were built beside the probe selection
                launch_pq_lut8(fused_lut ? nullptr : h->w_lut.as<float>(), h->w_q32.as<float>(), ld, h->d_codebooks.as<float>(), h->dsub, nq, h->M, h->Mpad,
                               h->w_dis0.as<float>(), nprobe, h->w_lut8.as<uint8_t>(), h->w_qparam.p, lut_ws, sliced ? 2 : rot ? 1 : 0, h->st, side_lut ? 2 : 0,
                               fused_lut ? lut32_out : nullptr, (lut_ws && !side_lut && h->lut_tiled >= 2) ? (lut_pass0_early ? 2 : 1) : 0, pg);
                tm.mark("lut8");
            };
            const bool group_in_tables = lut_ws && !side_lut && h->lut_tiled >= 2 && h->pq_group_fused != 0 && pairs <= PG_MAX_PAIRS &&
                                         nlist <= PG_BLOCKS * PG_MAX_LPB;
            int rot_log_cap = 64;
            auto rot_desc = [&](int64_t items, int ngq) -> void* {   // work-item records + run descriptors + survivor logs of the rotated-layout scan
                // the log pool = (persistent workgroups x 64 logs x log_cap keys): 1 / 2 / 4 GiB by k, never more than a quarter of the
                // temp budget.  A log that fills up only sends the queries of its later runs to the exact re-run (counted); the pool is
                // touched where survivors land, so its size costs nothing per batch; reported by rsx_get "workspace_bytes"
                const int nwg = pq_scan_rot_max_wgs(h->M) * ngq;
                int64_t pool = (int64_t)(k <= 64 ? 1 : k <= 512 ? 2 : 4) << 30;
                pool = std::min(pool, std::max<int64_t>(h->temp_budget / 4, (int64_t)64 << 20));
                int64_t cap = pool / 8 / ((int64_t)nwg * 64);
                cap = std::max<int64_t>(64, cap / 16 * 16);
                if (h->pq_log_cap > 0) cap = h->pq_log_cap;            // tests starve the logs to force the overflow path
                rot_log_cap = (int)std::min<int64_t>(cap, (int64_t)1 << 24);
                h->w_itemdesc.ensure(pq_scan_rot_ws(items * ngq, rot_log_cap, nwg));
                return h->w_itemdesc.p;
            };
            // sliced layout: eight queries per gather (two records per item) where lists are long and shared by several queries; a handful of
            // queries (about one per probed list) or short lists take the four-query single-pass scan (k_pq_scan_sl4).  pq_q8: 1 = choose,
            // 2 = always eight, 0 = always four
            const bool sl_eight = sliced && (h->pq_q8 == 2 || (h->pq_q8 != 0 && pairs >= (int64_t)3 * nlist && h->ntotal >= (int64_t)4096 * nlist));
            const int ngq = sliced ? (sl_eight ? 2 : 1) : rot ? pq_scan_rot_ngq(h->M, true, h->pq_q8) : 1;     // the filtered scan's 4-query records per work item (M = 16: 4)
            int64_t avg_slabs = std::max<int64_t>(1, (h->ntotal / std::max(1, nlist) + 63) / 64);
            // rotated layout: persistent workgroups draw items dynamically, so the tile is the whole (average) list — one table
            // staging per (list, query group) — as long as that leaves a few thousand items to balance over 256 CUs
            int vpl = 8;
            if (rot) { vpl = 32; while (vpl > 8 && 16 * (vpl / 2) >= avg_slabs) vpl /= 2; }
            if (h->scan_chunk > 0) vpl = std::max(1, std::min(rot ? 64 : 16, h->scan_chunk / 1024));
            else {
                // enough items to balance the chip: a few thousand for a full batch; for a handful of queries every (query, list)
                // pair is its own group and each item stages a whole table, so one item per CU is the better trade
                const int64_t groups_est = pairs <= nlist / 4 ? pairs : pairs / 4 + 1;
                const int64_t want_items = pairs <= nlist / 4 ? 256 : 2048;
                while (vpl > 1 && groups_est * ((avg_slabs + 16 * vpl - 1) / (16 * vpl)) < want_items) vpl /= 2;
            }
            if (vpl != 64 && vpl != 32 && vpl != 16 && vpl != 8 && vpl != 4 && vpl != 2) vpl = 1;
            const int tile_rows = 64 * 16 * vpl;
            h->w_pairs.ensure((size_t)(pairs + 5 * (size_t)(nlist + 1) + 8) * 4);
            int32_t* pairs_sorted = h->w_pairs.as<int32_t>();
            int32_t* cnt = pairs_sorted + pairs;
            int32_t* cursor = cnt + (nlist + 1);
            int32_t* pair_off = cursor + (nlist + 1);
            int32_t* group_off = pair_off + (nlist + 1);
            int32_t* item_off = group_off + (nlist + 1);
            int32_t* total_groups = item_off + (nlist + 1);
            int32_t* total_items = total_groups + 1;
            // Two stages, so that only a sliver of the scores ever leaves the scan kernel:
            //  stage 1: score ONLY the first tile of each query's closest list (probe rank 0) into the score
            //           buffer and take its top-K' -> state0; its K'-th key is a lower bound of the query's final
            //           K'-th best key;
            //  stage 2: scan everything else (all probes, all tiles, minus that piece) in the multi-query groups,
            //           appending to a small per-query candidate buffer only the keys that beat the bound
            //           (wave-aggregated atomics); a final select merges them with state0.
            // A full candidate buffer marks the query uncertain (-> exact fallback), so this is always exact.
            filtered = (nprobe > 1) && (h->pq_filter != 0);
            // the pre-pass only has to produce a threshold: it scores a (smaller) prefix of the closest list
            int pre_vpl = vpl;
            if (filtered && h->pq_pre_rows > 0) {
                while (pre_vpl > 1 && 64 * 16 * pre_vpl > h->pq_pre_rows) pre_vpl /= 2;
                while (pre_vpl > 1 && 64 * 16 * pre_vpl < KP * 4) pre_vpl *= 2;   // ... but well above K' candidates
                if (pre_vpl > vpl) pre_vpl = vpl;
            }
            int pre_rows = filtered ? 64 * 16 * pre_vpl : tile_rows;
            // one-launch pre-pass (k_pq_prepass: score a prefix of the closest list with byte gathers on the query's own table,
            // 16-bit integer sums in LDS, k-th largest by a radix walk -> threshold a_k - 2 eps) when its LDS footprint allows;
            // else grouping + scan of the prefix + selection (K'-th key of the prefix as the threshold)
            bool fused_pre = filtered && h->pq_prepass_fused != 0;
            bool pre4 = false;
            bool grouped_early = false;
            if (fused_pre) {
                // the sample's k-th best score is the threshold: the sample must be a large part of the closest list once k is large
                // (measured at 24k-vector lists: a 2048-vector prefix gives ~1000 candidates per query for k = 10 but ~20000 for
                // k = 100) — 160 k vectors, at least pq_pre_rows, at most 32768 (64 KiB of 16-bit sums in LDS)
                // (round 3, measured on the bench index at k = 10: 2048 / 4096 / 8192 / 16384 sample rows leave 1046 / 633 / 372 / 217
                // candidates per query; the pre-pass costs 85 / 131 / 239 / 446 us and the scan 2.50 / 2.41 / 2.43 / 2.42 ms: 4096 is
                // the best total for a full batch, a few queries keep the cheaper 2048)
                int64_t base_rows = h->pq_pre_rows > 0 ? h->pq_pre_rows : 2048;
                if (nq <= 64) base_rows = std::min<int64_t>(base_rows, 2048);
                int64_t want_rows = std::max<int64_t>(base_rows, std::min<int64_t>(h->pq_pre_max, (int64_t)h->pq_pre_mult * k));
                want_rows = std::min<int64_t>(round_up(want_rows, 64), round_up(std::max<int64_t>(maxlen, 64), 64));
                // small k, full batch, rotated layout: the 4-queries-per-workgroup form (k_pq_prepass4) — its sample is what fits the LDS
                // beside the four-query table image (3520 rows at M = 96)
                pre4 = rot && h->pq_prepass4 != 0 && nq >= 64 && (int64_t)160 * k <= base_rows && pq_prepass4_max_rows(h->Mpad) >= 1024;
                if (pre4) want_rows = std::min<int64_t>(want_rows, pq_prepass4_max_rows(h->Mpad));
                pre_rows = (int)want_rows;
                fused_pre = (size_t)pre_rows * 2 + (size_t)h->Mpad * 256 + 2048 <= 150 * 1024;
                if (!fused_pre) pre_rows = 64 * 16 * pre_vpl;
            }
            if (fused_pre) {
                cand_cap = (int)std::min<int64_t>(std::max<int64_t>(tmax, 1024), pq_cand_cap(k, h->M));
                h->w_cand.ensure((size_t)nq * cand_cap * 8);
                h->w_candcnt.ensure((size_t)nq * 8 * CCS);
                PQPrepassArgs pa{};
                pa.codes = h->data.as<uint8_t>(); pa.list_base = h->d_base.as<int64_t>(); pa.list_len = h->d_len.as<int64_t>();
                pa.probe_list = h->w_probelist.as<int32_t>(); pa.probe_dis0 = h->w_dis0.as<float>();
                pa.seg_start = h->w_segstart.as<int64_t>();
                pa.lut8 = h->w_lut8.as<uint8_t>(); pa.qparam = h->w_qparam.as<float>();
                pa.nprobe = nprobe; pa.Mpad = h->Mpad; pa.pre_rows = pre_rows; pa.KP = KP; pa.CB = h->CB;
                pa.k = k;
                pa.state = state; pa.cand_cnt = h->w_candcnt.as<unsigned long long>();
                h->w_tau.ensure((size_t)nq * 8);
                pa.tau = h->w_tau.as<uint64_t>();
                if (rot) {       // the sample's own candidates leave from the pre-pass; the scan drops that (query, list, tile 0)
                    h->w_excl.ensure((size_t)nq * 2);
                    pa.cand = h->w_cand.as<uint64_t>(); pa.cand_cap = cand_cap; pa.tile_rows = tile_rows; pa.excl = h->w_excl.as<uint16_t>();
                }
                if (group_in_tables) {      // the pairs grouped by list by extra workgroups of the table launch
                    if (!h->w_pgflags.p) { h->w_pgflags.ensure((size_t)PG_BLOCKS * 16); HIPCHECK(hipMemsetAsync(h->w_pgflags.p, 0, (size_t)PG_BLOCKS * 16, h->st)); }
                    PairGroupArgs pg{};
                    pg.probe_list = h->w_probelist.as<int32_t>(); pg.list_len = h->d_len.as<int64_t>();
                    pg.pair_off = pair_off; pg.group_off = group_off; pg.item_off = item_off; pg.total_groups = total_groups; pg.total_items = total_items;
                    pg.pairs_sorted = pairs_sorted; pg.flags = h->w_pgflags.as<uint32_t>();
                    pg.npairs = (int)pairs; pg.nlist = nlist; pg.G = 4 * ngq; pg.tile_rows = tile_rows; pg.tile_cap = 0; pg.nb = PG_BLOCKS;
                    if (++h->pg_epoch == 0) h->pg_epoch = 1;
                    pg.epoch = h->pg_epoch;
                    build_tables(&pg);
                    grouped_early = true;
                }
                build_tables(nullptr);
                if (side_lut) {       // the (list, tile, group) work items of the scan: built beside the pre-pass (they need the probes only)
                    HIPCHECK(hipStreamWaitEvent(h->st2, h->ev_probe, 0));
                    launch_group_pairs(h->w_probelist.as<int32_t>(), pairs, nlist, 4 * ngq, cnt, cursor, pair_off, group_off, total_groups,
                                       pairs_sorted, h->d_len.as<int64_t>(), tile_rows, item_off, total_items, nprobe, 0, nprobe, 0,
                                       h->st2);
                    HIPCHECK(hipEventRecord(h->ev_group, h->st2));
                    grouped_early = true;
                }
                // large k: the histogram form of the four-query pre-pass (any sample size, several lists; pq_prepass4 = 2 keeps k_pq_prepass)
                const bool pre4big = rot && !pre4 && h->pq_prepass4 == 1 && nq >= 64 && h->Mpad >= 32 && pre_rows <= 32768;
                if (!(pre4 && launch_pq_prepass4(pa, nq, h->st) == 0) && !(pre4big && launch_pq_prepass4_big(pa, nq, h->st) == 0))
                    launch_pq_prepass(pa, nq, h->st);
                fused_pre_used = true;
                done = true;
                if (h->tc && h->tc->active && allow_fast && std::this_thread::get_id() == h->tc->worker) {
                    // two-call search: the thresholds are final on the device; hand them to the caller and wait for rsx_search_scan
                    HIPCHECK(hipStreamSynchronize(h->st));
                    rsx_index::TwoCall& t = *h->tc;
                    std::unique_lock<std::mutex> lk(t.mu);
                    t.tau = h->w_tau.as<uint64_t>(); t.ntau = nq; t.parked = true;
                    t.cv.notify_all();
                    t.cv.wait(lk, [&] { return t.go; });
                    t.parked = false; t.tau = nullptr; t.ntau = 0;
                }
            } else {
                build_tables(nullptr);
                h->w_temp.ensure((size_t)nq * tmax * 4);       // this form scores a prefix / everything into the score rows
                a.temp = h->w_temp.as<float>();
                launch_group_pairs(h->w_probelist.as<int32_t>(), pairs, nlist, 4, cnt, cursor, pair_off, group_off, total_groups,
                                   pairs_sorted, h->d_len.as<int64_t>(), pre_rows, item_off, total_items, nprobe, 0,
                                   filtered ? 1 : nprobe, filtered ? 1 : 0, h->st);
                tm.mark("group");
                const int64_t mi = filtered ? (nq + nlist + 8) : max_scan_items(h, nq, nprobe, 4, tile_rows);
                void* rws0 = rot ? rot_desc(mi, 1) : nullptr;
                done = (rot ? launch_pq_scan_rot(a, h->w_lut8.as<uint8_t>(), h->w_qparam.p, pairs_sorted, pair_off, group_off,
                                                 total_groups, item_off, total_items, nlist, mi, filtered ? pre_vpl : vpl,
                                                 nullptr, 0, nullptr, nullptr, 0, rws0, rot_log_cap, 0, 0, nullptr, nullptr, 0, h->st)
                            : launch_pq_scan8(a, h->w_lut8.as<uint8_t>(), h->w_qparam.p, pairs_sorted, pair_off, group_off,
                                              total_groups, item_off, total_items, nlist, mi, filtered ? pre_vpl : vpl, h->st)) == 0;
            }
            if (done && filtered) {
                tm.mark("scan0");
                cand_cap = (int)std::min<int64_t>(std::max<int64_t>(tmax, 1024), pq_cand_cap(k, h->M));
                h->w_cand.ensure((size_t)nq * cand_cap * 8);
                h->w_candcnt.ensure((size_t)nq * 8 * CCS);
                // multi-launch form: top-K' of the scored prefix of the closest list, row prefix
                // [0, min(seg_start[q][1], pre_rows)), written as the threshold key + counter reset
                if (!fused_pre)
                    select_rows(h, h->w_temp.as<float>(), tmax, h->w_segstart.as<int64_t>() + 1, nprobe + 1,
                                std::min<int64_t>(maxlen, pre_rows), 0, nq, KP, BUF, KP, state, false,
                                h->w_candcnt.as<unsigned long long>());
                // The pre-pass is only a threshold: keep its K'-th key and let the main scan score EVERYTHING (the
                // prefix included), so the scan kernel carries no per-slab "already scored" test and no key can
                // arrive twice (the prefix keys above the threshold come back through the candidate buffer).
                // (the selection wrote only the K'-th key of each query and reset the query's candidate counter)
                // (a stage mark is an event record: ~5 us of stream time in profile mode — none for a stage that launched nothing)
                if (!fused_pre) tm.mark("select0");
                if (grouped_early) { if (side_lut) HIPCHECK(hipStreamWaitEvent(h->st, h->ev_group, 0)); }
                else launch_group_pairs(h->w_probelist.as<int32_t>(), pairs, nlist, 4 * ngq, cnt, cursor, pair_off, group_off, total_groups,
                                        pairs_sorted, h->d_len.as<int64_t>(), tile_rows, item_off, total_items, nprobe, 0, nprobe, 0,
                                        h->st);
                if (!(grouped_early && !side_lut)) tm.mark("group");
                const int64_t mi_main = max_scan_items(h, nq, nprobe, 4 * ngq, tile_rows);
                void* rws1 = rot ? rot_desc(mi_main, ngq) : nullptr;
                // threshold keys: one per query from the one-launch pre-pass, else the K'-th key the selection left in the state rows
                const uint64_t* tau_ptr = fused_pre ? h->w_tau.as<uint64_t>() : state + (KP - 1);
                const int64_t tau_stride = fused_pre ? 1 : KP;
                // candidate gather + selection in one launch when the (probe rank, tile) table of a query is small (rsx_internal.h)
                gs_tmax = (int)((maxlen + tile_rows - 1) / tile_rows);
                use_gather = rot && h->pq_gather != 0 && pq_gather_select_applies(nprobe, gs_tmax, KP);
                if (rot && h->pq_gather != 0 && !use_gather) h->timing["pq_gather_declined"] += 1;
                if (use_gather) {
                    h->w_qitems.ensure((size_t)nq * nprobe * gs_tmax * 4);
                    gs.probe_list = h->w_probelist.as<int32_t>(); gs.list_len = h->d_len.as<int64_t>(); gs.nprobe = nprobe;
                    gs.tile_rows = tile_rows; gs.tmax = gs_tmax; gs.qitems = h->w_qitems.as<int32_t>();
                    gs.seg_desc = pq_scan_rot_ws_desc(rws1, mi_main * ngq); gs.log_keys = pq_scan_rot_ws_keys(rws1, mi_main * ngq);
                    gs.cand = h->w_cand.as<uint64_t>(); gs.cand_cnt = h->w_candcnt.as<unsigned long long>(); gs.cand_cap = cand_cap;
                    gs.state = state; gs.KP = KP;
                }
                tm.pre_scan();
                done = (rot ? launch_pq_scan_rot(a, h->w_lut8.as<uint8_t>(), h->w_qparam.p, pairs_sorted, pair_off, group_off,
                                                 total_groups, item_off, total_items, nlist,
                                                 mi_main, vpl, tau_ptr, tau_stride,
                                                 h->w_cand.as<uint64_t>(), h->w_candcnt.as<unsigned long long>(), cand_cap,
                                                 rws1, rot_log_cap, h->pq_prune, (h->pq_pace & 0xffff), (fused_pre && rot) ? h->w_excl.as<uint16_t>() : nullptr,
                                                 use_gather ? h->w_qitems.as<int32_t>() : nullptr, gs_tmax, h->st, sliced ? (sl_eight ? 1 : 0) : h->pq_q8)
                            : launch_pq_scan8_filter(a, h->w_lut8.as<uint8_t>(), h->w_qparam.p, pairs_sorted, pair_off, group_off,
                                                     total_groups, item_off, total_items, nlist,
                                                     max_scan_items(h, nq, nprobe, 4, tile_rows), vpl, tau_ptr, tau_stride,
                                                     h->w_cand.as<uint64_t>(), h->w_candcnt.as<unsigned long long>(), cand_cap,
                                                     h->st)) == 0;
            }
            if (!done) RSX_THROW(RSX_ERR_UNSUPPORTED, "IVFPQ fast scan: no kernel for M=%d", h->M);
        }
        if (!done && h->scan_kernel != 1 && h->CB == 16 && !pq_l2) {
            // v2: list-major, two queries per LDS read
            int64_t avg_slabs = std::max<int64_t>(1, (h->ntotal / std::max(1, nlist) + 63) / 64);
            int vpl = 8;
            if (h->scan_chunk > 0) vpl = std::max(1, std::min(8, h->scan_chunk / 1024));
            else while (vpl > 1 && (pairs / 2 + 1) * ((avg_slabs + 16 * vpl - 1) / (16 * vpl)) < 2048) vpl /= 2;
            if (vpl != 8 && vpl != 4 && vpl != 2) vpl = 1;
            const int tile_rows = 64 * 16 * vpl;
            h->w_pairs.ensure((size_t)(pairs + 5 * (size_t)(nlist + 1) + 8) * 4);
            int32_t* pairs_sorted = h->w_pairs.as<int32_t>();
            int32_t* cnt = pairs_sorted + pairs;
            int32_t* cursor = cnt + (nlist + 1);
            int32_t* pair_off = cursor + (nlist + 1);
            int32_t* group_off = pair_off + (nlist + 1);
            int32_t* item_off = group_off + (nlist + 1);
            int32_t* total_groups = item_off + (nlist + 1);
            int32_t* total_items = total_groups + 1;
            launch_group_pairs(h->w_probelist.as<int32_t>(), pairs, nlist, 2, cnt, cursor, pair_off, group_off, total_groups,
                               pairs_sorted, h->d_len.as<int64_t>(), tile_rows, item_off, total_items, nprobe, 0, nprobe, 0, h->st);
            tm.mark("group");
            done = launch_pq_scan2(a, pairs_sorted, pair_off, group_off, total_groups, item_off, total_items, nlist,
                                   max_scan_items(h, nq, nprobe, 2, tile_rows), vpl, h->st) == 0;
        }
        if (!done) {
            int64_t spc;
            if (h->scan_chunk > 0) spc = std::max<int64_t>(16, h->scan_chunk / 64);
            else {
                // enough work items to fill 256 CUs several times over, but no smaller than 32 slabs
                int64_t want_items = 4096;
                int64_t chunks = std::max<int64_t>(1, (want_items + pairs - 1) / pairs);
                spc = std::max<int64_t>(32, (max_slabs + chunks - 1) / chunks);
            }
            a.slabs_per_chunk = (int)spc;
            a.max_chunks = (int)((max_slabs + spc - 1) / spc);
            if ((pq_l2 ? launch_pq_scan_l2(a, h->w_q32.as<float>(), ld, h->d_centroids.as<float>(), h->d_codebooks.as<float>(), d, h->dsub, h->st)
                       : rot ? launch_pq_scan_rot_exact(a, h->st) : launch_pq_scan(a, h->st)) != 0)
                RSX_THROW(RSX_ERR_UNSUPPORTED, "IVFPQ scan: no kernel for M=%d", h->M);
        }
        h->timing[allow_fast ? "scan_launches" : "fb_scan_launches"] += 1;
        tm.mark("scan");
    } else if (!allow_fast) {
        // exact mode (see the Flat branch): fp64 scores of every row of the probed lists into the score rows
        ExactScoreArgs ea{};
        ea.kind = KIND_IVFFLAT; ea.metric = h->metric; ea.nq = nq; ea.Q32 = h->w_q32.as<float>(); ea.ldq = ld; ea.d = d;
        ea.X = h->data.p; ea.x_f16 = h->storage_f16; ea.ld = ld;
        ea.probe_list = h->w_probelist.as<int32_t>(); ea.seg_start = h->w_segstart.as<int64_t>(); ea.nprobe = nprobe;
        ea.list_base = h->d_base.as<int64_t>(); ea.list_len = h->d_len.as<int64_t>();
        ea.temp = h->w_temp.as<float>(); ea.tstride = tmax;
        launch_exact_scores(ea, h->st);
        tm.mark("scan");
    } else {
        // group (query, probe) pairs by list, then list-major MFMA scan
        int64_t npairs = nq * nprobe;
        h->w_pairs.ensure((size_t)(npairs + 5 * (size_t)(nlist + 1) + 8) * 4);
        int32_t* pairs_sorted = h->w_pairs.as<int32_t>();
        int32_t* cnt = pairs_sorted + npairs;
        int32_t* cursor = cnt + (nlist + 1);
        int32_t* pair_off = cursor + (nlist + 1);
        int32_t* group_off = pair_off + (nlist + 1);
        int32_t* item_off = group_off + (nlist + 1);
        int32_t* total_groups = item_off + (nlist + 1);
        int32_t* total_items = total_groups + 1;
        // probing queries per group of the LDS-DMA list scans.  A group passes over its list's rows once, so the larger the group the fewer
        // passes: 16 (k_list_scan2<_, 1>: two workgroups per CU), 64 (the 8-wave form, queries in LDS) or 128 (k_list_scan3, queries in
        // registers: d = 384 / 512 / 768 / 1024).  Round 5, 20M x 768, inner product, batch 1024 (profiles/r05_ivfflat_wide.md): at 8 probing queries per
        // list on average 16 wins (5.47 ms against 5.59 with 64), at 16 already 64 does (nlist 2048 / nprobe 32: 5.08 against 5.95;
        // nlist 4096 / nprobe 64: 6.93 against 8.53), at 32 64 beats 128 (5.20 against 5.61: no list needs a second pass yet, and the
        // 8-wave form has no barrier), at 64 and more 128 wins (6.04 against 7.23; nlist 1024 / nprobe 128: 10.6 against 11.4).  The
        // 32-query form lost everywhere and is only reachable through the parameter.
        int ls_qt = 1;
        if (h->scan_chunk <= 0 && h->ivf_qtiles != 0 && list_scan2_chunk_rows(h->storage_f16, ld) > 0) {
            const int64_t qpl = npairs / std::max(1, nlist);
            const bool ls3 = list_scan3_applies(h->storage_f16, ld, h->metric == RSX_METRIC_L2) != 0;      // 128 queries per group, held in registers (k_list_scan3)
            ls_qt = h->ivf_qtiles > 1 ? h->ivf_qtiles : (qpl >= 48 && ls3 ? 8 : qpl >= 12 ? 4 : 1);
            if (ls_qt == 8 && !ls3) ls_qt = 4;
            if (ls_qt != 8) ls_qt = std::min(ls_qt == 3 ? 2 : ls_qt, list_scan2_max_qtiles(ld));
            if (ls_qt != 2 && ls_qt != 4 && ls_qt != 8) ls_qt = 1;
        }
        // (list, chunk, group) work items in list-major order for the LDS-DMA scan's XCD-aware 1-D grid (round 4): the groups of a
        // list chunk run on one XCD at the same moment and its rows cross HBM once — at nlist 2048 / nprobe 128 half of the lists
        // are probed by more than 64 queries, i.e. by two groups, which used to land on different XCDs (two fetches)
        const int ls2_rows = list_scan2_chunk_rows(h->storage_f16, ld);
        const bool ls_wide = ls_qt >= 4;     // the 8-wave forms: 1024 rows per work item
        const int item_rows = (h->scan_chunk <= 0 && ls2_rows > 0) ? (ls_wide ? 2 * ls2_rows : ls2_rows) : 0;
        launch_group_pairs(h->w_probelist.as<int32_t>(), npairs, nlist, 16 * ls_qt, cnt, cursor, pair_off, group_off, total_groups,
                           pairs_sorted, item_rows ? h->d_len.as<int64_t>() : nullptr, item_rows, item_rows ? item_off : nullptr,
                           item_rows ? total_items : nullptr, nprobe, 0, nprobe, 0, h->st);
        tm.mark("group");
        const float* bias = nullptr;
        if (h->metric == RSX_METRIC_L2) { bias = h->w_misc.as<float>(); }
        ListScanArgs a{};
        a.Q16 = h->w_q16.as<__half>(); a.ld = ld; a.X = h->data.p; a.x_f16 = h->storage_f16; a.bias = bias;
        a.list_base = h->d_base.as<int64_t>(); a.list_len = h->d_len.as<int64_t>();
        a.pairs_sorted = pairs_sorted; a.pair_off = pair_off; a.group_off = group_off; a.total_groups = total_groups;
        a.probe_list = h->w_probelist.as<int32_t>(); a.seg_start = h->w_segstart.as<int64_t>();
        a.nlist = nlist; a.nprobe = nprobe; a.flat_mode = 0; a.nq = (int)nq;
        a.temp = h->w_temp.as<float>(); a.tstride = tmax;
        a.max_groups = (int)std::min<int64_t>(npairs, npairs / 16 + std::min<int64_t>(nlist, npairs));
        int64_t chunk_rows = h->scan_chunk > 0 ? round_up(h->scan_chunk, 64) : 2048;
        int64_t want = 2048;  // work items
        while (chunk_rows > 256 && (int64_t)a.max_groups * ((maxlen + chunk_rows - 1) / chunk_rows) < want) chunk_rows /= 2;
        if (h->scan_chunk <= 0 && list_scan2_chunk_rows(h->storage_f16, ld) > 0) chunk_rows = list_scan2_chunk_rows(h->storage_f16, ld);
        a.chunk_rows = (int)chunk_rows;
        a.qtiles = ls_qt;
        a.max_chunks = (int)std::max<int64_t>(1, (maxlen + chunk_rows - 1) / chunk_rows);
        // Same two-stage shape as the IVF-PQ fast path when the LDS-DMA kernel applies: score a prefix of every
        // query's closest list, take its K'-th key as the query's threshold, then scan everything with the keys
        // above it going to a small per-query candidate buffer instead of a full score row.  A full buffer
        // (never seen at the bench sizes) falls back to the score-buffer path, so the result is always exact.
        // (ivf_filter: 1 = when the score rows would exceed ~2 GB — below that the second grouping pass and the
        //  count read-back cost more than the row traffic they save; 2 = always; 0 = never)
        // (round 4: for large k the pre-pass scores the first 4 K' rows of the closest list — up to 32 chunks — instead of giving up
        //  the filter when K' no longer fits one chunk: k = 1000 at nlist 2048 / nprobe 128 wrote and re-read 10 GB of score rows)
        const int64_t pre_chunks = std::max<int64_t>(1, ((int64_t)KP * std::max(1, h->ivf_pre_mult) + chunk_rows - 1) / chunk_rows);
        bool want_filter = h->ivf_filter != 0 && nprobe > 1 && chunk_rows == list_scan2_chunk_rows(h->storage_f16, ld) &&
                           pre_chunks <= 32 && (h->ivf_filter > 1 || nq * tmax >= (int64_t)500000000);
        if (want_filter) {
            // ... of the closest list — of the EIGHT closest lists when K' is large: a query whose closest list holds fewer than K'
            // rows would get no threshold, keep every row of its 128 lists and overflow (the prefix of its score row then runs on
            // into the next lists' first rows)
            // (round 4: two lists, not eight — with 64 probing queries per list nearly every list is among some query's eight closest,
            //  and the 'sample' read 26 of the 31 GB: 3.4 + 0.8 ms of a 13.2 ms batch at nlist 2048 / nprobe 128 / k 1000; two lists
            //  leave 11.3 ms and as few candidates; ONE list overflows the queries whose closest list is short: profiles/r04_n_docs_1000.md)
            // ivf_pre_lists = 0 (default): the query's TWO closest lists.  ONE list is not enough even when it is long: at nlist 2048 /
            // nprobe 128 its K'-th key lets > 131072 keys of some queries through (the score-row pass follows: 27 instead of 11.7 ms); at
            // 100M / nprobe 32 it would do (28.7 against 29.8 ms) — two is the setting that is safe at both (profiles/r04_n_docs_1000.md).
            // (a per-query list count — two, and up to two more where the closest lists are short — cost 1-2 % on the bench configs and was
            //  removed in round 5: profiles/r04_n_docs_1000.md)
            const int pre_want = h->ivf_pre_lists > 0 ? h->ivf_pre_lists : 2;
            const int pre_lists = (KP >= 256 && (int64_t)pre_want * pre_chunks * chunk_rows <= tmax) ? std::min(pre_want, nprobe) : 1;
            a.max_chunks = (int)pre_chunks;                        // the first chunk(s) of ...
            a.qtiles = 1;                                          // (groups of 16 there: most lists are the closest of at most a few queries)
            const int64_t pre_stride = pre_lists > 1 ? pre_chunks * chunk_rows : 0;    // several lists: one slice of the sample buffer each
            if (pre_stride) {
                a.pre_stride = pre_stride; a.tstride = pre_lists * pre_stride;
                launch_fill_f32(h->w_temp.as<float>(), nq * a.tstride, -INFINITY, h->st);
            }
            launch_group_pairs(h->w_probelist.as<int32_t>(), npairs, nlist, 16, cnt, cursor, pair_off, group_off, total_groups,
                               pairs_sorted, nullptr, 0, nullptr, nullptr, nprobe, 0, pre_lists, 0, h->st);   // ... the closest list(s) only
            launch_list_scan(a, h->st);
            tm.mark("scan0");
            cand_cap = (int)std::min<int64_t>(std::max<int64_t>(tmax, 1024), k > 512 ? 131072 : (k > 64 ? 65536 : 16384));
            h->w_cand.ensure((size_t)nq * cand_cap * 8);
            h->w_candcnt.ensure((size_t)nq * 8 * CCS);
            // the pre-pass is only a threshold (see the IVF-PQ path): the K'-th key, candidate counters reset
            if (pre_stride) {
                select_rows(h, h->w_temp.as<float>(), a.tstride, nullptr, 0, a.tstride, 0,
                            nq, KP, BUF, KP, state, false, h->w_candcnt.as<unsigned long long>());
                a.pre_stride = 0; a.tstride = tmax;
            } else
            select_rows(h, h->w_temp.as<float>(), tmax, h->w_segstart.as<int64_t>() + 1, nprobe + 1,
                        std::min<int64_t>(maxlen, pre_chunks * chunk_rows), 0, nq, KP, BUF, KP, state, false,
                        h->w_candcnt.as<unsigned long long>());
            tm.mark("select0");
            launch_group_pairs(h->w_probelist.as<int32_t>(), npairs, nlist, 16 * ls_qt, cnt, cursor, pair_off, group_off, total_groups,
                               pairs_sorted, item_rows ? h->d_len.as<int64_t>() : nullptr, item_rows, item_rows ? item_off : nullptr,
                               item_rows ? total_items : nullptr, nprobe, 0, nprobe, 0, h->st);
            tm.mark("group");
            a.qtiles = ls_qt;
            if (ls_wide) { chunk_rows *= 2; a.chunk_rows = (int)chunk_rows; }      // 8 waves, 1024 rows per work item
            a.max_chunks = (int)std::max<int64_t>(1, (maxlen + chunk_rows - 1) / chunk_rows);
            if (item_rows == (int)chunk_rows) { a.item_off = item_off; a.total_items = total_items; a.max_items = (int)max_scan_items(h, nq, nprobe, 16 * ls_qt, item_rows); }
            a.tau_key = state + (KP - 1); a.tau_stride = KP;
            a.cand = h->w_cand.as<uint64_t>(); a.cand_cnt = h->w_candcnt.as<unsigned long long>(); a.cand_cap = cand_cap;
            launch_list_scan(a, h->st);
            tm.mark("scan");
            std::vector<unsigned long long> cnts((size_t)nq * CCS);
            HIPCHECK(hipMemcpyAsync(cnts.data(), h->w_candcnt.p, (size_t)nq * 8 * CCS, hipMemcpyDeviceToHost, h->st));
            HIPCHECK(hipStreamSynchronize(h->st));
            filtered = true;
            // A few overflowed rows (a query between clusters whose sample gave a weak threshold) no longer send the whole batch through the
            // unfiltered scan (10 -> 25 ms at nlist 2048 / nprobe 128 / k 1000 for ONE such query): with the certificate behind the search
            // k_finalize flags them (cand_cnt > cand_cap) and they alone join the exact re-run (~1.1 ms per query at that size with k_exact_scores_f16: up to nq / 128
            // of them are cheaper than the rescan); more overflows mean the threshold failed for this data, and the batch is rescanned
            int64_t n_over = 0;
            for (int64_t qi = 0; qi < nq; qi++) n_over += cnts[(size_t)qi * CCS] > (unsigned long long)cand_cap;
            if (n_over > 0 && !(certify && n_over <= (h->ivf_overflow_max > 0 ? (int64_t)h->ivf_overflow_max : std::max<int64_t>(2, nq / 128)))) filtered = false;
            if (filtered && n_over > 0) { flag_overflows = true; h->timing["ivf_filter_overflow_queries"] += (double)n_over; }
            a.tau_key = nullptr; a.cand = nullptr; a.cand_cnt = nullptr; a.cand_cap = 0;
        }
        if (!filtered) {
            if (ls_wide && a.chunk_rows == list_scan2_chunk_rows(h->storage_f16, ld)) {     // 8 waves, 1024 rows per work item
                a.chunk_rows *= 2;
                a.max_chunks = (int)std::max<int64_t>(1, (maxlen + a.chunk_rows - 1) / a.chunk_rows);
            }
            if (!want_filter && item_rows > 0 && item_rows == a.chunk_rows) {     // (after a filtered attempt the grouping in place is the filtered scan's: same items)
                a.item_off = item_off; a.total_items = total_items; a.max_items = (int)max_scan_items(h, nq, nprobe, 16 * ls_qt, item_rows);
            }
            launch_list_scan(a, h->st);
            tm.mark("scan");
        }
    }
    // IVF-PQ, rotated layout, threshold by construction: finalize straight from the complete candidate row (k_pq_final_tab) when K'
    // is large or a table entry is a long chain (M = 16: dsub 48) — the K' cut, its certificate and the second chance disappear
    const int tabP = (fast && filtered && fused_pre_used && rot && h->pq_final_tab != 0) ? pq_final_tab_capacity(h->M, h->CB, k) : 0;
    const bool use_tab = tabP > 0 && (h->pq_final_tab == 2 || KP >= 512 || h->dsub > 8);
    // 3. per-query k-selection over the score rows
    if (filtered && use_gather) {
        if (use_tab) gs.KP = 0;        // gather only
        launch_pq_gather_select(gs, nq, h->st);
    } else if (filtered && use_tab) {
        // the compaction has laid the survivors end to end in the candidate rows already
    } else if (filtered) {
        // merge the filtered candidates (keys) into state0: one wave per query, the whole buffer in one segment
        SelectArgs b{};
        b.in = h->w_cand.p; b.in_is_keys = 1; b.row_stride = cand_cap;
        b.row_n = reinterpret_cast<const int64_t*>(h->w_candcnt.p); b.row_n_stride = CCS; b.n_uniform = cand_cap;
        b.seg_len = round_up(cand_cap, 256); b.nseg = 1; b.idx_base = 0;
        b.init = state; b.out = state; b.out_row_stride = KP;
        b.nrows = nq; b.KP = KP; b.BUF = BUF; b.k = KP;
        launch_select(b, h->st);
    } else {
        // fast scan: the certificate needs the TRUE top-K' by approximate score, so the selection threshold
        // is the K'-th key, not the k-th
        select_rows(h, h->w_temp.as<float>(), tmax, h->w_segstart.as<int64_t>() + nprobe, nprobe + 1, tmax, 0, nq, KP, BUF,
                    (fast || h->kind == KIND_IVFFLAT) ? KP : k, state, false);
    }
    tm.mark("select");
    if (filtered && h->profile >= 2) {   // diagnostics: keys that passed the in-kernel filter
        std::vector<unsigned long long> cnts((size_t)nq * CCS);
        HIPCHECK(hipMemcpyAsync(cnts.data(), h->w_candcnt.p, (size_t)nq * 8 * CCS, hipMemcpyDeviceToHost, h->st));
        HIPCHECK(hipStreamSynchronize(h->st));
        double tot = 0, mx = 0;
        for (int64_t qi = 0; qi < nq; qi++) { const double c = (double)cnts[(size_t)qi * CCS]; tot += c; mx = std::max(mx, c); }
        h->timing["cand_keys"] += tot; h->timing["cand_keys_max"] = std::max(h->timing["cand_keys_max"], mx);
    }
    fa.probe_list = h->w_probelist.as<int32_t>(); fa.seg_start = h->w_segstart.as<int64_t>(); fa.nprobe = nprobe;
    if (flag_overflows) { fa.cand_cnt = h->w_candcnt.as<unsigned long long>(); fa.cand_cap = cand_cap; }     // IVF-Flat: certify_rows flags the overflowed rows' queries
    if (fast) {
        fa.pq_rescore = 1; fa.codes = h->data.as<uint8_t>(); fa.M = h->M; fa.Mpad = h->Mpad; fa.CB = h->CB;
        // large K' (the reference's n_docs = 100 ... 2000): thousands of candidates per query are re-scored, and in the scan layouts a candidate's
        // M code bytes are M / 16 pieces in as many 64-byte sectors — the finalize then runs at the HBM's random-sector rate.  A row-major copy
        // of the codes (pq_plain_stride(M) bytes per vector more — 128 at M = 96, so that a row is ONE 128-byte line: 288 GB of HBM hold it); built on the first such search of an
        // index state, dropped by the next add.  rsx_set_param "pq_plain_codes" = 0 turns it off.
        if (rot && KP >= 256 && h->pq_plain_codes != 0 && h->M >= 32 && h->M % 16 == 0 && h->M == h->Mpad) {
            if (h->plain_gen != h->dir_gen || h->plain_of != h->data.p) {
                const size_t need = (size_t)std::max<int64_t>(h->total_cap, 1) * (size_t)pq_plain_stride(h->M);
                size_t fr = 0, tot = 0;
                h->plain_of = nullptr;
                if (hipMemGetInfo(&fr, &tot) == hipSuccess && (h->codes_plain.bytes >= need || fr > need + need / 8 + ((size_t)8 << 30))) {
                    h->codes_plain.ensure(need);
                    launch_pq_plain_rows(h->data.as<uint8_t>(), h->total_cap, h->M, h->CB, h->codes_plain.as<uint8_t>(), h->st);
                    h->plain_gen = h->dir_gen; h->plain_of = h->data.p;
                    h->timing["plain_codes_builds"] += 1;
                }
            }
            if (h->plain_of == h->data.p && h->plain_gen == h->dir_gen) { fa.codes_plain = h->codes_plain.as<uint8_t>(); fa.plain_stride = pq_plain_stride(h->M); }
        }
        fa.lut32 = fused_lut ? nullptr : h->w_lut.as<float>(); fa.codebooks = h->d_codebooks.as<float>(); fa.dsub = h->dsub;
        fa.probe_dis0 = h->w_dis0.as<float>(); fa.qparam = h->w_qparam.p;
        fa.uncertain = h->w_uncertain.as<int32_t>();
        if (filtered) { fa.cand_cnt = h->w_candcnt.as<unsigned long long>(); fa.cand_cap = cand_cap; }
    }
    // (the tie scratch is allocated where k_pq_final_tab actually runs, not whenever it is a possible second chance)
    if (use_tab) h->w_tiews.ensure((size_t)nq * cand_cap * 8);
    if (use_tab) {
        FinalizeArgs ft = fa; if (lut32_out) ft.lut32 = lut32_out;
        if (h->profile >= 2) {      // diagnostics: how many candidates the 2 eps cut of k_pq_final_tab leaves to the exact re-score
            h->w_flag.ensure(16);
            HIPCHECK(hipMemsetAsync(h->w_flag.p, 0, 16, h->st));
            ft.stat = reinterpret_cast<unsigned long long*>(h->w_flag.p);
        }
        launch_pq_final_tab(ft, h->w_cand.as<uint64_t>(), cand_cap, h->w_tiews.as<uint64_t>(), h->st);
        if (h->profile >= 2) {
            unsigned long long v = 0;
            HIPCHECK(hipMemcpyAsync(&v, h->w_flag.p, 8, hipMemcpyDeviceToHost, h->st));
            HIPCHECK(hipStreamSynchronize(h->st));
            h->timing["final_tab_rescored"] += (double)v;
        }
    }
    else launch_finalize(fa, h->st);
    tm.mark("finalize");
    tm.finish();
    side_join.armed = false;      // st has waited on every side-stream event of this batch
    std::function<void()> second;
    if (fast && filtered && fused_pre_used && !use_tab && tabP > 0) {
        second = [&]() {       // the flagged queries' candidate rows are complete: settle them from there (k_pq_final_tab)
            h->timing["rescore_all_launches"] += 1;
            FinalizeArgs fr = fa;
            fr.row_filter = h->w_uncertain.as<int32_t>();
            h->w_tiews.ensure((size_t)nq * cand_cap * 8);
            launch_pq_final_tab(fr, h->w_cand.as<uint64_t>(), cand_cap, h->w_tiews.as<uint64_t>(), h->st);
        };
    } else if (fast && filtered && fused_pre_used && !use_tab) {
        second = [&]() {
            // the flagged queries' candidate rows are complete (threshold by construction) and did not overflow: score every
            // candidate exactly in place, then the best K2 >= k + 64 of them by (exact score, index) go through k_finalize for
            // the (score, id) order — no certificate needed, no exact scan
            h->timing["rescore_all_launches"] += 1;
            FinalizeArgs fr = fa;
            fr.row_filter = h->w_uncertain.as<int32_t>();
            launch_pq_rescore_all(fr, h->w_cand.as<uint64_t>(), cand_cap, h->st);
            const int KP2 = std::min(4096, std::max(128, pow2ceil(k + 64)));
            h->w_state2.ensure((size_t)nq * KP2 * 8);
            SelectArgs b{};
            b.in = h->w_cand.p; b.in_is_keys = 1; b.row_stride = cand_cap;
            b.row_n = reinterpret_cast<const int64_t*>(h->w_candcnt.p); b.row_n_stride = CCS; b.n_uniform = cand_cap;
            b.seg_len = round_up(cand_cap, 256); b.nseg = 1; b.idx_base = 0;
            b.init = nullptr; b.out = h->w_state2.as<uint64_t>(); b.out_row_stride = KP2;
            b.nrows = nq; b.KP = KP2; b.BUF = 2 * KP2; b.k = KP2;
            b.row_filter = h->w_uncertain.as<int32_t>();
            launch_select(b, h->st);
            FinalizeArgs f2 = fa;
            f2.state = h->w_state2.as<uint64_t>(); f2.KP = KP2; f2.row_filter = h->w_uncertain.as<int32_t>(); f2.no_cert = 1;
            launch_finalize(f2, h->st);
        };
    }
    if (fast || certify) rerun_uncertified(h, nq, dq, dtype, k, dD, dI, (size_t)tmax * 4, second);     // the re-run fills score rows: chunked by the budget
}

// L2 ranking bias  -|x|^2/2  from the stored squared norms
__global__ void k_bias_from_norms(const float* norms, float* bias, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bias[i] = -0.5f * norms[i];
}

void search_impl(rsx_index* h, int64_t nq, const void* q, int dtype, int k, float* D, int64_t* I) {
    if (nq < 0 || k <= 0) RSX_THROW(RSX_ERR_INVALID, "search: nq=%lld k=%d", (long long)nq, k);
    if (k > 4096) RSX_THROW(RSX_ERR_UNSUPPORTED, "search: k = %d exceeds this build's maximum of 4096 (the reference backends' default k)", k);
    if (!h->trained) RSX_THROW(RSX_ERR_NOT_TRAINED, "search before train");
    if (nq == 0) return;
    if (!q || !D || !I) RSX_THROW(RSX_ERR_INVALID, "search: null pointer");
    bool q_dev = is_device_ptr(q), o_dev = is_device_ptr(D);
    if (o_dev != is_device_ptr(I)) RSX_THROW(RSX_ERR_INVALID, "search: D and I must both be host or both device pointers");
    size_t esz = dtype == RSX_F16 ? 2 : 4;

    if (h->ntotal == 0) {  // FAISS returns -1 / -inf for an empty index
        std::vector<float> hd((size_t)nq * k, h->metric == 0 ? -INFINITY : INFINITY);
        std::vector<int64_t> hi((size_t)nq * k, -1);
        HIPCHECK(hipMemcpy(D, hd.data(), hd.size() * 4, o_dev ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
        HIPCHECK(hipMemcpy(I, hi.data(), hi.size() * 8, o_dev ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
        return;
    }
    if (h->metric == RSX_METRIC_L2 && h->kind != KIND_IVFPQ) {
        int64_t rows = h->total_cap;
        h->w_misc.ensure((size_t)rows * 4);
        hipLaunchKernelGGL(k_bias_from_norms, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, h->st, h->norms.as<float>(),
                           h->w_misc.as<float>(), rows);
    }
    // batch size: bounded by the knob and by the score-buffer budget
    int64_t qb = std::max(1, h->query_batch);
    if (h->kind != KIND_FLAT) {
        const int nprobe = std::min(h->nprobe, h->nlist);
        const int pad_to = (h->kind == KIND_IVFPQ) ? 64 : 16;
        int64_t tmax = top_probe_sum(h, nprobe, pad_to, pad_to).first;
        tmax = std::max<int64_t>(round_up(tmax, 256), 256);
        if (pq_search_needs_score_rows(h, nprobe, k)) qb = std::max<int64_t>(1, std::min<int64_t>(qb, h->temp_budget / (tmax * 4)));
        else qb = std::max<int64_t>(1, std::min<int64_t>(qb, h->temp_budget / (pq_cand_cap(k, h->M) * 16)));      // candidate row + tie scratch
    } else if (nq <= 32) {
        qb = 32;
    }
    for (int64_t q0 = 0; q0 < nq; q0 += qb) {
        int64_t nb = std::min(qb, nq - q0);
        const void* dq;
        const bool small = nb <= 64;      // latency path: stage through pinned memory (see PinBuf)
        if (q_dev) dq = (const char*)q + (size_t)q0 * h->d * esz;
        else {
            const size_t qbytes = (size_t)nb * h->d * esz;
            const char* src = (const char*)q + (size_t)q0 * h->d * esz;
            h->w_qin.ensure(qbytes);
            if (small && h->pin_q.ensure(qbytes)) { memcpy(h->pin_q.p, src, qbytes); src = h->pin_q.as<char>(); }
            HIPCHECK(hipMemcpyAsync(h->w_qin.p, src, qbytes, hipMemcpyHostToDevice, h->st));
            dq = h->w_qin.p;
        }
        float* dD; int64_t* dI;
        if (o_dev) { dD = D + q0 * k; dI = I + q0 * k; }
        else {
            h->w_D.ensure((size_t)nb * k * 4); h->w_I.ensure((size_t)nb * k * 8);
            dD = h->w_D.as<float>(); dI = h->w_I.as<int64_t>();
        }
        search_batch(h, nb, dq, dtype, k, dD, dI);
        const size_t dbytes = (size_t)nb * k * 4, ibytes = (size_t)nb * k * 8;
        const bool pin_out = !o_dev && small && h->pin_out.ensure(round_up(dbytes, 16) + ibytes);
        if (pin_out) {
            HIPCHECK(hipMemcpyAsync(h->pin_out.p, dD, dbytes, hipMemcpyDeviceToHost, h->st));
            HIPCHECK(hipMemcpyAsync(h->pin_out.as<char>() + round_up(dbytes, 16), dI, ibytes, hipMemcpyDeviceToHost, h->st));
        } else if (!o_dev) {
            HIPCHECK(hipMemcpyAsync(D + q0 * k, dD, dbytes, hipMemcpyDeviceToHost, h->st));
            HIPCHECK(hipMemcpyAsync(I + q0 * k, dI, ibytes, hipMemcpyDeviceToHost, h->st));
        }
        HIPCHECK(hipStreamSynchronize(h->st));
        if (pin_out) {
            memcpy(D + q0 * k, h->pin_out.p, dbytes);
            memcpy(I + q0 * k, h->pin_out.as<char>() + round_up(dbytes, 16), ibytes);
        }
    }
    HIPCHECK(hipGetLastError());
}

