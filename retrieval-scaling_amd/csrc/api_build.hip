// api_build.hip — HBM layout management, add / train, list import / export, persistence (host control plane of librsx;
// shared declarations: rsx_host.h).
#include "rsx_host.h"

void use_device(rsx_index* h) { HIPCHECK(hipSetDevice(h->device)); }
void ensure_side_stream(rsx_index* h) {
    if (h->st2) return;
    HIPCHECK(hipStreamCreateWithFlags(&h->st2, hipStreamNonBlocking));
    for (hipEvent_t* e : {&h->ev_fork, &h->ev_probe, &h->ev_lut, &h->ev_group}) HIPCHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
}
// A two-call search that is parked between rsx_search_prepass and rsx_search_scan owns the handle's workspaces (thresholds,
// candidate rows, state): every other entry point refuses to touch the handle until rsx_search_scan has finished it.
void refuse_while_two_call(const rsx_index* h, const char* what) {
    if (h && h->tc && h->tc->active)
        RSX_THROW(RSX_ERR_INVALID, "%s: a two-call search is open on this handle (finish it with rsx_search_scan first)", what);
}

// HBM held by the search / add workspaces of one (unsharded) handle — grows with the largest batch served so far, is never
// part of the index payload (hbm_bytes) and is released with the handle.  The largest single item is the IVF-PQ fast scan's
// per-item survivor segments (w_itemdesc: a few GB at the bench configuration, rsx_internal.h: pq_scan_rot_ws).
int64_t workspace_bytes(const rsx_index* h) {
    const DevBuf* bufs[] = {&h->w_q32, &h->w_q16, &h->w_coarse, &h->w_keys1, &h->w_probekeys, &h->w_probelist, &h->w_dis0, &h->w_segstart,
                            &h->w_temp, &h->w_lut, &h->w_lutws, &h->w_state, &h->w_D, &h->w_I, &h->w_qin, &h->w_pairs, &h->w_flag, &h->w_x,
                            &h->w_partial, &h->w_assign, &h->w_dest, &h->w_idsin, &h->w_misc, &h->w_lut8, &h->w_qparam, &h->w_uncertain,
                            &h->w_fbq, &h->w_fbD, &h->w_fbI, &h->w_cand, &h->w_candcnt, &h->w_itemdesc, &h->w_tau, &h->w_excl, &h->w_state2, &h->w_addcnt, &h->w_addstart, &h->w_qitems, &h->w_tiews, &h->sh_D, &h->sh_I, &h->sh_q,
                            &h->sh_oD, &h->sh_oI, &h->codes_plain /* derived from the payload on demand, dropped by the next add */};
    int64_t t = 0;
    for (const DevBuf* b : bufs) t += (int64_t)b->bytes;
    return t;
}

void upload_dir(rsx_index* h) {
    h->dir_gen++;       // list lengths changed: the memoised host-side bounds below are stale
    size_t nb = (size_t)h->nlist * sizeof(int64_t);
    h->d_base.ensure(nb);
    h->d_len.ensure(nb);
    HIPCHECK(hipMemcpyAsync(h->d_base.p, h->h_base.data(), nb, hipMemcpyHostToDevice, h->st));
    HIPCHECK(hipMemcpyAsync(h->d_len.p, h->h_len.data(), nb, hipMemcpyHostToDevice, h->st));
    HIPCHECK(hipStreamSynchronize(h->st));
}

// Make every list able to hold need[l] rows; re-lays-out HBM when a list overflows.
void ensure_capacity(rsx_index* h, const std::vector<int64_t>& need, bool exact) {
    const int al = h->row_align();
    bool grow = false;
    for (int l = 0; l < h->nlist; l++)
        if (need[(size_t)l] > h->h_cap[(size_t)l]) { grow = true; break; }
    bool need_ids = (h->kind != KIND_FLAT) || h->custom_ids;
    bool need_norms = (h->metric == RSX_METRIC_L2) && h->kind != KIND_IVFPQ;
    if (!grow && h->data.p && (!need_ids || h->ids.p) && (!need_norms || h->norms.p)) return;

    // Amortised growth: a re-layout moves the whole index, so when ANY list overflows EVERY list gets
    // headroom proportional to its current need (2x for PQ codes, 1.5x for raw rows); the number of
    // re-layouts is then logarithmic in the final size instead of one per add batch.
    std::vector<int64_t> ncap(h->h_cap), nbase((size_t)h->nlist);
    for (int l = 0; l < h->nlist; l++) {
        int64_t nd = need[(size_t)l];
        int64_t want = exact ? nd : (h->kind == KIND_IVFPQ ? 2 * nd + 64 : nd + nd / 2);
        if (grow && round_up(want, al) > ncap[(size_t)l]) ncap[(size_t)l] = round_up(want, al);
        if (nd > ncap[(size_t)l]) ncap[(size_t)l] = round_up(nd, al);
        if (ncap[(size_t)l] == 0 && h->kind == KIND_FLAT) ncap[(size_t)l] = al;
    }
    int64_t tot = 0;
    for (int l = 0; l < h->nlist; l++) { nbase[(size_t)l] = tot; tot += ncap[(size_t)l]; }
    if (tot >= ((int64_t)1 << 32)) RSX_THROW(RSX_ERR_UNSUPPORTED, "more than 2^32 storage rows on one device");
    size_t rb = h->row_bytes();
    void* ndata = nullptr; void* nids = nullptr; void* nnorms = nullptr;
    size_t dbytes = (size_t)std::max<int64_t>(tot, al) * rb;
    HIPCHECK(hipMalloc(&ndata, dbytes));
    HIPCHECK(hipMemsetAsync(ndata, 0, dbytes, h->st));
    if (need_ids) { HIPCHECK(hipMalloc(&nids, (size_t)std::max<int64_t>(tot, 1) * 8)); }
    if (need_norms) {
        HIPCHECK(hipMalloc(&nnorms, (size_t)std::max<int64_t>(tot, 1) * 4));
        HIPCHECK(hipMemsetAsync(nnorms, 0, (size_t)std::max<int64_t>(tot, 1) * 4, h->st));
    }
    if (h->data.p && h->ntotal > 0) {
        DevBuf ob, nb2;
        size_t nb = (size_t)h->nlist * 8;
        ob.ensure(nb); nb2.ensure(nb);
        HIPCHECK(hipMemcpyAsync(ob.p, h->h_base.data(), nb, hipMemcpyHostToDevice, h->st));
        HIPCHECK(hipMemcpyAsync(nb2.p, nbase.data(), nb, hipMemcpyHostToDevice, h->st));
        h->d_len.ensure(nb);
        HIPCHECK(hipMemcpyAsync(h->d_len.p, h->h_len.data(), nb, hipMemcpyHostToDevice, h->st));
        int64_t unit_rows = (h->kind == KIND_IVFPQ) ? h->row_align() : 1;      // PQ codes move in whole slabs / slice-major groups
        int64_t unit_bytes = (int64_t)rb * unit_rows;
        launch_copy_lists(h->nlist, ob.as<int64_t>(), nb2.as<int64_t>(), h->d_len.as<int64_t>(), h->data.as<uint8_t>(),
                          (uint8_t*)ndata, unit_rows, unit_bytes, (need_ids && h->ids.p) ? h->ids.as<int64_t>() : nullptr,
                          (int64_t*)nids, (need_norms && h->norms.p) ? h->norms.as<float>() : nullptr, (float*)nnorms,
                          h->st);
        HIPCHECK(hipStreamSynchronize(h->st));
    }
    h->data.release(); h->data.p = ndata; h->data.bytes = dbytes;
    if (need_ids) { h->ids.release(); h->ids.p = nids; h->ids.bytes = (size_t)std::max<int64_t>(tot, 1) * 8; }
    if (need_norms) { h->norms.release(); h->norms.p = nnorms; h->norms.bytes = (size_t)std::max<int64_t>(tot, 1) * 4; }
    h->h_cap = ncap; h->h_base = nbase; h->total_cap = tot;
    upload_dir(h);
}

// Flat / IVF-Flat keep fp16 rows while every value ever added is fp16-representable, else fp32.
static void upgrade_storage_to_f32(rsx_index* h) {
    if (!h->storage_f16) return;
    if (h->data.p && h->total_cap > 0) {
        void* nd = nullptr;
        size_t nbytes = (size_t)std::max<int64_t>(h->total_cap, h->row_align()) * h->ld * 4;
        HIPCHECK(hipMalloc(&nd, nbytes));
        launch_widen_storage(h->data.as<__half>(), (float*)nd, (int64_t)std::max<int64_t>(h->total_cap, h->row_align()) * h->ld, h->st);
        HIPCHECK(hipStreamSynchronize(h->st));
        h->data.release(); h->data.p = nd; h->data.bytes = nbytes;
    }
    h->storage_f16 = 0;
}

// Stage `n` rows of caller data on the device; returns the device pointer (caller's own if it
// already lives in HBM).
const void* stage_rows(rsx_index* h, DevBuf& buf, const void* x, int64_t n, int d, int dtype) {
    size_t bytes = (size_t)n * d * (dtype == RSX_F16 ? 2 : 4);
    if (is_device_ptr(x)) return x;
    buf.ensure(bytes);
    HIPCHECK(hipMemcpyAsync(buf.p, x, bytes, hipMemcpyHostToDevice, h->st));
    return buf.p;
}

void set_centroids(rsx_index* h, const float* c) {
    size_t n = (size_t)h->nlist * h->d;
    h->h_centroids.assign(c, c + n);
    h->cent_gen++;
    h->d_centroids.ensure(n * 4);
    HIPCHECK(hipMemcpyAsync(h->d_centroids.p, h->h_centroids.data(), n * 4, hipMemcpyHostToDevice, h->st));
    HIPCHECK(hipStreamSynchronize(h->st));
}
void set_codebooks(rsx_index* h, const float* c) {
    size_t n = (size_t)h->M * 256 * h->dsub;
    h->h_codebooks.assign(c, c + n);
    h->d_codebooks.ensure(n * 4);
    HIPCHECK(hipMemcpyAsync(h->d_codebooks.p, h->h_codebooks.data(), n * 4, hipMemcpyHostToDevice, h->st));
    HIPCHECK(hipStreamSynchronize(h->st));
}
void update_trained(rsx_index* h) {
    if (h->kind == KIND_FLAT) h->trained = true;
    else if (h->kind == KIND_IVFFLAT) h->trained = !h->h_centroids.empty();
    else h->trained = !h->h_centroids.empty() && !h->h_codebooks.empty();
}

// ---------------------------------------------------------------------------------------
// creation
// ---------------------------------------------------------------------------------------
rsx_index* create_common(int kind, int d, int nlist, int M, int nbits, int metric, int device) {
    if (d <= 0) RSX_THROW(RSX_ERR_INVALID, "d must be positive (got %d)", d);
    if (metric != RSX_METRIC_INNER_PRODUCT && metric != RSX_METRIC_L2) RSX_THROW(RSX_ERR_INVALID, "unknown metric %d", metric);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        RSX_THROW(RSX_ERR_HIP, "no HIP device available: librsx has no CPU path");
    }
    if (device < 0 || device >= ndev) RSX_THROW(RSX_ERR_INVALID, "device %d out of range (have %d)", device, ndev);
    std::unique_ptr<rsx_index> h(new rsx_index());
    h->kind = kind; h->d = d; h->metric = metric; h->device = device;
#ifndef RSX_FLAGS_IN_HBM
    h->w_uncertain.host_mapped = true;
#endif
    h->nlist = (kind == KIND_FLAT) ? 1 : nlist;
    if (kind != KIND_FLAT && nlist <= 0) RSX_THROW(RSX_ERR_INVALID, "nlist must be positive (got %d)", nlist);
    h->ld = (int)round_up(d, 64);
    if (kind == KIND_IVFPQ) {
        if (nbits != 8) RSX_THROW(RSX_ERR_UNSUPPORTED, "IVFPQ: only nbits = 8 is implemented (got %d)", nbits);
        if (M <= 0 || d % M != 0) RSX_THROW(RSX_ERR_INVALID, "IVFPQ: d (%d) must be a multiple of M (%d)", d, M);
        // METRIC_INNER_PRODUCT is what the reference builds (src/indicies/ivf_pq.py:147-153) and what the fast scans serve; METRIC_L2 — squared
        // distance to the decoded vector over the same inner-product coarse quantiser — is served by the exact per-(query, list) scan (round 6)
        h->M = M; h->nbits = nbits; h->dsub = d / M;
        h->Mpad = (int)round_up(M, 4);
        h->CB = (h->Mpad % 16 == 0) ? 16 : 4;
        if (h->CB == 16) {
            int nch = h->Mpad / 16;
            if (!(nch == 1 || nch == 2 || nch == 3 || nch == 4 || nch == 6 || nch == 8)) h->CB = 4;
        }
        // M in {16, 32, 64, 96, 128}: the rotated layout (conflict-free table gathers, k_pq_rot.hip) unless RSX_PQ_LAYOUT=0;
        // rsx_set_param "pq_layout" switches an EMPTY index between the two.  (M = 16 — the reference's shipped IVF-PQ config,
        // ric/conf/ivf_pq.yaml:64-78 — joined in round 4, once the survivors went to per-wave logs instead of fixed segments.)
        h->CB_granule = h->CB;
        // M = 96 (round 6): the SLICED layout (PQ_SLICED: 32-vector blocks cut into 32-sub-quantiser slices) — its scan looks eight
        // queries up per table gather, which at M = 96 needs the table of one slice at a time (k_pq_scan_sl8).  RSX_PQ_LAYOUT = 0 / 1 / 2
        // forces granule / rotated / sliced where they apply; "pq_layout" switches an empty index.
        const char* e = getenv("RSX_PQ_LAYOUT");
        const int want = e ? atoi(e) : PQ_LAYOUT_DEFAULT;
        if (pq_rot_applies(M) && want != 0) h->CB = 0;
        if (pq_sliced_applies(M) && want == 2) h->CB = PQ_SLICED;
        if ((size_t)h->Mpad * 1024 > 160 * 1024) RSX_THROW(RSX_ERR_UNSUPPORTED, "IVFPQ: M = %d needs more than 160 KiB of LDS for the look-up table", M);
    }
    HIPCHECK(hipSetDevice(device));
    HIPCHECK(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
    h->h_base.assign((size_t)h->nlist, 0);
    h->h_len.assign((size_t)h->nlist, 0);
    h->h_cap.assign((size_t)h->nlist, 0);
    update_trained(h.get());
    return h.release();
}

// ---------------------------------------------------------------------------------------
// add
// ---------------------------------------------------------------------------------------
static void decide_storage(rsx_index* h, const void* dx, int64_t n, int dtype) {
    if (h->kind == KIND_IVFPQ) return;
    if (dtype == RSX_F16) { h->storage_decided = true; return; }
    if (h->storage_decided && !h->storage_f16) return;
    h->w_flag.ensure(sizeof(int));
    HIPCHECK(hipMemsetAsync(h->w_flag.p, 0, sizeof(int), h->st));
    launch_check_f16((const float*)dx, n * h->d, h->w_flag.as<int>(), h->st);
    int flag = 0;
    HIPCHECK(hipMemcpyAsync(&flag, h->w_flag.p, sizeof(int), hipMemcpyDeviceToHost, h->st));
    HIPCHECK(hipStreamSynchronize(h->st));
    if (flag) upgrade_storage_to_f32(h);
    h->storage_decided = true;
}

// fold the batch's largest |x|^2 into h->max_norm2 (read back here: every add path synchronises the stream anyway)
static void track_max_norm(rsx_index* h, const void* dx, int64_t n, int dtype) {
    if (!h->d_maxnorm.p) {
        h->d_maxnorm.ensure(sizeof(unsigned int));
        HIPCHECK(hipMemsetAsync(h->d_maxnorm.p, 0, sizeof(unsigned int), h->st));
    }
    launch_max_norm2(dx, dtype == RSX_F16, n, h->d, h->d_maxnorm.as<unsigned int>(), h->st);
    HIPCHECK(hipMemcpyAsync(&h->max_norm2, h->d_maxnorm.p, sizeof(float), hipMemcpyDeviceToHost, h->st));
    HIPCHECK(hipStreamSynchronize(h->st));
}

static void add_batch(rsx_index* h, int64_t n, const void* x, int dtype, const int64_t* ids) {
    const void* dx = stage_rows(h, h->w_x, x, n, h->d, dtype);
    const int64_t* dids = nullptr;
    if (ids) {
        if (is_device_ptr(ids)) dids = ids;
        else {
            h->w_idsin.ensure((size_t)n * 8);
            HIPCHECK(hipMemcpyAsync(h->w_idsin.p, ids, (size_t)n * 8, hipMemcpyHostToDevice, h->st));
            dids = h->w_idsin.as<int64_t>();
        }
    }
    decide_storage(h, dx, n, dtype);
    if (h->kind != KIND_IVFPQ) track_max_norm(h, dx, n, dtype);

    if (h->kind == KIND_FLAT) {
        if (ids && !h->custom_ids) {
            if (h->ntotal > 0) RSX_THROW(RSX_ERR_UNSUPPORTED, "Flat: cannot switch to explicit ids after sequential adds");
            h->custom_ids = true;
        } else if (!ids && h->custom_ids) {
            RSX_THROW(RSX_ERR_INVALID, "Flat: index was populated with explicit ids; ids required");
        }
        std::vector<int64_t> need(1, h->ntotal + n);
        ensure_capacity(h, need, false);
        size_t esz = h->storage_f16 ? 2 : 4;
        launch_scatter_rows(dx, dtype == RSX_F16, n, h->d, nullptr, h->data.as<uint8_t>() + (size_t)h->ntotal * h->ld * esz,
                            h->storage_f16, h->ld, h->norms.p ? h->norms.as<float>() + h->ntotal : nullptr, dids, 0,
                            h->custom_ids ? h->ids.as<int64_t>() + h->ntotal : nullptr, h->st);
        h->h_len[0] = h->ntotal + n;
        h->ntotal += n;
        upload_dir(h);
        return;
    }

    // IVF: assignment on the matrix cores (exact fp32 chain), placement on the host
    int ct = (h->nlist + 127) / 128;
    h->w_partial.ensure((size_t)n * 2 * ct * 8);
    h->w_assign.ensure((size_t)n * 4);
    launch_gemm_exact_argmax(dx, dtype == RSX_F16, n, h->d, h->d_centroids.as<float>(), h->nlist, h->d,
                             h->w_partial.as<uint64_t>(), h->w_assign.as<int32_t>(), nullptr, h->st);
    // placement on the device (round 3): only the per-list totals of the batch visit the host (4 bytes per list — it has to
    // grow the lists), not the assignments (4 bytes per vector out, 8 back): stable ranks = insertion order inside a list.
    // List-sharded multi-GPU index: this handle keeps only the lists l with l % add_list_mod == add_list_rem; the other
    // vectors of the stream are assigned (they advance the sequential ids) and dropped.
    const int lmod = std::max(1, h->add_list_mod), lrem = h->add_list_rem;
    const int64_t nseg = add_dest_segments(n);
    std::vector<int64_t> need(h->h_len);
    int64_t nkept = 0;
    h->w_dest.ensure((size_t)n * 8);
    if ((size_t)h->nlist * 4 > 60 * 1024) {
        // more lists than the placement kernels' LDS table holds (15360): the round-2 host placement
        std::vector<int32_t> assign((size_t)n);
        HIPCHECK(hipMemcpyAsync(assign.data(), h->w_assign.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->st));
        HIPCHECK(hipStreamSynchronize(h->st));
        std::vector<int64_t> pos((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            const int32_t l = assign[(size_t)i];
            if (lmod > 1 && l % lmod != lrem) { pos[(size_t)i] = -1; continue; }
            pos[(size_t)i] = need[(size_t)l]++;
            nkept++;
        }
        ensure_capacity(h, need, false);
        for (int64_t i = 0; i < n; i++) if (pos[(size_t)i] >= 0) pos[(size_t)i] += h->h_base[(size_t)assign[(size_t)i]];
        HIPCHECK(hipMemcpyAsync(h->w_dest.p, pos.data(), (size_t)n * 8, hipMemcpyHostToDevice, h->st));
        HIPCHECK(hipStreamSynchronize(h->st));   // pos is a local
    } else {
    h->w_addcnt.ensure((size_t)(nseg + 1) * h->nlist * 4);
    int32_t* seg_cnt = h->w_addcnt.as<int32_t>();
    int32_t* d_total = seg_cnt + (size_t)nseg * h->nlist;
    launch_add_destinations(h->w_assign.as<int32_t>(), n, h->nlist, lmod, lrem, seg_cnt, d_total, h->st);
    std::vector<int32_t> total((size_t)h->nlist);
    HIPCHECK(hipMemcpyAsync(total.data(), d_total, (size_t)h->nlist * 4, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(hipStreamSynchronize(h->st));
    for (int l = 0; l < h->nlist; l++) { need[(size_t)l] += total[(size_t)l]; nkept += total[(size_t)l]; }
    ensure_capacity(h, need, false);
    std::vector<int64_t> start((size_t)h->nlist);
    for (int l = 0; l < h->nlist; l++) start[(size_t)l] = h->h_base[(size_t)l] + h->h_len[(size_t)l];
    h->w_addstart.ensure((size_t)h->nlist * 8);
    HIPCHECK(hipMemcpyAsync(h->w_addstart.p, start.data(), (size_t)h->nlist * 8, hipMemcpyHostToDevice, h->st));
    launch_add_place(h->w_assign.as<int32_t>(), n, h->nlist, lmod, lrem, seg_cnt, h->w_addstart.as<int64_t>(), h->w_dest.as<int64_t>(), h->st);
    }

    if (h->kind == KIND_IVFPQ) {
        launch_pq_encode(dx, dtype == RSX_F16, n, h->d, h->d, h->M, h->Mpad, h->CB, h->d_centroids.as<float>(),
                         h->w_assign.as<int32_t>(), h->d_codebooks.as<float>(), h->w_dest.as<int64_t>(),
                         h->data.as<uint8_t>(), nullptr, h->st);
        launch_write_ids(h->w_dest.as<int64_t>(), dids, h->ntotal + h->ndropped, n, h->ids.as<int64_t>(), h->st);
    } else {
        launch_scatter_rows(dx, dtype == RSX_F16, n, h->d, h->w_dest.as<int64_t>(), h->data.p, h->storage_f16, h->ld,
                            h->norms.p ? h->norms.as<float>() : nullptr, dids, h->ntotal + h->ndropped, h->ids.as<int64_t>(), h->st);
    }
    HIPCHECK(hipStreamSynchronize(h->st));  // pos / staging buffers are reused by the next batch
    h->h_len = need;
    h->ntotal += nkept;
    h->ndropped += n - nkept;               // sequential ids count every vector of the add stream
    upload_dir(h);
}

void add_all(rsx_index* h, int64_t n, const void* x, int dtype, const int64_t* ids) {
    const int64_t B = 262144;  // rows per internal batch
    size_t esz = dtype == RSX_F16 ? 2 : 4;
    for (int64_t i0 = 0; i0 < n; i0 += B) {
        int64_t nb = std::min(B, n - i0);
        add_batch(h, nb, (const char*)x + (size_t)i0 * h->d * esz, dtype, ids ? ids + i0 : nullptr);
    }
    HIPCHECK(hipStreamSynchronize(h->st));
}

// ---------------------------------------------------------------------------------------
// training (faiss::Clustering / ProductQuantizer::train restated; assignment on the GPU,
// centroid update on the host in point order so that training is deterministic)
// ---------------------------------------------------------------------------------------
static void renorm_rows(int d, int k, float* c) {
    for (int i = 0; i < k; i++) {
        float nr = 0.0f;
        float* r = c + (size_t)i * d;
        for (int t = 0; t < d; t++) nr = fmaf(r[t], r[t], nr);
        if (nr > 0.0f) {
            float inv = 1.0f / sqrtf(nr);
            for (int t = 0; t < d; t++) r[t] *= inv;
        }
    }
}

// One Lloyd update given assignments.  The accumulation (sum of the assigned points, point order, fp32) runs on the GPU
// (k_kmeans_accumulate: one sequential chain per (centroid, dimension), all of them in flight); the host groups the points
// by centroid (a stable counting sort of the assignment vector) and finishes the update on the k x d sums: division by the
// counts, FAISS's empty-cluster split, both in the order the oracle uses.
//   dx: training points on the device [n, ldx]; nsets sub-spaces at column offsets s * col_stride, each of width d, with its
//   own assignment vector assign[s * astride_set + i * astride_pt]; cen: [nsets][k][d] on the host.
struct KmeansWs { DevBuf order, off, sums; std::vector<int32_t> h_order, h_off; std::vector<float> h_sums; };

static void kmeans_update_gpu(rsx_index* h, KmeansWs& ws, const float* dx, int64_t ldx, int col_stride, int d, int k, int nsets,
                              int64_t n, const int32_t* assign, int64_t astride_set, int astride_pt, float* cen) {
    ws.h_order.resize((size_t)nsets * n); ws.h_off.resize((size_t)nsets * (k + 1));
    for (int s = 0; s < nsets; s++) {
        int32_t* off = &ws.h_off[(size_t)s * (k + 1)];
        std::fill(off, off + k + 1, 0);
        const int32_t* as = assign + (size_t)s * astride_set;
        for (int64_t i = 0; i < n; i++) off[as[(size_t)i * astride_pt] + 1]++;
        for (int c = 0; c < k; c++) off[c + 1] += off[c];
        std::vector<int32_t> cur(off, off + k);
        int32_t* ord = &ws.h_order[(size_t)s * n];
        for (int64_t i = 0; i < n; i++) ord[cur[(size_t)as[(size_t)i * astride_pt]]++] = (int32_t)i;   // stable: point order kept
    }
    ws.order.ensure(ws.h_order.size() * 4); ws.off.ensure(ws.h_off.size() * 4); ws.sums.ensure((size_t)nsets * k * d * 4);
    HIPCHECK(hipMemcpyAsync(ws.order.p, ws.h_order.data(), ws.h_order.size() * 4, hipMemcpyHostToDevice, h->st));
    HIPCHECK(hipMemcpyAsync(ws.off.p, ws.h_off.data(), ws.h_off.size() * 4, hipMemcpyHostToDevice, h->st));
    launch_kmeans_accumulate(dx, ldx, col_stride, d, k, nsets, n, ws.order.as<int32_t>(), ws.off.as<int32_t>(), ws.sums.as<float>(), h->st);
    HIPCHECK(hipMemcpyAsync(cen, ws.sums.p, (size_t)nsets * k * d * 4, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(hipStreamSynchronize(h->st));
    for (int s = 0; s < nsets; s++) {
        const int32_t* off = &ws.h_off[(size_t)s * (k + 1)];
        float* cs = cen + (size_t)s * k * d;
        std::vector<int64_t> hassign((size_t)k);
        for (int j = 0; j < k; j++) hassign[(size_t)j] = off[j + 1] - off[j];
        for (int j = 0; j < k; j++) {
            if (hassign[(size_t)j] == 0) continue;
            const float norm = 1.0f / (float)hassign[(size_t)j];
            float* cc = cs + (size_t)j * d;
            for (int t = 0; t < d; t++) cc[t] *= norm;
        }
        uint64_t rs = 1234;
        for (int ci = 0; ci < k; ci++) {
            if (hassign[(size_t)ci] != 0) continue;
            int cj = 0;
            for (;;) {
                double p = ((double)hassign[(size_t)cj] - 1.0) / (double)(n - k);
                double r = (double)(splitmix(rs) >> 11) * (1.0 / 9007199254740992.0);
                if (r < p) break;
                cj = (cj + 1) % k;
            }
            float* a_ = cs + (size_t)ci * d; float* b_ = cs + (size_t)cj * d;
            memcpy(a_, b_, sizeof(float) * (size_t)d);
            for (int t = 0; t < d; t++) {
                if (t % 2 == 0) { a_[t] *= 1.0f + 1.0f / 1024.0f; b_[t] *= 1.0f - 1.0f / 1024.0f; }
                else { a_[t] *= 1.0f - 1.0f / 1024.0f; b_[t] *= 1.0f + 1.0f / 1024.0f; }
            }
            hassign[(size_t)ci] = hassign[(size_t)cj] / 2;
            hassign[(size_t)cj] -= hassign[(size_t)ci];
        }
    }
}

void train_impl(rsx_index* h, int64_t n, const void* x, int dtype) {
    if (h->kind == KIND_FLAT) return;
    if (n < h->nlist) RSX_THROW(RSX_ERR_INVALID, "train: %lld training points for %d centroids", (long long)n, h->nlist);
    const int d = h->d;
    // training set as fp32 on the host (the reference passes host numpy: ivf_flat.py:135)
    std::vector<float> hx((size_t)n * d);
    {
        const void* dx = stage_rows(h, h->w_x, x, n, d, dtype);
        DevBuf t32; t32.ensure((size_t)n * d * 4);
        launch_convert_to_f32(dx, dtype == RSX_F16, d, n, d, t32.as<float>(), d, h->st);
        HIPCHECK(hipMemcpyAsync(hx.data(), t32.p, (size_t)n * d * 4, hipMemcpyDeviceToHost, h->st));
        HIPCHECK(hipStreamSynchronize(h->st));
    }
    const uint64_t seed = 1234;
    // ---- coarse quantiser: k-means, IP assignment, spherical centroids, niter 10, <=256 pts/centroid
    {
        const int k = h->nlist;
        int64_t keep = (int64_t)k * 256;
        std::vector<float> xs;
        const float* xt = hx.data();
        int64_t nt = n;
        if (n > keep) {
            std::vector<int64_t> perm; rand_perm(n, seed, perm);
            xs.resize((size_t)keep * d);
            for (int64_t i = 0; i < keep; i++) memcpy(&xs[(size_t)i * d], &hx[(size_t)perm[(size_t)i] * d], sizeof(float) * (size_t)d);
            xt = xs.data(); nt = keep;
        }
        std::vector<float> cen((size_t)k * d);
        std::vector<int64_t> perm; rand_perm(nt, seed + 1, perm);
        for (int j = 0; j < k; j++) memcpy(&cen[(size_t)j * d], xt + (size_t)perm[(size_t)(j % nt)] * d, sizeof(float) * (size_t)d);
        renorm_rows(d, k, cen.data());
        DevBuf dxt; dxt.ensure((size_t)nt * d * 4);
        HIPCHECK(hipMemcpyAsync(dxt.p, xt, (size_t)nt * d * 4, hipMemcpyHostToDevice, h->st));
        DevBuf dcen; dcen.ensure((size_t)k * d * 4);
        int ct = (k + 127) / 128;
        h->w_partial.ensure((size_t)nt * 2 * ct * 8);
        h->w_assign.ensure((size_t)nt * 4);
        std::vector<int32_t> assign((size_t)nt);
        KmeansWs kws;
        for (int it = 0; it < 10; it++) {
            HIPCHECK(hipMemcpyAsync(dcen.p, cen.data(), (size_t)k * d * 4, hipMemcpyHostToDevice, h->st));
            launch_gemm_exact_argmax(dxt.p, 0, nt, d, dcen.as<float>(), k, d, h->w_partial.as<uint64_t>(),
                                     h->w_assign.as<int32_t>(), nullptr, h->st);
            HIPCHECK(hipMemcpyAsync(assign.data(), h->w_assign.p, (size_t)nt * 4, hipMemcpyDeviceToHost, h->st));
            HIPCHECK(hipStreamSynchronize(h->st));
            kmeans_update_gpu(h, kws, dxt.as<float>(), d, 0, d, k, 1, nt, assign.data(), 0, 1, cen.data());
            renorm_rows(d, k, cen.data());
        }
        set_centroids(h, cen.data());
    }
    // ---- PQ codebooks on residuals: <=65536 points, per-subspace L2 k-means, niter 25
    if (h->kind == KIND_IVFPQ) {
        const int M = h->M, dsub = h->dsub, Mpad = h->Mpad;
        int64_t keep = 256 * 256;
        std::vector<float> xs;
        const float* xt = hx.data();
        int64_t nt = n;
        if (n > keep) {
            std::vector<int64_t> perm; rand_perm(n, seed, perm);
            xs.resize((size_t)keep * d);
            for (int64_t i = 0; i < keep; i++) memcpy(&xs[(size_t)i * d], &hx[(size_t)perm[(size_t)i] * d], sizeof(float) * (size_t)d);
            xt = xs.data(); nt = keep;
        }
        if (nt < 256) RSX_THROW(RSX_ERR_INVALID, "train: %lld points cannot train 256 PQ codewords", (long long)nt);
        DevBuf dxt, dres;
        dxt.ensure((size_t)nt * d * 4); dres.ensure((size_t)nt * d * 4);
        HIPCHECK(hipMemcpyAsync(dxt.p, xt, (size_t)nt * d * 4, hipMemcpyHostToDevice, h->st));
        int ct = (h->nlist + 127) / 128;
        h->w_partial.ensure((size_t)nt * 2 * ct * 8);
        h->w_assign.ensure((size_t)nt * 4);
        launch_gemm_exact_argmax(dxt.p, 0, nt, d, h->d_centroids.as<float>(), h->nlist, d, h->w_partial.as<uint64_t>(),
                                 h->w_assign.as<int32_t>(), nullptr, h->st);
        launch_residuals(dxt.as<float>(), nt, d, h->d_centroids.as<float>(), h->w_assign.as<int32_t>(), dres.as<float>(), h->st);
        std::vector<float> res((size_t)nt * d);
        HIPCHECK(hipMemcpyAsync(res.data(), dres.p, (size_t)nt * d * 4, hipMemcpyDeviceToHost, h->st));
        HIPCHECK(hipStreamSynchronize(h->st));

        std::vector<float> cb((size_t)M * 256 * dsub);
        for (int m = 0; m < M; m++) {
            std::vector<int64_t> perm; rand_perm(nt, seed + (uint64_t)m + 1, perm);
            for (int j = 0; j < 256; j++)
                memcpy(&cb[((size_t)m * 256 + j) * dsub], &res[(size_t)perm[(size_t)(j % nt)] * d + (size_t)m * dsub], sizeof(float) * (size_t)dsub);
        }
        DevBuf dcb, dcodes;
        dcb.ensure(cb.size() * 4); dcodes.ensure((size_t)nt * Mpad);
        std::vector<uint8_t> codes((size_t)nt * Mpad);
        std::vector<int32_t> a32((size_t)M * nt);
        KmeansWs kws;
        for (int it = 0; it < 25; it++) {
            HIPCHECK(hipMemcpyAsync(dcb.p, cb.data(), cb.size() * 4, hipMemcpyHostToDevice, h->st));
            launch_pq_encode(dres.p, 0, nt, d, d, M, Mpad, h->CB, nullptr, nullptr, dcb.as<float>(), nullptr, nullptr,
                             dcodes.as<uint8_t>(), h->st);
            HIPCHECK(hipMemcpyAsync(codes.data(), dcodes.p, codes.size(), hipMemcpyDeviceToHost, h->st));
            HIPCHECK(hipStreamSynchronize(h->st));
            for (int m = 0; m < M; m++)
                for (int64_t i = 0; i < nt; i++) a32[(size_t)m * nt + i] = codes[(size_t)i * Mpad + m];
            // all M sub-spaces in one accumulation launch (M x 256 x dsub chains over the residuals on the device)
            kmeans_update_gpu(h, kws, dres.as<float>(), d, dsub, dsub, 256, M, nt, a32.data(), nt, 1, cb.data());
        }
        set_codebooks(h, cb.data());
    }
    update_trained(h);
}


// ---------------------------------------------------------------------------------------
// list export / import, persistence
// ---------------------------------------------------------------------------------------
void get_list_impl(rsx_index* h, int64_t l, int64_t* n_out, void* codes_out, int64_t* ids_out) {
    if (h->kind == KIND_FLAT) l = 0;
    if (l < 0 || l >= h->nlist) RSX_THROW(RSX_ERR_INVALID, "list %lld out of range", (long long)l);
    int64_t n = h->h_len[(size_t)l], base = h->h_base[(size_t)l];
    if (n_out) *n_out = n;
    if (n == 0) return;
    if (codes_out) {
        DevBuf t;
        size_t bytes;
        if (h->kind == KIND_IVFPQ) {
            bytes = (size_t)n * h->M;
            t.ensure(bytes);
            launch_pq_export_list(h->data.as<uint8_t>(), base, n, h->M, h->Mpad, h->CB, t.as<uint8_t>(), h->st);
        } else {
            bytes = (size_t)n * h->d * 4;
            t.ensure(bytes);
            size_t esz = h->storage_f16 ? 2 : 4;
            launch_convert_to_f32(h->data.as<uint8_t>() + (size_t)base * h->ld * esz, h->storage_f16, h->ld, n, h->d, t.as<float>(), h->d, h->st);
        }
        HIPCHECK(hipMemcpyAsync(codes_out, t.p, bytes, is_device_ptr(codes_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->st));
        HIPCHECK(hipStreamSynchronize(h->st));
    }
    if (ids_out) {
        if (h->kind == KIND_FLAT && !h->custom_ids) {
            std::vector<int64_t> v((size_t)n);
            for (int64_t i = 0; i < n; i++) v[(size_t)i] = i;
            HIPCHECK(hipMemcpy(ids_out, v.data(), (size_t)n * 8, is_device_ptr(ids_out) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
        } else {
            HIPCHECK(hipMemcpy(ids_out, h->ids.as<int64_t>() + base, (size_t)n * 8,
                               is_device_ptr(ids_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
        }
    }
}

void add_list_impl(rsx_index* h, int64_t l, int64_t n, const void* codes, int dtype, const int64_t* ids) {
    if (h->kind == KIND_FLAT) RSX_THROW(RSX_ERR_INVALID, "add_list: use rsx_add for Flat");
    if (!h->trained) RSX_THROW(RSX_ERR_NOT_TRAINED, "add_list before train");
    if (l < 0 || l >= h->nlist) RSX_THROW(RSX_ERR_INVALID, "list %lld out of range", (long long)l);
    if (n <= 0) return;
    if (!ids) RSX_THROW(RSX_ERR_INVALID, "add_list: ids required");
    std::vector<int64_t> need(h->h_len);
    int64_t pos0 = need[(size_t)l];
    need[(size_t)l] += n;
    std::vector<int64_t> dest((size_t)n);
    if (h->kind == KIND_IVFPQ) {
        ensure_capacity(h, need, true);
        DevBuf t;
        const void* dc = codes;
        if (!is_device_ptr(codes)) {
            t.ensure((size_t)n * h->M);
            HIPCHECK(hipMemcpyAsync(t.p, codes, (size_t)n * h->M, hipMemcpyHostToDevice, h->st));
            dc = t.p;
        }
        launch_pq_import_list((const uint8_t*)dc, h->h_base[(size_t)l], pos0, n, h->M, h->Mpad, h->CB, h->data.as<uint8_t>(), h->st);
        for (int64_t i = 0; i < n; i++) dest[(size_t)i] = h->h_base[(size_t)l] + pos0 + i;
        h->w_dest.ensure((size_t)n * 8);
        HIPCHECK(hipMemcpyAsync(h->w_dest.p, dest.data(), (size_t)n * 8, hipMemcpyHostToDevice, h->st));
        const int64_t* dids = ids;
        DevBuf ti;
        if (!is_device_ptr(ids)) {
            ti.ensure((size_t)n * 8);
            HIPCHECK(hipMemcpyAsync(ti.p, ids, (size_t)n * 8, hipMemcpyHostToDevice, h->st));
            dids = ti.as<int64_t>();
        }
        launch_write_ids(h->w_dest.as<int64_t>(), dids, 0, n, h->ids.as<int64_t>(), h->st);
        HIPCHECK(hipStreamSynchronize(h->st));
    } else {
        const void* dx = stage_rows(h, h->w_x, codes, n, h->d, dtype);
        decide_storage(h, dx, n, dtype);
        track_max_norm(h, dx, n, dtype);
        ensure_capacity(h, need, true);
        for (int64_t i = 0; i < n; i++) dest[(size_t)i] = h->h_base[(size_t)l] + pos0 + i;
        h->w_dest.ensure((size_t)n * 8);
        HIPCHECK(hipMemcpyAsync(h->w_dest.p, dest.data(), (size_t)n * 8, hipMemcpyHostToDevice, h->st));
        const int64_t* dids = ids;
        DevBuf ti;
        if (!is_device_ptr(ids)) {
            ti.ensure((size_t)n * 8);
            HIPCHECK(hipMemcpyAsync(ti.p, ids, (size_t)n * 8, hipMemcpyHostToDevice, h->st));
            dids = ti.as<int64_t>();
        }
        launch_scatter_rows(dx, dtype == RSX_F16, n, h->d, h->w_dest.as<int64_t>(), h->data.p, h->storage_f16, h->ld,
                            h->norms.p ? h->norms.as<float>() : nullptr, dids, 0, h->ids.as<int64_t>(), h->st);
        HIPCHECK(hipStreamSynchronize(h->st));
    }
    h->h_len = need;
    h->ntotal += n;
    upload_dir(h);
}

void save_impl(rsx_index* h, const char* path) {
    FILE* f = fopen(path, "wb");
    if (!f) RSX_THROW(RSX_ERR_IO, "cannot open %s for writing", path);
    try {
        FileHeader hd{};
        memcpy(hd.magic, "RSX1", 4);
        hd.version = 2; hd.kind = h->kind; hd.d = h->d; hd.metric = h->metric; hd.nlist = h->nlist; hd.M = h->M;
        hd.nbits = h->nbits; hd.trained = h->trained; hd.storage_f16 = h->storage_f16; hd.custom_ids = h->custom_ids;
        hd.nprobe = h->nprobe; hd.ntotal = h->ntotal;
        wr(f, &hd, sizeof(hd));
        FileHeaderV2 h2{h->ndropped, h->add_list_mod, h->add_list_rem};
        wr(f, &h2, sizeof(h2));
        int64_t nc = (int64_t)h->h_centroids.size(), ncb = (int64_t)h->h_codebooks.size();
        wr(f, &nc, 8); wr(f, h->h_centroids.data(), (size_t)nc * 4);
        wr(f, &ncb, 8); wr(f, h->h_codebooks.data(), (size_t)ncb * 4);
        std::vector<uint8_t> buf; std::vector<int64_t> ib;
        if (h->kind == KIND_FLAT) {
            // one list of ntotal rows, streamed in bounded chunks (a 10M x 768 index is 30 GB as fp32: no whole-index temporaries)
            const int64_t n = h->h_len[0], CH = 262144;
            wr(f, &n, 8);
            DevBuf t; t.ensure((size_t)std::min(CH, std::max<int64_t>(n, 1)) * h->d * 4);
            buf.resize((size_t)std::min(CH, std::max<int64_t>(n, 1)) * h->d * 4);
            const size_t esz = h->storage_f16 ? 2 : 4;
            for (int64_t r0 = 0; r0 < n; r0 += CH) {
                const int64_t nb = std::min(CH, n - r0);
                launch_convert_to_f32(h->data.as<uint8_t>() + (size_t)r0 * h->ld * esz, h->storage_f16, h->ld, nb, h->d, t.as<float>(), h->d, h->st);
                HIPCHECK(hipMemcpyAsync(buf.data(), t.p, (size_t)nb * h->d * 4, hipMemcpyDeviceToHost, h->st));
                HIPCHECK(hipStreamSynchronize(h->st));
                wr(f, buf.data(), (size_t)nb * h->d * 4);
            }
            ib.resize((size_t)std::min(CH, std::max<int64_t>(n, 1)));
            for (int64_t r0 = 0; r0 < n; r0 += CH) {
                const int64_t nb = std::min(CH, n - r0);
                if (h->custom_ids) HIPCHECK(hipMemcpy(ib.data(), h->ids.as<int64_t>() + r0, (size_t)nb * 8, hipMemcpyDeviceToHost));
                else for (int64_t i = 0; i < nb; i++) ib[(size_t)i] = r0 + i;
                wr(f, ib.data(), (size_t)nb * 8);
            }
        } else
        for (int l = 0; l < h->nlist; l++) {
            int64_t n = h->h_len[(size_t)l];
            wr(f, &n, 8);
            if (n == 0) continue;
            size_t pb = (h->kind == KIND_IVFPQ) ? (size_t)n * h->M : (size_t)n * h->d * 4;
            buf.resize(pb); ib.resize((size_t)n);
            get_list_impl(h, l, nullptr, buf.data(), ib.data());
            wr(f, buf.data(), pb);
            wr(f, ib.data(), (size_t)n * 8);
        }
    } catch (...) { fclose(f); throw; }
    if (fclose(f) != 0) RSX_THROW(RSX_ERR_IO, "close failed for %s", path);
}

rsx_index* load_impl(const char* path, int device) {
    FILE* f = fopen(path, "rb");
    if (!f) RSX_THROW(RSX_ERR_IO, "cannot open %s", path);
    rsx_index* h = nullptr;
    try {
        FileHeader hd{};
        rd(f, &hd, sizeof(hd));
        if (memcmp(hd.magic, "RSX1", 4) != 0) RSX_THROW(RSX_ERR_IO, "%s is not an RSX1 index file", path);
        h = create_common(hd.kind, hd.d, hd.nlist, hd.M, hd.nbits, hd.metric, device);
        h->nprobe = hd.nprobe;
        FileHeaderV2 h2{0, 1, 0};
        if (hd.version >= 2) rd(f, &h2, sizeof(h2));
        int64_t nc = 0, ncb = 0;
        rd(f, &nc, 8);
        std::vector<float> c((size_t)nc); rd(f, c.data(), (size_t)nc * 4);
        rd(f, &ncb, 8);
        std::vector<float> cb((size_t)ncb); rd(f, cb.data(), (size_t)ncb * 4);
        if (nc) { if (nc != (int64_t)h->nlist * h->d) RSX_THROW(RSX_ERR_IO, "bad centroid block"); set_centroids(h, c.data()); }
        if (ncb) { if (ncb != (int64_t)h->M * 256 * h->dsub) RSX_THROW(RSX_ERR_IO, "bad codebook block"); set_codebooks(h, cb.data()); }
        update_trained(h);
        std::vector<int64_t> lens((size_t)h->nlist);
        long dir_pos = ftell(f);
        // first pass: list sizes (to reserve exactly)
        for (int l = 0; l < h->nlist; l++) {
            int64_t n = 0; rd(f, &n, 8); lens[(size_t)l] = n;
            size_t pb = (h->kind == KIND_IVFPQ) ? (size_t)n * h->M : (size_t)n * h->d * 4;
            if (n && fseek(f, (long)(pb + (size_t)n * 8), SEEK_CUR) != 0) RSX_THROW(RSX_ERR_IO, "seek failed");
        }
        fseek(f, dir_pos, SEEK_SET);
        if (!hd.storage_f16 && h->kind != KIND_IVFPQ) { h->storage_f16 = 0; h->storage_decided = true; }
        if (h->kind == KIND_FLAT && hd.custom_ids) h->custom_ids = true;
        ensure_capacity(h, lens, true);      // exact reservation: the load never re-lays-out HBM
        std::vector<uint8_t> buf; std::vector<int64_t> ib;
        if (h->kind == KIND_FLAT) {
            // rows then ids, both streamed in bounded chunks (the ids sit behind the rows: two file cursors)
            int64_t n = 0; rd(f, &n, 8);
            const int64_t CH = 262144;
            const long rows_pos = ftell(f);
            const long ids_pos = rows_pos + (long)((size_t)n * h->d * 4);
            buf.resize((size_t)std::min(CH, std::max<int64_t>(n, 1)) * h->d * 4); ib.resize((size_t)std::min(CH, std::max<int64_t>(n, 1)));
            for (int64_t r0 = 0; r0 < n; r0 += CH) {
                const int64_t nb = std::min(CH, n - r0);
                if (fseek(f, rows_pos + (long)((size_t)r0 * h->d * 4), SEEK_SET) != 0) RSX_THROW(RSX_ERR_IO, "seek failed");
                rd(f, buf.data(), (size_t)nb * h->d * 4);
                if (hd.custom_ids) {
                    if (fseek(f, ids_pos + (long)((size_t)r0 * 8), SEEK_SET) != 0) RSX_THROW(RSX_ERR_IO, "seek failed");
                    rd(f, ib.data(), (size_t)nb * 8);
                }
                add_all(h, nb, buf.data(), RSX_F32, hd.custom_ids ? ib.data() : nullptr);
            }
        } else
        for (int l = 0; l < h->nlist; l++) {
            int64_t n = 0; rd(f, &n, 8);
            if (n == 0) continue;
            size_t pb = (h->kind == KIND_IVFPQ) ? (size_t)n * h->M : (size_t)n * h->d * 4;
            buf.resize(pb); ib.resize((size_t)n);
            rd(f, buf.data(), pb); rd(f, ib.data(), (size_t)n * 8);
            add_list_impl(h, l, n, buf.data(), RSX_F32, ib.data());
        }
        h->ndropped = h2.ndropped; h->add_list_mod = h2.add_list_mod; h->add_list_rem = h2.add_list_rem;
    } catch (...) {
        fclose(f);
        if (h) { if (h->st) (void)hipStreamDestroy(h->st); delete h; }
        throw;
    }
    fclose(f);
    return h;
}

