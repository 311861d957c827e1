// api_sharded.hip — the single-process multi-GPU handle (rsx_sharded_create / rsx_load_sharded).  Shared declarations: rsx_host.h.
#include "rsx_host.h"

// ---------------------------------------------------------------------------------------
// Single-process multi-GPU handle: N child indexes (one per device) behind one rsx_index_t.  The reference's driver makes
// ONE index.search(all_queries, k) call (src/search.py:296) and its serving tier fans the query out to shard workers over
// HTTP and re-sorts (api/serve_main_node.py:281-323); here the fan-out is N host threads driving N GPUs and the fan-in is
// a device-to-device copy of each shard's [nq, k] block plus one merge kernel on the first device.
//   * add: every call's rows are cut into N contiguous pieces, piece r -> shard r, ids = the logical index's sequential
//     ids, so the union of the shards' lists IS the single index's lists and the merged result (score desc, id asc) is
//     bit-identical to one index holding everything.
//   * trained parameters are identical on every shard (trained once on the first, copied).
// ---------------------------------------------------------------------------------------
rsx_index* sharded_create(int kind, int d, int nlist, int M, int nbits, int metric, int ndev, const int* devices) {
    if (ndev <= 0 || !devices) RSX_THROW(RSX_ERR_INVALID, "sharded_create: need at least one device");
    if (ndev > 64) RSX_THROW(RSX_ERR_INVALID, "sharded_create: %d shards", ndev);
    std::unique_ptr<rsx_index> p(new rsx_index());
    p->w_uncertain.host_mapped = true;
    try {
        for (int r = 0; r < ndev; r++) p->shards.push_back(create_common(kind, d, nlist, M, nbits, metric, devices[r]));
    } catch (...) {
        for (auto* c : p->shards) { if (c->st) { (void)hipSetDevice(c->device); (void)hipStreamDestroy(c->st); } delete c; }
        throw;
    }
    rsx_index* c0 = p->shards[0];
    p->kind = kind; p->d = d; p->metric = metric; p->device = devices[0];
    p->nlist = c0->nlist; p->M = c0->M; p->nbits = c0->nbits; p->Mpad = c0->Mpad; p->CB = c0->CB; p->dsub = c0->dsub; p->ld = c0->ld;
    p->trained = c0->trained;
    HIPCHECK(hipSetDevice(p->device));
    HIPCHECK(hipStreamCreateWithFlags(&p->st, hipStreamNonBlocking));
    return p.release();
}

void sharded_sync_trained(rsx_index* h) {
    rsx_index* c0 = h->shards[0];
    for (size_t r = 1; r < h->shards.size(); r++) {
        rsx_index* c = h->shards[r];
        HIPCHECK(hipSetDevice(c->device));
        if (!c0->h_centroids.empty()) set_centroids(c, c0->h_centroids.data());
        if (!c0->h_codebooks.empty()) set_codebooks(c, c0->h_codebooks.data());
        update_trained(c);
    }
    h->trained = c0->trained;
}

static int ptr_device(const void* p) {   // device ordinal of a device pointer, -1 for host memory
    hipPointerAttribute_t at;
    if (!p || hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged) ? at.device : -1;
}

void sharded_add(rsx_index* h, int64_t n, const void* x, int dtype, const int64_t* ids) {
    const int N = (int)h->shards.size();
    const size_t esz = dtype == RSX_F16 ? 2 : 4;
    const int xdev = ptr_device(x);
    std::vector<int64_t> seq;
    if (!ids) { seq.resize((size_t)n); for (int64_t i = 0; i < n; i++) seq[(size_t)i] = h->sh_next_id + i; }
    std::vector<int64_t> hids;
    if (ids && ptr_device(ids) >= 0) {   // device ids: bring them to the host once (children stage host ids themselves)
        hids.resize((size_t)n);
        HIPCHECK(hipMemcpy(hids.data(), ids, (size_t)n * 8, hipMemcpyDeviceToHost));
    }
    const int64_t* hid = ids ? (hids.empty() ? ids : hids.data()) : seq.data();
    for_each_shard_parallel(h, [&](int r, rsx_index* c) {
        const int64_t lo = n * r / N, hi = n * (r + 1) / N;
        if (hi <= lo) return;
        const char* xp = (const char*)x + (size_t)lo * h->d * esz;
        DevBuf tmp;
        if (xdev >= 0 && xdev != c->device) {     // rows live on another GPU: one peer copy into this shard's staging buffer
            // (on the child's own stream + a stream synchronise: a plain device-to-device hipMemcpy is not guaranteed to block
            //  the host, and the child's kernels run on a non-blocking stream that is not ordered after the null stream)
            tmp.ensure((size_t)(hi - lo) * h->d * esz);
            HIPCHECK(hipMemcpyAsync(tmp.p, xp, (size_t)(hi - lo) * h->d * esz, hipMemcpyDefault, c->st));
            HIPCHECK(hipStreamSynchronize(c->st));
            xp = (const char*)tmp.p;
        }
        add_all(c, hi - lo, xp, dtype, hid + lo);
    });
    h->ntotal += n;
    if (!ids) h->sh_next_id += n;
}

void sharded_search(rsx_index* h, int64_t nq, const void* q, int dtype, int k, float* D, int64_t* I) {
    const int N = (int)h->shards.size();
    if (nq < 0 || k <= 0) RSX_THROW(RSX_ERR_INVALID, "search: nq=%lld k=%d", (long long)nq, k);
    if (k > 4096) RSX_THROW(RSX_ERR_UNSUPPORTED, "search: k = %d exceeds this build's maximum of 4096 (the reference backends' default k)", k);
    if (nq == 0) return;
    if (!q || !D || !I) RSX_THROW(RSX_ERR_INVALID, "search: null pointer");
    const bool o_dev = is_device_ptr(D);
    if (o_dev != is_device_ptr(I)) RSX_THROW(RSX_ERR_INVALID, "search: D and I must both be host or both device pointers");
    const size_t esz = dtype == RSX_F16 ? 2 : 4;
    const int qdev = ptr_device(q);
    const size_t blk = (size_t)nq * k;
    HIPCHECK(hipSetDevice(h->device));
    h->sh_D.ensure((size_t)N * blk * 4); h->sh_I.ensure((size_t)N * blk * 8);
    for_each_shard_parallel(h, [&](int r, rsx_index* c) {
        const void* qp = q;
        if (qdev >= 0 && qdev != c->device) {     // the caller's queries sit on another GPU: peer copy on this shard's stream
            c->sh_q.ensure((size_t)nq * h->d * esz);
            HIPCHECK(hipMemcpyAsync(c->sh_q.p, q, (size_t)nq * h->d * esz, hipMemcpyDefault, c->st));
            HIPCHECK(hipStreamSynchronize(c->st));
            qp = c->sh_q.p;
        }
        c->sh_oD.ensure(blk * 4); c->sh_oI.ensure(blk * 8);
        search_impl(c, nq, qp, dtype, k, c->sh_oD.as<float>(), c->sh_oI.as<int64_t>());   // synchronises c->st
        // fan-in: this shard's [nq, k] block -> the parent device's gather buffers, asynchronously on the shard's stream (all
        // shards copy concurrently); the stream is synchronised before the thread joins, so the merge below sees every block
        HIPCHECK(hipMemcpyAsync(h->sh_D.as<float>() + (size_t)r * blk, c->sh_oD.p, blk * 4, hipMemcpyDefault, c->st));
        HIPCHECK(hipMemcpyAsync(h->sh_I.as<int64_t>() + (size_t)r * blk, c->sh_oI.p, blk * 8, hipMemcpyDefault, c->st));
        HIPCHECK(hipStreamSynchronize(c->st));
    });
    HIPCHECK(hipSetDevice(h->device));
    float* dD = D; int64_t* dI = I;
    if (!o_dev || ptr_device(D) != h->device) {
        h->sh_oD.ensure(blk * 4); h->sh_oI.ensure(blk * 8);
        dD = h->sh_oD.as<float>(); dI = h->sh_oI.as<int64_t>();
    }
    // merge by (score, id) — associative, so any k works for any shard count: rounds of groups of G blocks with G * k <= 8192
    // (one launch for the usual k; k = 4096 on 8 shards takes three rounds of pairs)
    float* srcD = h->sh_D.as<float>(); int64_t* srcI = h->sh_I.as<int64_t>();
    int cur = N;
    const int G = std::max(2, 8192 / k);
    DevBuf tD[2], tI[2];
    int flip = 0;
    while (cur > G) {
        const int groups = (cur + G - 1) / G;
        tD[flip].ensure((size_t)groups * blk * 4); tI[flip].ensure((size_t)groups * blk * 8);
        for (int g = 0; g < groups; g++) {
            const int n = std::min(G, cur - g * G);
            launch_merge_topk_byid(n, nq, k, h->metric, srcD + (size_t)g * G * blk, srcI + (size_t)g * G * blk,
                                   tD[flip].as<float>() + (size_t)g * blk, tI[flip].as<int64_t>() + (size_t)g * blk, h->st);
        }
        srcD = tD[flip].as<float>(); srcI = tI[flip].as<int64_t>();
        cur = groups; flip ^= 1;
    }
    launch_merge_topk_byid(cur, nq, k, h->metric, srcD, srcI, dD, dI, h->st);
    if (dD != D) {
        HIPCHECK(hipMemcpyAsync(D, dD, blk * 4, hipMemcpyDefault, h->st));
        HIPCHECK(hipMemcpyAsync(I, dI, blk * 8, hipMemcpyDefault, h->st));
    }
    HIPCHECK(hipStreamSynchronize(h->st));     // also keeps the round buffers alive until the merges have run
    HIPCHECK(hipGetLastError());
}

// Bulk import of one inverted list into a sharded handle: the rows are cut into N contiguous pieces like every add call
// (piece r -> shard r, ids kept), so a FAISS / RSX1 file written from ONE index can be spread over the node while loading.
void sharded_add_list(rsx_index* h, int64_t l, int64_t n, const void* codes, int dtype, const int64_t* ids) {
    if (h->kind == KIND_FLAT) RSX_THROW(RSX_ERR_INVALID, "add_list: use rsx_add for Flat");
    if (!ids) RSX_THROW(RSX_ERR_INVALID, "add_list: ids required");
    if (ptr_device(codes) >= 0 || ptr_device(ids) >= 0) RSX_THROW(RSX_ERR_UNSUPPORTED, "add_list on a sharded handle takes host pointers");
    if (n <= 0) return;
    const int N = (int)h->shards.size();
    const size_t rowb = h->kind == KIND_IVFPQ ? (size_t)h->M : (size_t)h->d * (dtype == RSX_F16 ? 2 : 4);
    for_each_shard_parallel(h, [&](int r, rsx_index* c) {
        const int64_t lo = n * r / N, hi = n * (r + 1) / N;
        if (hi > lo) add_list_impl(c, l, hi - lo, (const char*)codes + (size_t)lo * rowb, dtype, ids + lo);
    });
    h->ntotal += n;
    // a later rsx_add without ids continues the sequence where the unsharded handle (and FAISS) would: at ntotal — a `.faiss`
    // file re-sharded through rsx_add_list used to leave the counter at 0 and hand out ids 0..n-1 again
    h->sh_next_id = std::max(h->sh_next_id, h->ntotal);
}
void sharded_reserve(rsx_index* h, const int64_t* counts) {      // exact when every list arrives in ONE add_list call
    const int N = (int)h->shards.size();
    for_each_shard_parallel(h, [&](int r, rsx_index* c) {
        std::vector<int64_t> need((size_t)c->nlist);
        for (int l = 0; l < c->nlist; l++)
            need[(size_t)l] = std::max(counts[l] * (r + 1) / N - counts[l] * r / N, c->h_len[(size_t)l]);
        ensure_capacity(c, need, true);
    });
}

void sharded_save(rsx_index* h, const char* path) {
    FILE* f = fopen(path, "wb");
    if (!f) RSX_THROW(RSX_ERR_IO, "cannot open %s for writing", path);
    int32_t hdr[4] = {0, 1, (int32_t)h->shards.size(), 0};
    memcpy(hdr, "RSXS", 4);
    int64_t next = h->sh_next_id;
    bool ok = fwrite(hdr, 1, sizeof(hdr), f) == sizeof(hdr) && fwrite(&next, 1, 8, f) == 8;
    if (fclose(f) != 0 || !ok) RSX_THROW(RSX_ERR_IO, "write failed for %s", path);
    for_each_shard_parallel(h, [&](int r, rsx_index* c) {
        save_impl(c, (std::string(path) + ".shard" + std::to_string(r)).c_str());
    });
}

void destroy_handle(rsx_index* h) {
    if (!h) return;
    for (auto* c : h->shards) { if (c->st) { (void)hipSetDevice(c->device); (void)hipStreamDestroy(c->st); } delete c; }
    if (h->st) { (void)hipSetDevice(h->device); (void)hipStreamDestroy(h->st); }
    delete h;
}

// An RSX1 file written from ONE (unsharded) index, loaded onto several devices: the lists / rows are spread over the
// shards as they stream in (nothing is staged on one GPU first), ids are kept, so the handle answers exactly like the
// index the file was written from.
static rsx_index* sharded_load_plain(const char* path, int ndev, const int* devices) {
    FILE* f = fopen(path, "rb");
    if (!f) RSX_THROW(RSX_ERR_IO, "cannot open %s", path);
    rsx_index* p = nullptr;
    try {
        FileHeader hd{};
        rd(f, &hd, sizeof(hd));
        if (memcmp(hd.magic, "RSX1", 4) != 0) RSX_THROW(RSX_ERR_IO, "%s is not an RSX1 index file", path);
        FileHeaderV2 h2{0, 1, 0};
        if (hd.version >= 2) rd(f, &h2, sizeof(h2));
        if (h2.add_list_mod > 1) RSX_THROW(RSX_ERR_UNSUPPORTED, "%s is a list shard (add_list_mod = %d): it cannot be re-sharded", path, h2.add_list_mod);
        p = sharded_create(hd.kind, hd.d, hd.nlist, hd.M, hd.nbits, hd.metric, ndev, devices);
        p->nprobe = hd.nprobe;
        for (auto* c : p->shards) c->nprobe = hd.nprobe;
        int64_t nc = 0, ncb = 0;
        rd(f, &nc, 8);
        std::vector<float> cen((size_t)nc); rd(f, cen.data(), (size_t)nc * 4);
        rd(f, &ncb, 8);
        std::vector<float> cb((size_t)ncb); rd(f, cb.data(), (size_t)ncb * 4);
        if (nc && nc != (int64_t)p->nlist * p->d) RSX_THROW(RSX_ERR_IO, "bad centroid block");
        if (ncb && ncb != (int64_t)p->M * 256 * p->dsub) RSX_THROW(RSX_ERR_IO, "bad codebook block");
        for (auto* c : p->shards) {
            HIPCHECK(hipSetDevice(c->device));
            if (nc) set_centroids(c, cen.data());
            if (ncb) set_codebooks(c, cb.data());
            update_trained(c);
            if (!hd.storage_f16 && c->kind != KIND_IVFPQ) { c->storage_f16 = 0; c->storage_decided = true; }
        }
        p->trained = p->shards[0]->trained;
        std::vector<uint8_t> buf; std::vector<int64_t> ib;
        if (p->kind == KIND_FLAT) {
            int64_t n = 0; rd(f, &n, 8);
            const int64_t CH = 262144;
            const long rows_pos = ftell(f);
            const long ids_pos = rows_pos + (long)((size_t)n * p->d * 4);
            buf.resize((size_t)std::min(CH, std::max<int64_t>(n, 1)) * p->d * 4); ib.resize((size_t)std::min(CH, std::max<int64_t>(n, 1)));
            for (int64_t r0 = 0; r0 < n; r0 += CH) {
                const int64_t nb = std::min(CH, n - r0);
                if (fseek(f, rows_pos + (long)((size_t)r0 * p->d * 4), SEEK_SET) != 0) RSX_THROW(RSX_ERR_IO, "seek failed");
                rd(f, buf.data(), (size_t)nb * p->d * 4);
                if (hd.custom_ids) {
                    if (fseek(f, ids_pos + (long)((size_t)r0 * 8), SEEK_SET) != 0) RSX_THROW(RSX_ERR_IO, "seek failed");
                    rd(f, ib.data(), (size_t)nb * 8);
                }
                sharded_add(p, nb, buf.data(), RSX_F32, hd.custom_ids ? ib.data() : nullptr);
            }
        } else {
            std::vector<int64_t> lens((size_t)p->nlist);
            const long dir_pos = ftell(f);
            for (int l = 0; l < p->nlist; l++) {
                int64_t n = 0; rd(f, &n, 8); lens[(size_t)l] = n;
                const size_t pb = (p->kind == KIND_IVFPQ) ? (size_t)n * p->M : (size_t)n * p->d * 4;
                if (n && fseek(f, (long)(pb + (size_t)n * 8), SEEK_CUR) != 0) RSX_THROW(RSX_ERR_IO, "seek failed");
            }
            fseek(f, dir_pos, SEEK_SET);
            sharded_reserve(p, lens.data());
            for (int l = 0; l < p->nlist; l++) {
                int64_t n = 0; rd(f, &n, 8);
                if (n == 0) continue;
                const size_t pb = (p->kind == KIND_IVFPQ) ? (size_t)n * p->M : (size_t)n * p->d * 4;
                buf.resize(pb); ib.resize((size_t)n);
                rd(f, buf.data(), pb); rd(f, ib.data(), (size_t)n * 8);
                sharded_add_list(p, l, n, buf.data(), RSX_F32, ib.data());
            }
        }
        p->sh_next_id = hd.ntotal + h2.ndropped;
    } catch (...) {
        fclose(f);
        destroy_handle(p);
        throw;
    }
    fclose(f);
    return p;
}

rsx_index* sharded_load(const char* path, int ndev, const int* devices) {
    FILE* f = fopen(path, "rb");
    if (!f) RSX_THROW(RSX_ERR_IO, "cannot open %s", path);
    int32_t hdr[4] = {0, 0, 0, 0}; int64_t next = 0;
    bool ok = fread(hdr, 1, sizeof(hdr), f) == sizeof(hdr) && fread(&next, 1, 8, f) == 8;
    fclose(f);
    if (ok && memcmp(hdr, "RSX1", 4) == 0) {      // a plain single-index file: re-shard it over the devices while loading
        if (ndev <= 0 || !devices) RSX_THROW(RSX_ERR_INVALID, "load_sharded: need at least one device");
        return sharded_load_plain(path, ndev, devices);
    }
    if (!ok || memcmp(hdr, "RSXS", 4) != 0) RSX_THROW(RSX_ERR_IO, "%s is not a sharded (RSXS) index manifest", path);
    const int ns = hdr[2];
    if (ns <= 0 || ns > 64) RSX_THROW(RSX_ERR_IO, "bad shard count in %s", path);
    if (ndev <= 0 || !devices) RSX_THROW(RSX_ERR_INVALID, "load_sharded: need at least one device");
    std::unique_ptr<rsx_index> p(new rsx_index());
    p->w_uncertain.host_mapped = true;
    try {
        for (int r = 0; r < ns; r++)   // more shards than devices: several shards share a device
            p->shards.push_back(load_impl((std::string(path) + ".shard" + std::to_string(r)).c_str(), devices[r % ndev]));
    } catch (...) {
        for (auto* c : p->shards) { if (c->st) { (void)hipSetDevice(c->device); (void)hipStreamDestroy(c->st); } delete c; }
        throw;
    }
    rsx_index* c0 = p->shards[0];
    p->kind = c0->kind; p->d = c0->d; p->metric = c0->metric; p->device = c0->device;
    p->nlist = c0->nlist; p->M = c0->M; p->nbits = c0->nbits; p->Mpad = c0->Mpad; p->CB = c0->CB; p->dsub = c0->dsub; p->ld = c0->ld;
    p->trained = c0->trained; p->nprobe = c0->nprobe; p->sh_next_id = next;
    for (auto* c : p->shards) p->ntotal += c->ntotal;
    HIPCHECK(hipSetDevice(p->device));
    HIPCHECK(hipStreamCreateWithFlags(&p->st, hipStreamNonBlocking));
    return p.release();
}

