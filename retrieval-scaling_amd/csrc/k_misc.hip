// k_misc.hip — dtype conversion, synthetic data, list-storage maintenance (scatter, re-layout,
// PQ slab import/export).  Memory-bound byte movers: coalesced 16-byte accesses, grid-stride.
#include "rsx_internal.h"

namespace rsx {

// ---------------------------------------------------------------------------------------
// Conversions.  The reference up-casts fp16 embeddings to fp32 at the FAISS boundary
// (src/indicies/flat.py:86,139); here fp16 stays fp16 in HBM and queries are widened once.
// ---------------------------------------------------------------------------------------
__global__ void k_to_f32(const void* src, int src_f16, int64_t src_ld, int64_t n_rows, int d, float* dst, int ld) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = n_rows * ld;
    for (; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i / ld; int t = (int)(i - r * ld);
        float v = 0.0f;
        if (t < d) v = src_f16 ? __half2float(((const __half*)src)[r * src_ld + t]) : ((const float*)src)[r * src_ld + t];
        dst[i] = v;
    }
}
void launch_convert_to_f32(const void* src, int src_f16, int64_t src_ld, int64_t n_rows, int d, float* dst, int ld, hipStream_t st) {
    int64_t total = n_rows * ld;
    if (total <= 0) return;
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_to_f32, dim3(blocks), dim3(256), 0, st, src, src_f16, src_ld, n_rows, d, dst, ld);
}

// both copies of a query batch in one launch (IVF-PQ with the certified fp16 coarse quantiser): fp32 [n_rows][ld] and fp16 [pad_rows_to][ld]
__global__ void k_to_f32_f16(const void* src, int src_f16, int64_t src_ld, int64_t n_rows, int d, float* dst32, __half* dst16, int ld, int64_t pad_rows_to) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = pad_rows_to * ld;
    for (; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld; const int t = (int)(i - r * ld);
        float v = 0.0f;
        if (r < n_rows && t < d) v = src_f16 ? __half2float(((const __half*)src)[r * src_ld + t]) : ((const float*)src)[r * src_ld + t];
        if (r < n_rows) dst32[i] = v;
        dst16[i] = __float2half_rn(v);
    }
}
void launch_convert_to_f32_f16(const void* src, int src_f16, int64_t src_ld, int64_t n_rows, int d, float* dst32, __half* dst16, int ld,
                               int64_t pad_rows_to, hipStream_t st) {
    const int64_t total = pad_rows_to * ld;
    if (total <= 0) return;
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_to_f32_f16, dim3(blocks), dim3(256), 0, st, src, src_f16, src_ld, n_rows, d, dst32, dst16, ld, pad_rows_to);
}

// rows [n_rows, pad_rows_to) and columns [d, ld) are zero-filled.  *flag |= 1 if an fp32 input
// value is not exactly representable in fp16.
__global__ void k_to_f16(const void* src, int src_f16, int64_t n_rows, int d, __half* dst, int ld,
                         int64_t pad_rows_to, int* flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = pad_rows_to * ld;
    bool bad = false;
    for (; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i / ld; int t = (int)(i - r * ld);
        __half h = __float2half(0.0f);
        if (r < n_rows && t < d) {
            if (src_f16) h = ((const __half*)src)[r * d + t];
            else {
                float v = ((const float*)src)[r * d + t];
                h = __float2half_rn(v);
                if (!(__half2float(h) == v)) bad = true;
            }
        }
        dst[i] = h;
    }
    if (flag && bad) atomicOr(flag, 1);
}
void launch_convert_to_f16(const void* src, int src_f16, int64_t n_rows, int d, __half* dst, int ld,
                           int64_t pad_rows_to, int* flag, hipStream_t st) {
    int64_t total = pad_rows_to * ld;
    if (total <= 0) return;
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_to_f16, dim3(blocks), dim3(256), 0, st, src, src_f16, n_rows, d, dst, ld, pad_rows_to, flag);
}

__global__ void k_fill_u64(uint64_t* p, int64_t n, uint64_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill_u64(uint64_t* p, int64_t n, uint64_t v, hipStream_t st) {
    if (n <= 0) return;
    int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fill_u64, dim3(blocks), dim3(256), 0, st, p, n, v);
}
// rows of `len` keys: keep only the last key of each row (a selection state reduced to its threshold)
__global__ void k_keep_last_u64(uint64_t* p, int64_t rows, int len) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = rows * len;
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (i % len != len - 1) p[i] = 0;
}
void launch_keep_last_u64(uint64_t* p, int64_t rows, int len, hipStream_t st) {
    if (rows <= 0 || len <= 1) return;
    int64_t n = rows * len;
    int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_keep_last_u64, dim3(blocks), dim3(256), 0, st, p, rows, len);
}
__global__ void k_fill_f32(float* p, int64_t n, float v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill_f32(float* p, int64_t n, float v, hipStream_t st) {
    if (n <= 0) return;
    int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fill_f32, dim3(blocks), dim3(256), 0, st, p, n, v);
}

// ---------------------------------------------------------------------------------------
// Synthetic Gaussian mixture — bit-identical to oracle/rsx_oracle.c orc_synth_* (integer hash,
// Irwin-Hall(8) normal approximation, one fmaf, one round-to-nearest-even to fp16).
// ---------------------------------------------------------------------------------------
__device__ inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ inline uint32_t hash4(uint32_t seed, uint64_t i, uint32_t t, uint32_t s) {
    uint32_t x = mix32(seed + 0x9E3779B9u * (s + 1u));
    x = mix32(x ^ (uint32_t)(i & 0xffffffffu));
    x = mix32(x ^ ((uint32_t)(i >> 32) * 0x85EBCA6Bu) ^ (t * 0xC2B2AE35u));
    return x;
}
__device__ inline float synth_z(uint32_t seed, uint64_t i, uint32_t t) {
    uint32_t S = 0;
#pragma unroll
    for (uint32_t s = 0; s < 4; s++) { uint32_t h = hash4(seed, i, t, s); S += (h & 0xffffu) + (h >> 16); }
    return __fmul_rn(__fsub_rn((float)S, 262140.0f), 1.8688064e-5f);
}
__device__ inline uint32_t synth_pick(uint32_t seed, uint64_t i, uint32_t n) { return hash4(seed, i, 0xffffffffu, 7u) % n; }
__device__ inline __half synth_elem(int ncentres, uint32_t seed_c, uint32_t seed_x, float sigma, uint64_t i, uint32_t t) {
    uint32_t j = synth_pick(seed_x, i, (uint32_t)ncentres);
    float c = synth_z(seed_c, j, t);
    float v = __fmaf_rn(sigma, synth_z(seed_x, i, t), c);
    // keep the fp32 rounding step: without this hipcc fuses fma + convert into v_fma_mixlo_f16
    // (one rounding straight to fp16), which is not what the spec / oracle compute
    asm volatile("" : "+v"(v));
    return __float2half_rn(v);
}
__global__ void k_synth_vectors(int d, int ncentres, uint32_t seed_c, uint32_t seed_x, float sigma, int64_t i0,
                                int64_t n, __half* out) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = n * d;
    for (; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e / d; uint32_t t = (uint32_t)(e - r * d);
        out[e] = synth_elem(ncentres, seed_c, seed_x, sigma, (uint64_t)(i0 + r), t);
    }
}
void launch_synth_vectors(int d, int ncentres, uint32_t seed_c, uint32_t seed_x, float sigma, int64_t i0,
                          int64_t n, __half* out, hipStream_t st) {
    int64_t total = n * d;
    if (total <= 0) return;
    int64_t blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_synth_vectors, dim3((unsigned)blocks), dim3(256), 0, st, d, ncentres, seed_c, seed_x, sigma, i0, n, out);
}
__global__ void k_synth_queries(int d, int ncentres, uint32_t seed_c, uint32_t seed_x, float sigma, int64_t nbase,
                                uint32_t seed_q, float sigma_q, int64_t r0, int64_t n, __half* out) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = n * d;
    for (; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e / d; uint32_t t = (uint32_t)(e - r * d);
        uint64_t rr = (uint64_t)(r0 + r);
        uint64_t b = synth_pick(seed_q, rr, (uint32_t)nbase);
        float base = __half2float(synth_elem(ncentres, seed_c, seed_x, sigma, b, t));
        float v = __fmaf_rn(sigma_q, synth_z(seed_q, rr, t), base);
        asm volatile("" : "+v"(v));
        out[e] = __float2half_rn(v);
    }
}
void launch_synth_queries(int d, int ncentres, uint32_t seed_c, uint32_t seed_x, float sigma, int64_t nbase,
                          uint32_t seed_q, float sigma_q, int64_t r0, int64_t n, __half* out, hipStream_t st) {
    int64_t total = n * d;
    if (total <= 0) return;
    int64_t blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_synth_queries, dim3((unsigned)blocks), dim3(256), 0, st, d, ncentres, seed_c, seed_x, sigma,
                       nbase, seed_q, sigma_q, r0, n, out);
}

// ---------------------------------------------------------------------------------------
// Flat / IVF-Flat population: copy batch row i to storage row dest_row[i] (one wave per row),
// widening/narrowing as the storage dtype requires, zero the [d, ld) padding columns, record
// |x|^2 (fp32 fmaf chain) for the L2 ranking bias, and the id.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scatter_rows(const void* x, int x_f16, int64_t n, int d,
                                                      const int64_t* dest_row, void* storage, int storage_f16,
                                                      int ld, float* norms, const int64_t* ids_in, int64_t id0,
                                                      int64_t* ids_storage) {
    const int lane = threadIdx.x & 63;
    int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    int64_t dst = dest_row ? dest_row[i] : i;
    if (dst < 0) return;                         // the vector's list belongs to another shard
    for (int t = lane; t < ld; t += 64) {
        float v = 0.0f;
        if (t < d) v = x_f16 ? __half2float(((const __half*)x)[i * d + t]) : ((const float*)x)[i * d + t];
        if (storage_f16) ((__half*)storage)[dst * ld + t] = __float2half_rn(v);
        else ((float*)storage)[dst * ld + t] = v;
    }
    if (lane == 0) {
        if (norms) {
            float s = 0.0f;
            for (int t = 0; t < d; t++) {
                float v = x_f16 ? __half2float(((const __half*)x)[i * d + t]) : ((const float*)x)[i * d + t];
                s = __fmaf_rn(v, v, s);
            }
            norms[dst] = s;
        }
        if (ids_storage) ids_storage[dst] = ids_in ? ids_in[i] : id0 + i;
    }
}
// Lloyd update, accumulation step (faiss::Clustering restated in oracle/rsx_oracle.c): sums[s][c][t] = x[i0][col_s + t] +
// x[i1][col_s + t] + ... over the points assigned to centroid c of segment-set s IN POINT ORDER, one sequential fp32 chain
// per (s, c, t) starting from 0 — bit-identical to the host loop it replaces (which summed point by point into the
// centroid rows), with nsets * k * d independent chains in flight.  order[s][.] lists the points grouped by centroid
// (stable), seg_off[s][c .. c+1] delimits centroid c's run.  Coarse quantiser: nsets = 1, col = 0, d = the full
// dimension; PQ codebooks: nsets = M sub-spaces, col_s = s * dsub, d = dsub, k = 256.
__global__ __launch_bounds__(256) void k_kmeans_accumulate(const float* x, int64_t ldx, int col_stride, int d, int k, int nsets, int64_t n,
                                                           const int32_t* order, const int32_t* seg_off, float* sums) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (s, c, t), t fastest
    const int64_t per_set = (int64_t)k * d;
    const int s = (int)(gid / per_set);
    if (s >= nsets) return;
    const int64_t r = gid - (int64_t)s * per_set;
    const int c = (int)(r / d), t = (int)(r - (int64_t)c * d);
    const int32_t* ord = order + (int64_t)s * n;
    const int32_t* so = seg_off + (int64_t)s * (k + 1);
    const float* xs = x + (int64_t)s * col_stride + t;
    float acc = 0.0f;
    for (int32_t j = so[c]; j < so[c + 1]; j++) acc += xs[(int64_t)ord[j] * ldx];
    sums[gid] = acc;
}
void launch_kmeans_accumulate(const float* x, int64_t ldx, int col_stride, int d, int k, int nsets, int64_t n, const int32_t* order,
                              const int32_t* seg_off, float* sums, hipStream_t st) {
    const int64_t total = (int64_t)nsets * k * d;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_kmeans_accumulate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, ldx, col_stride, d, k, nsets, n,
                       order, seg_off, sums);
}

// max over the batch rows of |x|^2 (fp32 wave sums: an upper-bound ingredient of the Flat / IVF-Flat certificate, not a
// score), folded into *out with an atomic max on the (non-negative) float's bit pattern.
__global__ __launch_bounds__(256) void k_max_norm2(const void* x, int x_f16, int64_t n, int d, unsigned int* out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    float s = 0.0f;
    for (int t = lane; t < d; t += 64) {
        const float v = x_f16 ? __half2float(((const __half*)x)[i * d + t]) : ((const float*)x)[i * d + t];
        s = __fmaf_rn(v, v, s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0 && s == s) atomicMax(out, __float_as_uint(s));
}
void launch_max_norm2(const void* x, int x_f16, int64_t n, int d, unsigned int* out_bits, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_max_norm2, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, x, x_f16, n, d, out_bits);
}

void launch_scatter_rows(const void* x, int x_f16, int64_t n, int d, const int64_t* dest_row, void* storage,
                         int storage_f16, int ld, float* norms, const int64_t* ids_in, int64_t id0,
                         int64_t* ids_storage, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, x, x_f16, n, d, dest_row,
                       storage, storage_f16, ld, norms, ids_in, id0, ids_storage);
}

// ---------------------------------------------------------------------------------------
// Re-layout on growth: list l moves from old_base[l] to new_base[l] (rows); data is copied in
// units of `unit_rows` rows = `unit_bytes` bytes (PQ: 64-row slab = 64*Mpad bytes; flat rows:
// 1 row = ld*elem bytes).  One workgroup per (list, stripe).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_copy_lists(int nlist, const int64_t* old_base, const int64_t* new_base,
                                                    const int64_t* len, const uint8_t* old_data, uint8_t* new_data,
                                                    int64_t unit_rows, int64_t unit_bytes, const int64_t* old_ids,
                                                    int64_t* new_ids, const float* old_norms, float* new_norms) {
    int l = blockIdx.x;
    int64_t n = len[l];
    if (n <= 0) return;
    int64_t units = (n + unit_rows - 1) / unit_rows;
    int64_t bytes = units * unit_bytes;
    const uint8_t* src = old_data + old_base[l] / unit_rows * unit_bytes;
    uint8_t* dst = new_data + new_base[l] / unit_rows * unit_bytes;
    // unit_bytes is a multiple of 16 for every layout used
    int64_t v16 = bytes / 16;
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < v16; i += (int64_t)gridDim.y * blockDim.x)
        ((uint4*)dst)[i] = ((const uint4*)src)[i];
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.y * blockDim.x) {
        if (old_ids) new_ids[new_base[l] + i] = old_ids[old_base[l] + i];
        if (old_norms) new_norms[new_base[l] + i] = old_norms[old_base[l] + i];
    }
}
void launch_copy_lists(int nlist, const int64_t* old_base, const int64_t* new_base, const int64_t* len,
                       const uint8_t* old_data, uint8_t* new_data, int64_t unit_rows, int64_t unit_bytes,
                       const int64_t* old_ids, int64_t* new_ids, const float* old_norms, float* new_norms,
                       hipStream_t st) {
    if (nlist <= 0) return;
    hipLaunchKernelGGL(k_copy_lists, dim3((unsigned)nlist, 8), dim3(256), 0, st, nlist, old_base, new_base, len,
                       old_data, new_data, unit_rows, unit_bytes, old_ids, new_ids, old_norms, new_norms);
}

// ---------------------------------------------------------------------------------------
// PQ slab layout <-> plain [n, M].  A slab is 64 consecutive vectors of one list:
//   byte(slab s, granule g, lane v, b) = codes[(s*Mpad/CB + g) * 64*CB + v*CB + b],  m = g*CB + b
// so that the scan's lane v reads CB contiguous bytes and the wave reads 64*CB contiguous bytes.
// ---------------------------------------------------------------------------------------
// (CB = 0: the rotated 16-vector-block layout — both are pq_code_addr in rsx_internal.h)
__device__ inline int64_t pq_byte_addr(int64_t row, int m, int Mpad, int CB) { return pq_code_addr(row, m, Mpad, CB); }
__global__ void k_pq_export(const uint8_t* codes, int64_t base_row, int64_t n, int M, int Mpad, int CB, uint8_t* out) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = n * M;
    for (; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e / M; int m = (int)(e - r * M);
        out[e] = codes[pq_byte_addr(base_row + r, m, Mpad, CB)];
    }
}
void launch_pq_export_list(const uint8_t* codes, int64_t base_row, int64_t n, int M, int Mpad, int CB,
                           uint8_t* out, hipStream_t st) {
    int64_t total = n * M;
    if (total <= 0) return;
    int64_t blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_pq_export, dim3((unsigned)blocks), dim3(256), 0, st, codes, base_row, n, M, Mpad, CB, out);
}
__global__ void k_pq_import(const uint8_t* plain, int64_t base_row, int64_t pos0, int64_t n, int M, int Mpad, int CB,
                            uint8_t* codes) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = n * Mpad;
    for (; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e / Mpad; int m = (int)(e - r * Mpad);
        codes[pq_byte_addr(base_row + pos0 + r, m, Mpad, CB)] = (m < M) ? plain[r * M + m] : (uint8_t)0;
    }
}
void launch_pq_import_list(const uint8_t* plain, int64_t base_row, int64_t pos0, int64_t n, int M, int Mpad,
                           int CB, uint8_t* codes, hipStream_t st) {
    int64_t total = n * Mpad;
    if (total <= 0) return;
    int64_t blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_pq_import, dim3((unsigned)blocks), dim3(256), 0, st, plain, base_row, pos0, n, M, Mpad, CB, codes);
}

// ---------------------------------------------------------------------------------------
// small helpers for the host layer
// ---------------------------------------------------------------------------------------
__global__ void k_check_f16(const float* x, int64_t count, int* flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    for (; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        float v = x[i];
        if (!(__half2float(__float2half_rn(v)) == v)) bad = true;
    }
    if (bad) atomicOr(flag, 1);
}
void launch_check_f16(const float* x, int64_t count, int* flag, hipStream_t st) {
    if (count <= 0) return;
    int64_t blocks = (count + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_check_f16, dim3((unsigned)blocks), dim3(256), 0, st, x, count, flag);
}
__global__ void k_write_ids(const int64_t* dest_row, const int64_t* ids_in, int64_t id0, int64_t n, int64_t* ids_storage) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && dest_row[i] >= 0) ids_storage[dest_row[i]] = ids_in ? ids_in[i] : id0 + i;   // dest < 0: dropped (other shard's list)
}
void launch_write_ids(const int64_t* dest_row, const int64_t* ids_in, int64_t id0, int64_t n, int64_t* ids_storage, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_write_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dest_row, ids_in, id0, n, ids_storage);
}
// widen an fp16 row storage to fp32 in a new buffer (storage upgrade when a non-fp16 value arrives)
__global__ void k_widen_storage(const __half* src, float* dst, int64_t count) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < count; i += (int64_t)gridDim.x * blockDim.x) dst[i] = __half2float(src[i]);
}
void launch_widen_storage(const __half* src, float* dst, int64_t count, hipStream_t st) {
    if (count <= 0) return;
    int64_t blocks = (count + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_widen_storage, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, count);
}
// residuals r = x - centroid[assign] in fp32 (training only)
__global__ void k_residuals(const float* x, int64_t n, int d, const float* centroids, const int32_t* assign, float* out) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = n * d;
    for (; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e / d; int t = (int)(e - r * d);
        out[e] = __fsub_rn(x[e], centroids[(int64_t)assign[r] * d + t]);
    }
}
void launch_residuals(const float* x, int64_t n, int d, const float* centroids, const int32_t* assign, float* out, hipStream_t st) {
    int64_t total = n * d;
    if (total <= 0) return;
    int64_t blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_residuals, dim3((unsigned)blocks), dim3(256), 0, st, x, n, d, centroids, assign, out);
}

}  // namespace rsx


// ---------------------------------------------------------------------------------------
// Destination rows of an add batch, computed on the device (round 3; reference: InvertedLists::add_entries appends in
// arrival order — src/indicies/ivf_flat.py:180, ivf_pq.py:185).  dest[i] = first free row of list a[i] + the number of earlier
// rows of the batch assigned to the same list: a STABLE rank, so the in-list order is the insertion order whatever the
// launch geometry.  Three small kernels around one 4-byte-per-list read-back (the host must see the per-list totals to grow
// the lists) instead of the assignments going to the host and a row table coming back (12 bytes per vector):
//   k_add_count : segment g (one wave) -> cnt[g][l] = rows of the segment assigned to list l (kept lists only)
//   k_add_scan  : per list, exclusive scan over the segments (in place) and the batch total
//   k_add_place : segment g walks its rows 64 at a time; LDS holds the running first-free row of every list
// Lists that are not this shard's (l % mod != rem) get dest = -1.
// ---------------------------------------------------------------------------------------
namespace rsx {
constexpr int ADD_SEG = 1024;      // rows per segment

__global__ __launch_bounds__(64) void k_add_count(const int32_t* assign, int64_t n, int nlist, int lmod, int lrem, int32_t* cnt) {
    extern __shared__ int32_t ac_lds[];
    const int lane = threadIdx.x;
    const int64_t g = blockIdx.x;
    for (int l = lane; l < nlist; l += 64) ac_lds[l] = 0;
    __syncthreads();
    const int64_t r0 = g * ADD_SEG;
    for (int64_t i = r0 + lane; i < r0 + ADD_SEG && i < n; i += 64) {
        const int32_t l = assign[i];
        if (lmod <= 1 || l % lmod == lrem) atomicAdd(&ac_lds[l], 1);
    }
    __syncthreads();
    for (int l = lane; l < nlist; l += 64) cnt[g * nlist + l] = ac_lds[l];
}
__global__ void k_add_scan(int32_t* cnt, int64_t nseg, int nlist, int32_t* total) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlist) return;
    int32_t run = 0;
    for (int64_t g = 0; g < nseg; g++) { const int32_t c = cnt[g * nlist + l]; cnt[g * nlist + l] = run; run += c; }
    total[l] = run;
}
__global__ __launch_bounds__(64) void k_add_place(const int32_t* assign, int64_t n, int nlist, int lmod, int lrem, const int32_t* off,
                                                  const int64_t* start, int64_t* dest) {
    extern __shared__ int32_t ap_lds[];      // running offset (rows of this batch already placed) per list
    const int lane = threadIdx.x;
    const int64_t g = blockIdx.x;
    for (int l = lane; l < nlist; l += 64) ap_lds[l] = off[g * nlist + l];
    __syncthreads();
    const int64_t r0 = g * ADD_SEG;
    for (int64_t c0 = r0; c0 < r0 + ADD_SEG && c0 < n; c0 += 64) {
        const int64_t i = c0 + lane;
        const bool in = i < n;
        const int32_t l = in ? assign[i] : -1;
        const bool keep = in && (lmod <= 1 || l % lmod == lrem);
        // stable rank among the lanes of this chunk with the same list: walk the distinct lists, lowest lane first
        int rank = 0, base = 0;
        uint64_t todo = __builtin_amdgcn_ballot_w64(keep);
        while (todo) {
            const int lead = __builtin_ctzll(todo);
            const int32_t lk = __builtin_amdgcn_readlane(l, lead);
            const uint64_t same = __builtin_amdgcn_ballot_w64(keep && l == lk);
            if (keep && l == lk) {
                rank = __builtin_popcountll(same & ((1ull << lane) - 1ull));
                base = ap_lds[lk];
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == lead) ap_lds[lk] = base + __builtin_popcountll(same);
            __builtin_amdgcn_wave_barrier();
            todo &= ~same;
        }
        if (in) dest[i] = keep ? start[l] + base + rank : -1;
    }
}
void launch_add_destinations(const int32_t* assign, int64_t n, int nlist, int lmod, int lrem, int32_t* seg_cnt /* [nseg][nlist] */,
                             int32_t* total /* [nlist] */, hipStream_t st) {
    const int64_t nseg = (n + ADD_SEG - 1) / ADD_SEG;
    hipLaunchKernelGGL(k_add_count, dim3((unsigned)nseg), dim3(64), (size_t)nlist * 4, st, assign, n, nlist, lmod, lrem, seg_cnt);
    hipLaunchKernelGGL(k_add_scan, dim3((unsigned)((nlist + 255) / 256)), dim3(256), 0, st, seg_cnt, nseg, nlist, total);
}
void launch_add_place(const int32_t* assign, int64_t n, int nlist, int lmod, int lrem, const int32_t* seg_off, const int64_t* start,
                      int64_t* dest, hipStream_t st) {
    const int64_t nseg = (n + ADD_SEG - 1) / ADD_SEG;
    hipLaunchKernelGGL(k_add_place, dim3((unsigned)nseg), dim3(64), (size_t)nlist * 4, st, assign, n, nlist, lmod, lrem, seg_off, start, dest);
}
int64_t add_dest_segments(int64_t n) { return (n + ADD_SEG - 1) / ADD_SEG; }
}  // namespace rsx
