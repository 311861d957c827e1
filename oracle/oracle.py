"""CPU oracle bindings — TEST INFRASTRUCTURE ONLY (see rsx_oracle.c header).

Two tiers:
  * Tier 0 (numpy fp64 brute force, this file): defines *truth* for Flat / IVF-Flat.
  * Tier 1 (rsx_oracle.c via ctypes): the restated FAISS-CPU algorithms (coarse quantiser,
    IVF scan, PQ look-up tables, ADC, k-means, PQ training/encoding, shard merge).

PARITY UNPINNED: FAISS 1.8.0 (the reference's arithmetic) is neither vendored in the reference
nor installable here, and the reference has no tests for this path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_f32p = ctypes.POINTER(ctypes.c_float)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_u16p = ctypes.POINTER(ctypes.c_uint16)


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "rsx_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_kmeans_sample.restype = ctypes.c_int64
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------- Tier 0 (numpy fp64)
def np_flat_search(xq, xb, k, metric="ip", ids=None):
    """Exact brute force in fp64; (score desc | dist asc, id asc); pads with -1/∓inf."""
    xq64 = np.asarray(xq, dtype=np.float64)
    xb64 = np.asarray(xb, dtype=np.float64)
    nb = xb64.shape[0]
    ids = np.arange(nb, dtype=np.int64) if ids is None else np.asarray(ids, dtype=np.int64)
    if metric == "ip":
        s = (xq64 @ xb64.T).astype(np.float32)
        key = -s.astype(np.float64)
    else:
        s = ((xq64[:, None, :] - xb64[None, :, :]) ** 2).sum(-1).astype(np.float32)
        key = s.astype(np.float64)
    nq = xq64.shape[0]
    D = np.full((nq, k), -np.inf if metric == "ip" else np.inf, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        order = np.lexsort((ids, key[q]))[:k]
        D[q, : len(order)] = s[q, order]
        I[q, : len(order)] = ids[order]
    return D, I


# ---------------------------------------------------------------- Tier 1 (C)
def half_to_float(h):
    h = np.ascontiguousarray(h).view(np.uint16)
    out = np.empty(h.shape, dtype=np.float32)
    lib().orc_half_to_float(ctypes.c_int64(h.size), _p(h, c_u16p), _p(out, c_f32p))
    return out


def synth_vectors(d, ncentres, seed_c, seed_x, sigma, i0, n):
    out = np.empty((n, d), dtype=np.uint16)
    lib().orc_synth_vectors(d, ncentres, ctypes.c_uint32(seed_c), ctypes.c_uint32(seed_x),
                            ctypes.c_float(sigma), ctypes.c_int64(i0), ctypes.c_int64(n), _p(out, c_u16p))
    return out.view(np.float16)


def synth_queries(d, ncentres, seed_c, seed_x, sigma, nbase, seed_q, sigma_q, r0, n):
    out = np.empty((n, d), dtype=np.uint16)
    lib().orc_synth_queries(d, ncentres, ctypes.c_uint32(seed_c), ctypes.c_uint32(seed_x),
                            ctypes.c_float(sigma), ctypes.c_int64(nbase), ctypes.c_uint32(seed_q),
                            ctypes.c_float(sigma_q), ctypes.c_int64(r0), ctypes.c_int64(n), _p(out, c_u16p))
    return out.view(np.float16)


def flat_search(xq, xb, k, metric=0, ids=None):
    xq, xb = _f32(xq), _f32(xb)
    nq, d = xq.shape
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
    lib().orc_flat_search(metric, d, ctypes.c_int64(xb.shape[0]), _p(xb, c_f32p), _p(ids, c_i64p),
                          ctypes.c_int64(nq), _p(xq, c_f32p), k, _p(D, c_f32p), _p(I, c_i64p))
    return D, I


def assign_ip(centroids, x):
    centroids, x = _f32(centroids), _f32(x)
    n, d = x.shape
    a = np.empty(n, dtype=np.int32)
    best = np.empty(n, dtype=np.float32)
    lib().orc_assign_ip(d, centroids.shape[0], _p(centroids, c_f32p), ctypes.c_int64(n), _p(x, c_f32p),
                        _p(a, c_i32p), _p(best, c_f32p))
    return a, best


def coarse_probe(centroids, xq, nprobe):
    centroids, xq = _f32(centroids), _f32(xq)
    nq, d = xq.shape
    pid = np.empty((nq, nprobe), dtype=np.int64)
    ps = np.empty((nq, nprobe), dtype=np.float32)
    lib().orc_coarse_probe(d, centroids.shape[0], _p(centroids, c_f32p), ctypes.c_int64(nq), _p(xq, c_f32p),
                           nprobe, _p(pid, c_i64p), _p(ps, c_f32p))
    return pid, ps


def residuals(centroids, x, assign):
    centroids, x = _f32(centroids), _f32(x)
    assign = np.ascontiguousarray(assign, dtype=np.int32)
    out = np.empty_like(x)
    lib().orc_residuals(x.shape[1], _p(centroids, c_f32p), ctypes.c_int64(x.shape[0]), _p(x, c_f32p),
                        _p(assign, c_i32p), _p(out, c_f32p))
    return out


def pq_encode(codebooks, x):
    codebooks, x = _f32(codebooks), _f32(x)
    M = codebooks.shape[0]
    n, d = x.shape
    codes = np.empty((n, M), dtype=np.uint8)
    lib().orc_pq_encode(d, M, _p(codebooks, c_f32p), ctypes.c_int64(n), _p(x, c_f32p), _p(codes, c_u8p))
    return codes


def pq_lut(codebooks, xq):
    codebooks, xq = _f32(codebooks), _f32(xq)
    M = codebooks.shape[0]
    nq, d = xq.shape
    T = np.empty((nq, M, 256), dtype=np.float32)
    lib().orc_pq_lut(d, M, _p(codebooks, c_f32p), ctypes.c_int64(nq), _p(xq, c_f32p), _p(T, c_f32p))
    return T


class ListMajor:
    """Inverted lists flattened list-major, the layout orc_ivf*_search consume."""

    def __init__(self, assign, ids, payload, nlist):
        assign = np.asarray(assign)
        order = np.argsort(assign, kind="stable")  # insertion order inside each list
        counts = np.bincount(assign, minlength=nlist)
        self.list_off = np.zeros(nlist + 1, dtype=np.int64)
        np.cumsum(counts, out=self.list_off[1:])
        self.ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64)[order])
        self.payload = np.ascontiguousarray(payload[order])
        self.order = order


def ivfflat_search(metric, centroids, lm, xq, nprobe, k):
    centroids, xq = _f32(centroids), _f32(xq)
    vecs = _f32(lm.payload)
    nq, d = xq.shape
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    lib().orc_ivfflat_search(metric, d, centroids.shape[0], _p(centroids, c_f32p), _p(lm.list_off, c_i64p),
                             _p(vecs, c_f32p), _p(lm.ids, c_i64p), ctypes.c_int64(nq), _p(xq, c_f32p),
                             nprobe, k, _p(D, c_f32p), _p(I, c_i64p))
    return D, I


def ivfpq_search(centroids, codebooks, lm, xq, nprobe, k, heap=False, metric=0):
    """metric 0: inner product (what the reference builds); 1: squared L2 to the decoded vector (orc_ivfpq_search_l2)."""
    centroids, codebooks, xq = _f32(centroids), _f32(codebooks), _f32(xq)
    codes = np.ascontiguousarray(lm.payload, dtype=np.uint8)
    nq, d = xq.shape
    M = codebooks.shape[0]
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    fn = lib().orc_ivfpq_search_l2 if metric else lib().orc_ivfpq_search_heap if heap else lib().orc_ivfpq_search
    fn(d, centroids.shape[0], M, _p(centroids, c_f32p), _p(codebooks, c_f32p), _p(lm.list_off, c_i64p),
       _p(codes, c_u8p), _p(lm.ids, c_i64p), ctypes.c_int64(nq), _p(xq, c_f32p), nprobe, k,
       _p(D, c_f32p), _p(I, c_i64p))
    return D, I


def kmeans_sample(n, k, mppc, seed):
    sel = np.empty(min(n, k * mppc) if n > k * mppc else n, dtype=np.int64)
    buf = np.empty(max(n, 1), dtype=np.int64)
    kept = lib().orc_kmeans_sample(ctypes.c_int64(n), k, mppc, ctypes.c_uint64(seed), _p(buf, c_i64p))
    sel[:] = buf[:kept]
    return sel


def kmeans(mode, x, k, niter, seed):
    x = _f32(x)
    n, d = x.shape
    c = np.empty((k, d), dtype=np.float32)
    lib().orc_kmeans(mode, d, k, ctypes.c_int64(n), _p(x, c_f32p), niter, ctypes.c_uint64(seed), _p(c, c_f32p))
    return c


def pq_train(x, M, niter, seed):
    x = _f32(x)
    n, d = x.shape
    cb = np.empty((M, 256, d // M), dtype=np.float32)
    lib().orc_pq_train(d, M, ctypes.c_int64(n), _p(x, c_f32p), niter, ctypes.c_uint64(seed), _p(cb, c_f32p))
    return cb


def merge_topk(D, I, metric=0):
    D = np.ascontiguousarray(D, dtype=np.float32)
    I = np.ascontiguousarray(I, dtype=np.int64)
    ns, nq, k = D.shape
    Do = np.empty((nq, k), dtype=np.float32)
    Io = np.empty((nq, k), dtype=np.int64)
    lib().orc_merge_topk(ns, ctypes.c_int64(nq), k, metric, _p(D, c_f32p), _p(I, c_i64p), _p(Do, c_f32p), _p(Io, c_i64p))
    return Do, Io


def num_threads():
    return int(lib().orc_num_threads())
