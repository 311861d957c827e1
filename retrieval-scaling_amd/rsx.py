"""rsx — Python host binding of librsx.so (include/rsx.h), the MI355X dense-retrieval engine.

The module exposes the subset of the `faiss` Python surface that the reference's search path
uses (every call site listed in SURVEY.md §2.2), with the same names, argument meaning and error
behaviour, so `src/indicies/*.py` is written against `rsx` exactly as the reference's files are
written against `faiss`:

    reference                                          here
    faiss.IndexFlatIP(d)            flat.py:42         rsx.IndexFlatIP(d)
    faiss.IndexIVFFlat(q,d,nlist,METRIC_INNER_PRODUCT) ivf_flat.py:144   rsx.IndexIVFFlat(...)
    faiss.IndexIVFPQ(q,d,nlist,M,nbits,METRIC_IP)      ivf_pq.py:147     rsx.IndexIVFPQ(...)
    index.train / add / search / nprobe / ntotal / is_trained            same
    faiss.read_index / write_index                                        rsx.read_index / write_index
    StandardGpuResources / GpuClonerOptions / index_cpu_to_gpu / index_gpu_to_cpu
        (ivf_flat.py:155-163: CUDA-only training clone)                   identity shims — the index
                                                                          already lives on the GPU

All arithmetic happens in the HIP library.  There is NO CPU fallback: importing works without a
GPU (so host logic can be tested), but constructing an index raises RuntimeError unless librsx.so
is built and a HIP device is present.
"""
import ctypes
import os
import subprocess

import numpy as np

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1

_F32, _F16 = 0, 1
_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIB_PATH = os.environ.get("RSX_LIB", os.path.join(_CSRC, "librsx.so"))   # RSX_LIB: another build, for A/B measurements
_lib = None

#: every symbol include/rsx.h declares (checked by tests/test_abi.py against the header)
ABI_SYMBOLS = [
    "rsx_last_error", "rsx_version", "rsx_device_count", "rsx_flat_create", "rsx_ivfflat_create",
    "rsx_ivfpq_create", "rsx_sharded_create", "rsx_load_sharded", "rsx_destroy", "rsx_train", "rsx_set_centroids", "rsx_set_codebooks",
    "rsx_get_centroids", "rsx_get_codebooks", "rsx_add", "rsx_assign", "rsx_reset", "rsx_reserve_lists", "rsx_add_list",
    "rsx_get_list", "rsx_get_list_sizes", "rsx_set_nprobe", "rsx_search", "rsx_search_prepass", "rsx_search_scan", "rsx_merge_topk", "rsx_pack_topk", "rsx_merge_packed", "rsx_get",
    "rsx_set_param",
    "rsx_get_timing", "rsx_save", "rsx_load", "rsx_synth_vectors", "rsx_synth_queries",
]


def build(force=False, verbose=False):
    """Compile librsx.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", _CSRC, "-j8"]
    if force:
        args.append("-B")
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(args, stdout=out)
    return _LIB_PATH


def lib():
    """The loaded C-ABI library; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the search path)")
        # PyTorch-ROCm ships its own libamdhip64: whichever copy is mapped FIRST serves the whole process (same soname), and a
        # process that maps /opt/rocm's through librsx and torch's afterwards ends up with two HIP runtimes — torch then
        # reports "No HIP GPUs are available".  Load torch's first whenever torch is installed.
        if not os.environ.get("RSX_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = ctypes.CDLL(_LIB_PATH)
        L.rsx_last_error.restype = ctypes.c_char_p
        for name in ABI_SYMBOLS:
            getattr(L, name)  # AttributeError if the library does not export it
        _lib = L
    return _lib


def _check(status):
    if status != 0:
        msg = lib().rsx_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"rsx error {status}: {msg}")


def get_num_gpus():
    n = ctypes.c_int(0)
    st = lib().rsx_device_count(ctypes.byref(n))
    return n.value if st == 0 else 0


def default_device():
    for key in ("RSX_DEVICE", "LOCAL_RANK"):
        if key in os.environ:
            return int(os.environ[key])
    return 0


_default_devices = None


def set_default_devices(devices):
    """Devices new indexes span when no `device` / `devices` argument is given: a list of ordinals (one shard per
    entry, single process — rsx_sharded_create), "all", or None for one GPU.  The Indexer facade sets this from
    cfg.datastore.index.devices; the RSX_DEVICES environment variable ("0,1,2,3" or "all") is the fallback."""
    global _default_devices
    _default_devices = devices


def get_default_devices():
    return _default_devices


def _resolve_devices(device, devices):
    """-> None (single-GPU handle on `device`) or a list of device ordinals (sharded handle)."""
    if devices is None and device is None:
        devices = _default_devices
        if devices is None and os.environ.get("RSX_DEVICES"):
            devices = os.environ["RSX_DEVICES"]
    if devices is None:
        return None
    if isinstance(devices, str):
        devices = list(range(get_num_gpus())) if devices.strip().lower() == "all" else [int(t) for t in devices.split(",") if t.strip()]
    devices = [int(t) for t in devices]
    return devices or None


def _create(kind, d, nlist, M, nbits, metric, device, devices):
    h = ctypes.c_void_p()
    devs = _resolve_devices(device, devices)
    if devs is not None and len(devs) > 1:
        arr = (ctypes.c_int * len(devs))(*devs)
        _check(lib().rsx_sharded_create(kind, int(d), int(nlist), int(M), int(nbits), int(metric), len(devs), arr, ctypes.byref(h)))
        return h
    if devs is not None:           # a one-entry list ("2", [2]) names the device of an ordinary handle
        device = devs[0]
    dev = default_device() if device is None else int(device)
    if kind == 0:
        _check(lib().rsx_flat_create(int(d), int(metric), dev, ctypes.byref(h)))
    elif kind == 1:
        _check(lib().rsx_ivfflat_create(int(d), int(nlist), int(metric), dev, ctypes.byref(h)))
    else:
        _check(lib().rsx_ivfpq_create(int(d), int(nlist), int(M), int(nbits), int(metric), dev, ctypes.byref(h)))
    return h


# ------------------------------------------------------------------------------------------
# array plumbing: numpy (host) or torch CUDA tensors (HBM) -> raw pointers
# ------------------------------------------------------------------------------------------
def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


def _as_matrix(x, d, what):
    """-> (keepalive, pointer, n, dtype_code, on_device).  Mirrors faiss's SWIG wrapper: 2-D,
    C-contiguous, second dimension d; fp32 (FAISS) or fp16 (kept as is — no up-cast needed)."""
    if _is_torch(x):
        import torch
        if x.dim() != 2 or x.shape[1] != d:
            raise AssertionError(f"{what}: expected shape [n, {d}], got {tuple(x.shape)}")
        if x.dtype not in (torch.float32, torch.float16):
            x = x.float()
        x = x.contiguous()
        return x, ctypes.c_void_p(x.data_ptr()), x.shape[0], (_F16 if x.dtype == torch.float16 else _F32), x.is_cuda
    x = np.asarray(x)
    if x.ndim != 2 or x.shape[1] != d:
        raise AssertionError(f"{what}: expected shape [n, {d}], got {x.shape}")
    if x.dtype not in (np.float32, np.float16):
        x = x.astype(np.float32)
    x = np.ascontiguousarray(x)
    return x, ctypes.c_void_p(x.ctypes.data), x.shape[0], (_F16 if x.dtype == np.float16 else _F32), False


def _sync_producer(keep, on_dev):
    """The library reads caller data on its OWN (non-blocking) stream: a CUDA tensor must be complete before the
    call — including the .float()/.contiguous() copies _as_matrix may have enqueued on torch's current stream."""
    if on_dev:
        import torch
        torch.cuda.current_stream(keep.device).synchronize()


class Index:
    """Base handle.  Attributes follow faiss.Index: d, ntotal, is_trained, metric_type."""

    @property
    def nshards(self):
        """0 for a single-GPU handle, else the number of per-device shards behind this (single-process) handle."""
        return self._get("nshards")

    def __init__(self, handle, d, metric):
        self._h = handle
        self.d = d
        self.metric_type = metric

    # -- lifetime
    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and _lib is not None:
            try:
                _lib.rsx_destroy(h)
            except Exception:
                pass
            self._h = None

    # -- properties
    def _get(self, key):
        v = ctypes.c_int64(0)
        _check(lib().rsx_get(self._h, key.encode(), ctypes.byref(v)))
        return v.value

    @property
    def ntotal(self):
        return self._get("ntotal")

    @property
    def is_trained(self):
        return bool(self._get("is_trained"))

    @property
    def storage_dtype(self):
        return {0: "float32", 1: "float16", -1: "pq"}[self._get("storage_dtype")]

    def set_param(self, key, value):
        _check(lib().rsx_set_param(self._h, key.encode(), ctypes.c_double(value)))

    def get_timing(self, key):
        v = ctypes.c_double(0)
        _check(lib().rsx_get_timing(self._h, key.encode(), ctypes.byref(v)))
        return v.value

    # -- faiss.Index API
    def train(self, x):
        keep, p, n, dt, on_dev = _as_matrix(x, self.d, "train")
        _sync_producer(keep, on_dev)
        _check(lib().rsx_train(self._h, ctypes.c_int64(n), p, dt))

    def add(self, x):
        keep, p, n, dt, on_dev = _as_matrix(x, self.d, "add")
        _sync_producer(keep, on_dev)
        _check(lib().rsx_add(self._h, ctypes.c_int64(n), p, dt, None))

    def reset(self):
        _check(lib().rsx_reset(self._h))

    def add_with_ids(self, x, ids):
        keep, p, n, dt, on_dev = _as_matrix(x, self.d, "add_with_ids")
        _sync_producer(keep, on_dev)
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        assert ids.shape == (n,), "ids must have one entry per vector"
        _check(lib().rsx_add(self._h, ctypes.c_int64(n), p, dt, ids.ctypes.data_as(ctypes.c_void_p)))

    def search(self, x, k):
        """-> (D float32 [n,k], I int64 [n,k]); numpy in -> numpy out, CUDA tensor in -> CUDA tensors out."""
        keep, p, n, dt, on_dev = _as_matrix(x, self.d, "search")
        k = int(k)
        assert k > 0, "k must be positive"
        if on_dev:
            import torch
            D = torch.empty((n, k), dtype=torch.float32, device=keep.device)
            I = torch.empty((n, k), dtype=torch.int64, device=keep.device)
            _sync_producer(keep, on_dev)
            _check(lib().rsx_search(self._h, ctypes.c_int64(n), p, dt, k, ctypes.c_void_p(D.data_ptr()),
                                    ctypes.c_void_p(I.data_ptr())))
            return D, I
        D = np.empty((n, k), dtype=np.float32)
        I = np.empty((n, k), dtype=np.int64)
        _check(lib().rsx_search(self._h, ctypes.c_int64(n), p, dt, k, D.ctypes.data_as(ctypes.c_void_p),
                                I.ctypes.data_as(ctypes.c_void_p)))
        return D, I

    def search_prepass(self, x, k):
        """First half of a two-call search (rsx_search_prepass): CUDA-tensor queries in -> a CUDA int64 tensor VIEW of the nq
        threshold keys (or None when this search has none).  The keys are unsigned 64-bit integers: flip the sign bit
        (`t ^= -2**63`) before comparing / reducing them as int64 and flip it back afterwards.  Finish with search_scan()."""
        import torch
        keep, p, n, dt, on_dev = _as_matrix(x, self.d, "search")
        assert on_dev, "search_prepass takes CUDA-tensor queries (the thresholds live in HBM)"
        k = int(k)
        D = torch.empty((n, k), dtype=torch.float32, device=keep.device)
        I = torch.empty((n, k), dtype=torch.int64, device=keep.device)
        _sync_producer(keep, on_dev)
        tau, ntau = ctypes.c_void_p(), ctypes.c_int64(0)
        _check(lib().rsx_search_prepass(self._h, ctypes.c_int64(n), p, dt, k, ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()),
                                        ctypes.byref(tau), ctypes.byref(ntau)))
        self._two_call = (keep, D, I)
        if not tau.value or ntau.value == 0:
            return None

        class _View:      # zero-copy view of library memory through the CUDA array interface
            __cuda_array_interface__ = {"shape": (ntau.value,), "typestr": "<i8", "data": (tau.value, False), "version": 2}
        return torch.as_tensor(_View(), device=keep.device)

    def search_scan(self):
        """Second half: runs the scan with the (possibly raised) thresholds -> (D, I) CUDA tensors."""
        keep, D, I = self._two_call
        self._two_call = None
        _check(lib().rsx_search_scan(self._h))
        return D, I

    # -- inspection (parity tests, writer)
    def get_list(self, list_no=0):
        """-> (payload, ids): IVFPQ payload = codes uint8 [n, M]; Flat/IVFFlat payload = float32 [n, d]."""
        n = ctypes.c_int64(0)
        _check(lib().rsx_get_list(self._h, ctypes.c_int64(list_no), ctypes.byref(n), None, None))
        n = n.value
        if self._get("kind") == 2:
            payload = np.empty((n, self._get("M")), dtype=np.uint8)
        else:
            payload = np.empty((n, self.d), dtype=np.float32)
        ids = np.empty(n, dtype=np.int64)
        if n:
            _check(lib().rsx_get_list(self._h, ctypes.c_int64(list_no), None, payload.ctypes.data_as(ctypes.c_void_p),
                                      ids.ctypes.data_as(ctypes.c_void_p)))
        return payload, ids

    def list_sizes(self):
        """int64 [nlist]: vectors per inverted list (Flat: one entry = ntotal) — the bench's list-length histogram."""
        out = np.zeros(max(1, self._get("nlist")), dtype=np.int64)
        _check(lib().rsx_get_list_sizes(self._h, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def reconstruct_n(self, i0, n):
        assert self._get("kind") == 0, "reconstruct_n is implemented for Flat indexes"
        payload, _ = self.get_list(0)
        return payload[i0:i0 + n]


class IndexFlat(Index):
    def __init__(self, d, metric=METRIC_L2, device=None, devices=None):
        super().__init__(_create(0, d, 1, 0, 8, metric, device, devices), int(d), int(metric))


class IndexFlatIP(IndexFlat):
    def __init__(self, d, device=None, devices=None):
        super().__init__(d, METRIC_INNER_PRODUCT, device, devices)


class IndexFlatL2(IndexFlat):
    def __init__(self, d, device=None, devices=None):
        super().__init__(d, METRIC_L2, device, devices)


class _IndexIVF(Index):
    @property
    def nprobe(self):
        return self._get("nprobe")

    @nprobe.setter
    def nprobe(self, v):
        _check(lib().rsx_set_nprobe(self._h, int(v)))

    @property
    def nlist(self):
        return self._get("nlist")

    def set_centroids(self, c):
        c = np.ascontiguousarray(c, dtype=np.float32)
        assert c.shape == (self.nlist, self.d)
        _check(lib().rsx_set_centroids(self._h, c.ctypes.data_as(ctypes.c_void_p)))

    def get_centroids(self):
        c = np.empty((self.nlist, self.d), dtype=np.float32)
        _check(lib().rsx_get_centroids(self._h, c.ctypes.data_as(ctypes.c_void_p)))
        return c

    def assign(self, x):
        """index.quantizer.assign(x): list number of every vector (exact argmax-IP, as `add` computes it)."""
        keep, p, n, dt, on_dev = _as_matrix(x, self.d, "assign")
        _sync_producer(keep, on_dev)
        labels = np.empty(n, dtype=np.int64)
        _check(lib().rsx_assign(self._h, ctypes.c_int64(n), p, dt, labels.ctypes.data_as(ctypes.c_void_p)))
        return labels

    def reserve_lists(self, counts):
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        assert counts.shape == (self.nlist,)
        _check(lib().rsx_reserve_lists(self._h, counts.ctypes.data_as(ctypes.c_void_p)))

    def add_list(self, list_no, payload, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        if self._get("kind") == 2:
            payload = np.ascontiguousarray(payload, dtype=np.uint8)
            dt = _F32
        else:
            payload = np.ascontiguousarray(payload)
            if payload.dtype not in (np.float32, np.float16):
                payload = payload.astype(np.float32)
            dt = _F16 if payload.dtype == np.float16 else _F32
        assert payload.shape[0] == ids.shape[0]
        _check(lib().rsx_add_list(self._h, ctypes.c_int64(list_no), ctypes.c_int64(ids.shape[0]),
                                  payload.ctypes.data_as(ctypes.c_void_p), dt, ids.ctypes.data_as(ctypes.c_void_p)))


def _quantizer_metric(quantizer, metric):
    # The reference passes faiss.IndexFlatIP(d) as the coarse quantiser (ivf_flat.py:143): its
    # metric decides the assignment rule.  The library folds the quantiser into the IVF handle.
    if quantizer is not None and getattr(quantizer, "metric_type", metric) != METRIC_INNER_PRODUCT:
        raise RuntimeError("rsx: only inner-product coarse quantisers are implemented (the reference uses IndexFlatIP)")


class IndexIVFFlat(_IndexIVF):
    def __init__(self, quantizer, d, nlist, metric=METRIC_L2, device=None, devices=None):
        _quantizer_metric(quantizer, metric)
        super().__init__(_create(1, d, nlist, 0, 8, metric, device, devices), int(d), int(metric))
        self.quantizer = quantizer


class IndexIVFPQ(_IndexIVF):
    def __init__(self, quantizer, d, nlist, M, nbits, metric=METRIC_L2, device=None, devices=None):
        _quantizer_metric(quantizer, metric)
        super().__init__(_create(2, d, nlist, M, nbits, metric, device, devices), int(d), int(metric))
        self.quantizer = quantizer
        self.M = int(M)
        self.nbits = int(nbits)

    def set_codebooks(self, cb):
        cb = np.ascontiguousarray(cb, dtype=np.float32)
        assert cb.shape == (self.M, 256, self.d // self.M)
        _check(lib().rsx_set_codebooks(self._h, cb.ctypes.data_as(ctypes.c_void_p)))

    def get_codebooks(self):
        cb = np.empty((self.M, 256, self.d // self.M), dtype=np.float32)
        _check(lib().rsx_get_codebooks(self._h, cb.ctypes.data_as(ctypes.c_void_p)))
        return cb


# ------------------------------------------------------------------------------------------
# persistence
# ------------------------------------------------------------------------------------------
def _wrap_handle(h):
    v = ctypes.c_int64(0)

    def g(key):
        _check(lib().rsx_get(h, key.encode(), ctypes.byref(v)))
        return v.value

    kind, d, metric = g("kind"), g("d"), g("metric")
    cls = {0: IndexFlat, 1: IndexIVFFlat, 2: IndexIVFPQ}[kind]
    obj = cls.__new__(cls)
    Index.__init__(obj, h, d, metric)
    if kind == 2:
        obj.M, obj.nbits = g("M"), g("nbits")
    if kind != 0:
        obj.quantizer = None
    return obj


def write_index(index, path):
    """faiss.write_index(index, path).  Default: the native RSX1 container (magic-tagged; rsx.read_index loads it whatever the
    file is called — the reference's mirrors keep its `index_*.faiss` names).  RSX_INDEX_FORMAT=faiss writes a FAISS-format file
    through rsx_faiss_io instead, for hand-over to a real faiss.read_index (written from upstream knowledge of the format,
    unverified against a FAISS-produced file: DESIGN.md)."""
    if os.environ.get("RSX_INDEX_FORMAT", "").lower() == "faiss" and index.nshards == 0:
        from rsx_faiss_io import write_faiss_index
        return write_faiss_index(index, os.fspath(path))
    _check(lib().rsx_save(index._h, os.fspath(path).encode()))


def read_index(path, device=None, devices=None):
    """faiss.read_index(path): loads an RSX1 container (RSXS: the manifest of a sharded index, one shard per device), or
    a FAISS-format file via rsx_faiss_io.

    A multi-device request (devices=[...] / "all", cfg.datastore.index.devices, RSX_DEVICES) on a file written from ONE
    index — RSX1 or FAISS — RE-SHARDS it while loading: every inverted list / row block is cut over the devices as it
    streams in (ids kept), so an index built on one GPU or handed over from FAISS is served by the whole node and answers
    exactly as before.  (write_index on the result saves the sharded form; the next load is then direct.)"""
    path = os.fspath(path)
    with open(path, "rb") as f:
        magic = f.read(4)
    devs = None if device is not None else _resolve_devices(None, devices)
    multi = devs is not None and len(devs) > 1
    if magic == b"RSXS" or (magic == b"RSX1" and multi):
        devs = devs or [default_device() if device is None else int(device)]
        arr = (ctypes.c_int * len(devs))(*devs)
        h = ctypes.c_void_p()
        _check(lib().rsx_load_sharded(path.encode(), len(devs), arr, ctypes.byref(h)))
        return _wrap_handle(h)
    if device is None and devs:        # a one-entry device list names the device of an ordinary handle
        device = devs[0]
    if magic != b"RSX1":
        from rsx_faiss_io import read_faiss_index
        return read_faiss_index(path, device=device, devices=devs if multi else None)
    h = ctypes.c_void_p()
    _check(lib().rsx_load(path.encode(), default_device() if device is None else int(device), ctypes.byref(h)))
    return _wrap_handle(h)


# ------------------------------------------------------------------------------------------
# multi-shard merge (src/search.py:362-367)
# ------------------------------------------------------------------------------------------
def merge_topk(D, I, metric=METRIC_INNER_PRODUCT, device=None):
    """D, I: [nshards, nq, k] (numpy or CUDA tensors) -> merged (D, I) [nq, k] on the GPU kernel."""
    dev = default_device() if device is None else int(device)
    if _is_torch(D):
        import torch
        D = D.contiguous().float()
        I = I.contiguous().to(torch.int64)
        ns, nq, k = D.shape
        if D.is_cuda:
            Do = torch.empty((nq, k), dtype=torch.float32, device=D.device)
            Io = torch.empty((nq, k), dtype=torch.int64, device=D.device)
            torch.cuda.current_stream(D.device).synchronize()
            _check(lib().rsx_merge_topk(ns, ctypes.c_int64(nq), k, int(metric), ctypes.c_void_p(D.data_ptr()),
                                        ctypes.c_void_p(I.data_ptr()), ctypes.c_void_p(Do.data_ptr()),
                                        ctypes.c_void_p(Io.data_ptr()), D.device.index if D.device.index is not None else dev))
            return Do, Io
        D, I = D.numpy(), I.numpy()
    D = np.ascontiguousarray(D, dtype=np.float32)
    I = np.ascontiguousarray(I, dtype=np.int64)
    ns, nq, k = D.shape
    Do = np.empty((nq, k), dtype=np.float32)
    Io = np.empty((nq, k), dtype=np.int64)
    _check(lib().rsx_merge_topk(ns, ctypes.c_int64(nq), k, int(metric), D.ctypes.data_as(ctypes.c_void_p),
                                I.ctypes.data_as(ctypes.c_void_p), Do.ctypes.data_as(ctypes.c_void_p),
                                Io.ctypes.data_as(ctypes.c_void_p), dev))
    return Do, Io


def pack_topk(D, I, id_offset=0):
    """One rank's CUDA (D, I) [nq, k] -> packed [2, nq, k] int64 (score bits | ids + id_offset) on torch's current stream."""
    import torch
    D = D.contiguous().float(); I = I.contiguous().to(torch.int64)
    nq, k = D.shape
    out = torch.empty((2, nq, k), dtype=torch.int64, device=D.device)
    _check(lib().rsx_pack_topk(ctypes.c_int64(nq), k, ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()),
                               ctypes.c_int64(int(id_offset)), ctypes.c_void_p(out.data_ptr()), D.device.index or 0,
                               ctypes.c_void_p(torch.cuda.current_stream(D.device).cuda_stream)))
    return out


def merge_packed(gathered, metric=METRIC_INNER_PRODUCT):
    """gathered: CUDA int64 [nshards, 2, nq, k] (an all-gather of pack_topk blocks) -> merged (D, I) [nq, k], same rule as
    merge_topk; runs on torch's current stream without a host synchronisation."""
    import torch
    gathered = gathered.contiguous()
    ns, two, nq, k = gathered.shape
    assert two == 2 and gathered.dtype == torch.int64 and gathered.is_cuda
    Do = torch.empty((nq, k), dtype=torch.float32, device=gathered.device)
    Io = torch.empty((nq, k), dtype=torch.int64, device=gathered.device)
    _check(lib().rsx_merge_packed(ns, ctypes.c_int64(nq), k, int(metric), ctypes.c_void_p(gathered.data_ptr()),
                                  ctypes.c_void_p(Do.data_ptr()), ctypes.c_void_p(Io.data_ptr()), gathered.device.index or 0,
                                  ctypes.c_void_p(torch.cuda.current_stream(gathered.device).cuda_stream)))
    return Do, Io


# ------------------------------------------------------------------------------------------
# synthetic data (bench / tests)
# ------------------------------------------------------------------------------------------
def synth_vectors(d, ncentres, seed_c, seed_x, sigma, i0, n, out=None, device=None):
    dev = default_device() if device is None else int(device)
    if out is not None and _is_torch(out):
        _check(lib().rsx_synth_vectors(dev, d, ncentres, ctypes.c_uint32(seed_c), ctypes.c_uint32(seed_x),
                                       ctypes.c_float(sigma), ctypes.c_int64(i0), ctypes.c_int64(n),
                                       ctypes.c_void_p(out.data_ptr())))
        return out
    arr = np.empty((n, d), dtype=np.float16)
    _check(lib().rsx_synth_vectors(dev, d, ncentres, ctypes.c_uint32(seed_c), ctypes.c_uint32(seed_x),
                                   ctypes.c_float(sigma), ctypes.c_int64(i0), ctypes.c_int64(n),
                                   arr.ctypes.data_as(ctypes.c_void_p)))
    return arr


def synth_queries(d, ncentres, seed_c, seed_x, sigma, nbase, seed_q, sigma_q, r0, n, out=None, device=None):
    dev = default_device() if device is None else int(device)
    if out is not None and _is_torch(out):
        _check(lib().rsx_synth_queries(dev, d, ncentres, ctypes.c_uint32(seed_c), ctypes.c_uint32(seed_x),
                                       ctypes.c_float(sigma), ctypes.c_int64(nbase), ctypes.c_uint32(seed_q),
                                       ctypes.c_float(sigma_q), ctypes.c_int64(r0), ctypes.c_int64(n),
                                       ctypes.c_void_p(out.data_ptr())))
        return out
    arr = np.empty((n, d), dtype=np.float16)
    _check(lib().rsx_synth_queries(dev, d, ncentres, ctypes.c_uint32(seed_c), ctypes.c_uint32(seed_x),
                                   ctypes.c_float(sigma), ctypes.c_int64(nbase), ctypes.c_uint32(seed_q),
                                   ctypes.c_float(sigma_q), ctypes.c_int64(r0), ctypes.c_int64(n),
                                   arr.ctypes.data_as(ctypes.c_void_p)))
    return arr


# ------------------------------------------------------------------------------------------
# CUDA-era shims the reference's training branch touches (ivf_flat.py:152-163)
# ------------------------------------------------------------------------------------------
class StandardGpuResources:
    """faiss.StandardGpuResources(): nothing to hold — librsx owns its HBM per index handle."""


class GpuClonerOptions:
    useFloat16 = False


def index_cpu_to_gpu(res, device, index, co=None):
    """The index is already resident on the MI355X; training runs there (rsx_train)."""
    return index


def index_gpu_to_cpu(index):
    return index
