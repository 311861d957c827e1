"""FAISS on-disk index format (.faiss) reader / writer for the three index types of the path.

Reference call sites: faiss.read_index / faiss.write_index in src/indicies/flat.py:39,63,69,
ivf_flat.py:71,167,170,185, ivf_pq.py:75,171,174,190 — and the released MassiveDS artefacts
(README.md:146-150) are files in this format.

The layout below restates faiss/impl/index_write.cpp (v1.8.0) from upstream knowledge; NO
FAISS-produced file is available offline, so byte-compatibility with real FAISS is UNVERIFIED
(round-trip and field-level tests only):

  header      : d i32 | ntotal i64 | dummy i64 (1<<20) | dummy i64 | is_trained u8 | metric i32
  IndexFlat   : "IxFI" (IP) / "IxF2" (L2) | header | n_words u64 (= bytes/4) | fp32 vectors
  ivf header  : header | nlist u64 | nprobe u64 | <quantizer index> | direct-map type u8 | array (u64 n + i64[n])
  IndexIVFFlat: "IwFl" | ivf header | inverted lists
  IndexIVFPQ  : "IwPQ" | ivf header | by_residual u8 | code_size u64 | PQ (d u64, M u64, nbits u64,
                centroids: u64 n + f32[n], laid out [M][256][dsub]) | inverted lists
  inv. lists  : "ilar" | nlist u64 | code_size u64 | "full" + sizes (u64 n + u64[n])
                                                   | "sprs" + (list, size) pairs (u64 n + u64[n])
                then per non-empty list: codes (size*code_size bytes) followed by ids (i64[size])

Parsing / serialising is pure numpy (testable without a GPU); index_from_parsed builds rsx indexes.
"""
import io
import struct

import numpy as np

METRIC_INNER_PRODUCT, METRIC_L2 = 0, 1


# ------------------------------------------------------------------------------------------ parse
class _R:
    def __init__(self, f):
        self.f = f

    def raw(self, n):
        b = self.f.read(n)
        if len(b) != n:
            raise RuntimeError("truncated FAISS index file")
        return b

    def fourcc(self):
        return self.raw(4).decode("latin1")

    def u8(self):
        return struct.unpack("<B", self.raw(1))[0]

    def i32(self):
        return struct.unpack("<i", self.raw(4))[0]

    def i64(self):
        return struct.unpack("<q", self.raw(8))[0]

    def u64(self):
        return struct.unpack("<Q", self.raw(8))[0]

    def array(self, dtype, n):
        dt = np.dtype(dtype)
        return np.frombuffer(self.raw(int(n) * dt.itemsize), dtype=dt).copy()

    def vector(self, dtype):
        return self.array(dtype, self.u64())


def _read_header(r):
    h = {"d": r.i32(), "ntotal": r.i64()}
    r.i64(); r.i64()
    h["is_trained"] = bool(r.u8())
    h["metric"] = r.i32()
    if h["metric"] > 1:
        h["metric_arg"] = struct.unpack("<f", r.raw(4))[0]
    return h


def _read_flat(r, cc):
    h = _read_header(r)
    nwords = r.u64()
    x = r.array(np.float32, nwords)
    h.update(kind="Flat", fourcc=cc, vectors=x.reshape(-1, h["d"]) if h["d"] else x.reshape(0, 0))
    return h


def _read_invlists(r):
    cc = r.fourcc()
    if cc == "il00":
        return None
    if cc != "ilar":
        raise RuntimeError(f"unsupported inverted-list container '{cc}'")
    nlist, code_size = r.u64(), r.u64()
    lt = r.fourcc()
    sizes = np.zeros(nlist, dtype=np.int64)
    if lt == "full":
        s = r.vector(np.uint64)
        assert len(s) == nlist
        sizes[:] = s
    elif lt == "sprs":
        s = r.vector(np.uint64).reshape(-1, 2)
        sizes[s[:, 0].astype(np.int64)] = s[:, 1]
    else:
        raise RuntimeError(f"unsupported list type '{lt}'")
    codes, ids = [], []
    for n in sizes:
        if n:
            codes.append(r.array(np.uint8, int(n) * code_size).reshape(int(n), code_size))
            ids.append(r.array(np.int64, n))
        else:
            codes.append(np.zeros((0, code_size), np.uint8))
            ids.append(np.zeros(0, np.int64))
    return {"nlist": nlist, "code_size": code_size, "codes": codes, "ids": ids}


def _read_ivf_header(r):
    h = _read_header(r)
    h["nlist"], h["nprobe"] = r.u64(), r.u64()
    h["quantizer"] = _read_index(r)
    h["direct_map_type"] = r.u8()
    h["direct_map"] = r.vector(np.int64)
    if h["direct_map_type"] == 2:
        raise RuntimeError("hashtable direct maps are not supported")
    return h


def _read_index(r):
    cc = r.fourcc()
    if cc in ("IxFI", "IxF2", "IxFl"):
        return _read_flat(r, cc)
    if cc == "IwFl":
        h = _read_ivf_header(r)
        h.update(kind="IVFFlat", fourcc=cc, invlists=_read_invlists(r))
        return h
    if cc == "IwPQ":
        h = _read_ivf_header(r)
        h["by_residual"] = bool(r.u8())
        h["code_size"] = r.u64()
        pq_d, M, nbits = r.u64(), r.u64(), r.u64()
        cen = r.vector(np.float32)
        h.update(kind="IVFPQ", fourcc=cc, M=M, nbits=nbits,
                 codebooks=cen.reshape(M, 1 << nbits, pq_d // M), invlists=_read_invlists(r))
        return h
    raise RuntimeError(f"unsupported FAISS index type '{cc}' (supported: IxFI, IxF2, IwFl, IwPQ)")


def parse_faiss(path_or_bytes):
    if isinstance(path_or_bytes, (bytes, bytearray)):
        return _read_index(_R(io.BytesIO(path_or_bytes)))
    with open(path_or_bytes, "rb") as f:
        return _read_index(_R(f))


# ------------------------------------------------------------------------------------------ write
def _w_header(out, d, ntotal, is_trained, metric):
    out.write(struct.pack("<iqqqBi", d, ntotal, 1 << 20, 1 << 20, 1 if is_trained else 0, metric))


def _w_vector(out, arr, dtype):
    arr = np.ascontiguousarray(arr, dtype=dtype)
    out.write(struct.pack("<Q", arr.size))
    out.write(arr.tobytes())


def _w_flat(out, d, metric, vectors):
    vectors = np.ascontiguousarray(vectors, dtype=np.float32).reshape(-1, d) if d else np.zeros((0, 0), np.float32)
    out.write(b"IxFI" if metric == METRIC_INNER_PRODUCT else b"IxF2")
    _w_header(out, d, len(vectors), True, metric)
    out.write(struct.pack("<Q", vectors.size))  # codes bytes / 4 == number of floats
    out.write(vectors.tobytes())


def _w_invlists(out, code_size, codes, ids):
    nlist = len(ids)
    out.write(b"ilar")
    out.write(struct.pack("<QQ", nlist, code_size))
    sizes = np.array([len(i) for i in ids], dtype=np.uint64)
    if int((sizes > 0).sum()) > nlist // 2:
        out.write(b"full")
        _w_vector(out, sizes, np.uint64)
    else:
        out.write(b"sprs")
        nz = np.nonzero(sizes)[0]
        _w_vector(out, np.stack([nz.astype(np.uint64), sizes[nz]], 1).reshape(-1), np.uint64)
    for c, i in zip(codes, ids):
        if len(i):
            out.write(np.ascontiguousarray(c).view(np.uint8).tobytes())
            out.write(np.ascontiguousarray(i, dtype=np.int64).tobytes())


def serialize_faiss(p):
    """Inverse of parse_faiss for the dict layout it returns (only the fields it needs)."""
    out = io.BytesIO()
    kind = p["kind"]
    if kind == "Flat":
        _w_flat(out, p["d"], p["metric"], p["vectors"])
        return out.getvalue()
    out.write(b"IwFl" if kind == "IVFFlat" else b"IwPQ")
    _w_header(out, p["d"], p["ntotal"], p["is_trained"], p["metric"])
    out.write(struct.pack("<QQ", p["nlist"], p["nprobe"]))
    _w_flat(out, p["d"], METRIC_INNER_PRODUCT, p["quantizer"]["vectors"])
    out.write(struct.pack("<B", 0))
    _w_vector(out, np.zeros(0, np.int64), np.int64)
    il = p["invlists"]
    if kind == "IVFPQ":
        out.write(struct.pack("<BQ", 1, p["M"]))
        out.write(struct.pack("<QQQ", p["d"], p["M"], 8))
        _w_vector(out, p["codebooks"], np.float32)
        _w_invlists(out, p["M"], il["codes"], il["ids"])
    else:
        codes = [np.ascontiguousarray(c, dtype=np.float32).view(np.uint8).reshape(len(c), -1) if len(c) else
                 np.zeros((0, p["d"] * 4), np.uint8) for c in il["codes"]]
        _w_invlists(out, p["d"] * 4, codes, il["ids"])
    return out.getvalue()


# ------------------------------------------------------------------------------------------ rsx bridge
def index_to_parsed(index):
    """rsx index -> the dict layout above (vectors / codes pulled out of HBM list by list)."""
    kind = {0: "Flat", 1: "IVFFlat", 2: "IVFPQ"}[index._get("kind")]
    p = {"kind": kind, "d": index.d, "metric": index.metric_type, "ntotal": index.ntotal,
         "is_trained": index.is_trained}
    if kind == "Flat":
        p["vectors"], ids = index.get_list(0)
        if len(ids) and not np.array_equal(ids, np.arange(len(ids))):
            raise RuntimeError("a Flat index with explicit ids has no FAISS IndexFlat representation (use IndexIDMap)")
        return p
    nlist = index.nlist
    p.update(nlist=nlist, nprobe=index.nprobe)
    p["quantizer"] = {"kind": "Flat", "d": index.d, "metric": METRIC_INNER_PRODUCT,
                      "vectors": index.get_centroids() if index.is_trained else np.zeros((0, index.d), np.float32)}
    lists = [index.get_list(l) for l in range(nlist)]
    p["invlists"] = {"nlist": nlist, "codes": [c for c, _ in lists], "ids": [i for _, i in lists]}
    if kind == "IVFPQ":
        p["M"] = index.M
        p["codebooks"] = index.get_codebooks() if index.is_trained else np.zeros((index.M, 256, index.d // index.M), np.float32)
    return p


def write_faiss_index(index, path):
    """rsx index -> .faiss file, STREAMED: the header blocks first, then one inverted list at a time out of HBM (a 100M x 96-byte
    IVF-PQ index is 10.4 GB; nothing here holds more than one list on the host).  Byte-identical to
    serialize_faiss(index_to_parsed(index))."""
    kind = {0: "Flat", 1: "IVFFlat", 2: "IVFPQ"}[index._get("kind")]
    if kind == "Flat":
        data = serialize_faiss(index_to_parsed(index))
        with open(path, "wb") as f:
            f.write(data)
        return
    d, nlist = index.d, index.nlist
    sizes = np.asarray(index.list_sizes(), dtype=np.uint64)
    with open(path, "wb") as out:
        out.write(b"IwFl" if kind == "IVFFlat" else b"IwPQ")
        _w_header(out, d, index.ntotal, index.is_trained, index.metric_type)
        out.write(struct.pack("<QQ", nlist, index.nprobe))
        _w_flat(out, d, METRIC_INNER_PRODUCT, index.get_centroids() if index.is_trained else np.zeros((0, d), np.float32))
        out.write(struct.pack("<B", 0))
        _w_vector(out, np.zeros(0, np.int64), np.int64)
        if kind == "IVFPQ":
            out.write(struct.pack("<BQ", 1, index.M))
            out.write(struct.pack("<QQQ", d, index.M, 8))
            _w_vector(out, index.get_codebooks() if index.is_trained else np.zeros((index.M, 256, d // index.M), np.float32), np.float32)
        code_size = index.M if kind == "IVFPQ" else d * 4
        out.write(b"ilar")
        out.write(struct.pack("<QQ", nlist, code_size))
        if int((sizes > 0).sum()) > nlist // 2:
            out.write(b"full")
            _w_vector(out, sizes, np.uint64)
        else:
            out.write(b"sprs")
            nz = np.nonzero(sizes)[0]
            _w_vector(out, np.stack([nz.astype(np.uint64), sizes[nz]], 1).reshape(-1), np.uint64)
        for l in np.nonzero(sizes)[0]:
            c, i = index.get_list(int(l))
            c = np.ascontiguousarray(c, dtype=np.uint8 if kind == "IVFPQ" else np.float32)
            out.write(c.view(np.uint8).tobytes())
            out.write(np.ascontiguousarray(i, dtype=np.int64).tobytes())


def index_from_parsed(p, device=None, devices=None):
    """devices = [d0, d1, ...]: build ONE handle over several GPUs (rsx_sharded_create) — the file's lists are cut over the
    shards as they are imported (rsx_add_list on a sharded handle), ids kept."""
    import rsx
    kind, d, metric = p["kind"], p["d"], p["metric"]
    if metric not in (METRIC_INNER_PRODUCT, METRIC_L2):
        raise RuntimeError(f"unsupported metric {metric}")
    if kind == "Flat":
        ix = rsx.IndexFlat(d, metric, device=device, devices=devices)
        if len(p["vectors"]):
            ix.add(p["vectors"])
        return ix
    q = p["quantizer"]
    if q["kind"] != "Flat" or q["metric"] != METRIC_INNER_PRODUCT:
        raise RuntimeError("only IndexFlatIP coarse quantisers are supported (the reference builds no other)")
    if kind == "IVFFlat":
        ix = rsx.IndexIVFFlat(None, d, p["nlist"], metric, device=device, devices=devices)
    else:
        if p["nbits"] != 8 or not p["by_residual"]:
            raise RuntimeError("only nbits = 8, by_residual IVFPQ files are supported")
        ix = rsx.IndexIVFPQ(None, d, p["nlist"], p["M"], 8, metric, device=device, devices=devices)
    if len(q["vectors"]):
        ix.set_centroids(q["vectors"])
        if kind == "IVFPQ":
            ix.set_codebooks(p["codebooks"])
    il = p["invlists"]
    if il is not None and p["ntotal"]:
        ix.reserve_lists(np.array([len(i) for i in il["ids"]], dtype=np.int64))
        for l, (c, i) in enumerate(zip(il["codes"], il["ids"])):
            if len(i):
                payload = c if kind == "IVFPQ" else np.ascontiguousarray(c).view(np.float32).reshape(len(i), d)
                ix.add_list(l, payload, i)
    ix.nprobe = max(1, int(p["nprobe"]))       # any value, as FAISS: the engine probes min(nprobe, nlist) lists
    return ix


def read_faiss_index(path, device=None, devices=None):
    return index_from_parsed(parse_faiss(path), device=device, devices=devices)
