"""FAISS .faiss container: pure-numpy parse/serialise (CPU) and the rsx bridge (GPU).

No FAISS-produced file exists offline, so these pin the layout field by field against the published
index_write.cpp structure restated in rsx_faiss_io.py, plus round trips."""
import os
import struct

import numpy as np
import pytest

import rsx_faiss_io as fio


def _flat(d=8, n=5, metric=0, seed=0):
    rng = np.random.RandomState(seed)
    return {"kind": "Flat", "d": d, "metric": metric, "vectors": rng.randn(n, d).astype(np.float32)}


def test_flat_layout_and_roundtrip():
    p = _flat()
    b = fio.serialize_faiss(p)
    assert b[:4] == b"IxFI"
    d, ntotal, dm1, dm2, trained, metric = struct.unpack("<iqqqBi", b[4:4 + 33])
    assert (d, ntotal, dm1, dm2, trained, metric) == (8, 5, 1 << 20, 1 << 20, 1, 0)
    (nwords,) = struct.unpack("<Q", b[37:45])
    assert nwords == 5 * 8 and len(b) == 45 + 5 * 8 * 4          # codes counted in 4-byte words
    q = fio.parse_faiss(b)
    assert q["kind"] == "Flat" and q["metric"] == 0 and np.array_equal(q["vectors"], p["vectors"])
    assert fio.serialize_faiss(_flat(metric=1))[:4] == b"IxF2"


def _ivf(kind, nlist=6, d=8, M=4, seed=1, sparse=False):
    rng = np.random.RandomState(seed)
    sizes = [0, 3, 0, 0, 0, 2] if sparse else [2, 3, 0, 1, 4, 2]
    p = {"kind": kind, "d": d, "metric": 0, "ntotal": sum(sizes), "is_trained": True, "nlist": nlist, "nprobe": 3,
         "quantizer": _flat(d, nlist, 0, seed + 1)}
    nid = 0
    codes, ids = [], []
    for n in sizes:
        ids.append(np.arange(nid, nid + n, dtype=np.int64) * 3 + 7); nid += n
        codes.append(rng.randint(0, 256, size=(n, M)).astype(np.uint8) if kind == "IVFPQ" else rng.randn(n, d).astype(np.float32))
    p["invlists"] = {"nlist": nlist, "codes": codes, "ids": ids}
    if kind == "IVFPQ":
        p["M"] = M
        p["codebooks"] = rng.randn(M, 256, d // M).astype(np.float32)
    return p


@pytest.mark.parametrize("sparse", [False, True])
@pytest.mark.parametrize("kind", ["IVFFlat", "IVFPQ"])
def test_ivf_roundtrip(kind, sparse):
    p = _ivf(kind, sparse=sparse)
    b = fio.serialize_faiss(p)
    assert b[:4] == (b"IwFl" if kind == "IVFFlat" else b"IwPQ")
    assert (b"sprs" if sparse else b"full") in b and b"ilar" in b and b"IxFI" in b
    q = fio.parse_faiss(b)
    assert q["kind"] == kind and q["nlist"] == 6 and q["nprobe"] == 3 and q["ntotal"] == p["ntotal"]
    assert np.array_equal(q["quantizer"]["vectors"], p["quantizer"]["vectors"])
    for l in range(6):
        assert np.array_equal(q["invlists"]["ids"][l], p["invlists"]["ids"][l])
        got = q["invlists"]["codes"][l]
        want = p["invlists"]["codes"][l]
        if kind == "IVFFlat":
            got = got.view(np.float32).reshape(len(want), -1) if len(want) else want
        assert np.array_equal(got, want)
    if kind == "IVFPQ":
        assert q["M"] == 4 and q["nbits"] == 8 and q["by_residual"] and q["code_size"] == 4
        assert np.array_equal(q["codebooks"], p["codebooks"])
    assert fio.serialize_faiss({**q, "invlists": {**q["invlists"], "codes": p["invlists"]["codes"]}}) == b


def test_rejects_unknown_and_truncated():
    with pytest.raises(RuntimeError, match="unsupported FAISS index type"):
        fio.parse_faiss(b"IxPQ" + b"\0" * 64)
    with pytest.raises(RuntimeError, match="truncated"):
        fio.parse_faiss(fio.serialize_faiss(_flat())[:-3])


@pytest.mark.gpu
def test_engine_roundtrip_through_faiss_files(gpu, orc, tmp_path):
    from util import load_golden, regen_gpu, assert_same_results
    g = load_golden("ivfpq_d64_m16")
    x, q = regen_gpu(gpu, g)
    ix = gpu.IndexIVFPQ(None, g["d"], g["nlist"], g["M"], 8, 0)
    ix.set_centroids(g["centroids"]); ix.set_codebooks(g["codebooks"]); ix.add(x)
    ix.nprobe = g["nprobe"]
    path = str(tmp_path / "index_IVFPQ.faiss")
    fio.write_faiss_index(ix, path)
    assert open(path, "rb").read(4) == b"IwPQ"
    ix2 = gpu.read_index(path)                     # magic != RSX1 -> the FAISS reader
    assert ix2.ntotal == g["n"] and ix2.nprobe == g["nprobe"]
    D, I = ix2.search(q, g["k"])
    assert_same_results(D, I, g["D"], g["I"], "IwPQ file")
    gf = load_golden("flat_ip_d100")
    xf, qf = regen_gpu(gpu, gf)
    f = gpu.IndexFlatIP(gf["d"]); f.add(xf)
    p2 = str(tmp_path / "index_Flat.faiss")
    fio.write_faiss_index(f, p2)
    D, I = gpu.read_index(p2).search(qf, gf["k"])
    assert_same_results(D, I, gf["D"], gf["I"], "IxFI file")
    gi = load_golden("ivfflat_d768")
    xi, qi = regen_gpu(gpu, gi)
    iv = gpu.IndexIVFFlat(None, gi["d"], gi["nlist"], 0)
    iv.set_centroids(gi["centroids"]); iv.add(xi); iv.nprobe = gi["nprobe"]
    p3 = str(tmp_path / "index_IVFFlat.faiss")
    fio.write_faiss_index(iv, p3)
    D, I = gpu.read_index(p3).search(qi, gi["k"])
    assert_same_results(D, I, gi["D"], gi["I"], "IwFl file")


@pytest.mark.gpu
def test_streamed_writer_matches_serializer_and_faiss_leg(gpu, orc, tmp_path, monkeypatch):
    """write_faiss_index streams list by list; its bytes must equal serialize_faiss(index_to_parsed(index)).  The bench's
    opportunistic FAISS leg (tools/faiss_leg.py) is then driven end to end with a stand-in `faiss` module whose read_index
    parses the file and searches with the CPU oracle: the hand-over file, the comparison and the verdict keys all run."""
    import sys
    import types
    from util import load_golden, regen_gpu
    g = load_golden("ivfpq_d64_m16")
    x, q = regen_gpu(gpu, g)
    ix = gpu.IndexIVFPQ(None, g["d"], g["nlist"], g["M"], 8, 0)
    ix.set_centroids(g["centroids"]); ix.set_codebooks(g["codebooks"]); ix.add(x)
    ix.nprobe = g["nprobe"]
    path = str(tmp_path / "streamed.faiss")
    fio.write_faiss_index(ix, path)
    assert open(path, "rb").read() == fio.serialize_faiss(fio.index_to_parsed(ix))

    class _Fx:
        def __init__(self, p):
            self.p, self.nprobe, self.ntotal = p, 1, p["ntotal"]

        def search(self, xq, k):
            il = self.p["invlists"]
            a = np.concatenate([np.full(len(i), l, np.int64) for l, i in enumerate(il["ids"])])
            lm = orc.ListMajor(a, np.concatenate(il["ids"]), np.concatenate(il["codes"]), self.p["nlist"])
            return orc.ivfpq_search(self.p["quantizer"]["vectors"], self.p["codebooks"], lm, np.asarray(xq, np.float32), self.nprobe, k)

    fake = types.ModuleType("faiss")
    fake.__version__ = "stand-in"
    fake.read_index = lambda p: _Fx(fio.parse_faiss(p))
    fake.omp_get_max_threads = lambda: 1
    monkeypatch.setitem(sys.modules, "faiss", fake)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from faiss_leg import faiss_leg, compare
    D, I = ix.search(q, g["k"])
    r = faiss_leg(ix, q, g["k"], g["nprobe"], D, I, log=lambda *a: None, repeats=1)
    assert r["available"] and r["parity"] == "green" and r["ids_identical"] and r["scores_bit_identical"] and r["faiss_ntotal"] == g["n"]
    # compare(): a swap inside a run of equal scores is a tie-order difference, a different id is not
    Df = np.array([[3.0, 2.0, 2.0, 1.0]], np.float32); If = np.array([[5, 7, 9, 1]])
    c = compare(Df, If, Df.copy(), np.array([[5, 9, 7, 1]]))
    assert c["parity"] == "green" and c["ids_differ_only_in_tie_order_queries"] == 1 and not c["ids_identical"]
    assert compare(Df, If, Df.copy(), np.array([[5, 7, 8, 1]]))["parity"] == "differs"
