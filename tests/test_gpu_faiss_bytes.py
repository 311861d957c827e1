"""GPU: FAISS .faiss files assembled BY HAND in this test — struct.pack calls in the field order of faiss/impl/index_write.cpp
(v1.8.0: write_index_header, WRITEXBVECTOR for IndexFlat codes, write_ivf_header, write_direct_map, write_ProductQuantizer,
write_InvertedLists "ilar" with "full" / "sprs" size tables) — NOT through rsx_faiss_io.serialize_faiss, so the reader is
checked against an independent statement of the layout (no FAISS-produced file exists offline).  Each file is
loaded through rsx.read_index (the reference's faiss.read_index call sites: src/indicies/flat.py:39, ivf_flat.py:71,
ivf_pq.py:75) onto the GPU and searched; ids and fp32 scores must equal the golden results / the CPU oracle bit for bit."""
import struct

import numpy as np
import pytest

from util import assert_same_results, load_golden, regen_gpu

pytestmark = pytest.mark.gpu

DUMMY = 1 << 20


def _header(d, ntotal, is_trained, metric):
    # write_index_header: int d; idx_t ntotal; idx_t dummy; idx_t dummy; bool is_trained; int metric_type
    return struct.pack("<i", d) + struct.pack("<q", ntotal) + struct.pack("<q", DUMMY) + struct.pack("<q", DUMMY) + \
        struct.pack("<?", bool(is_trained)) + struct.pack("<i", metric)


def _flat_bytes(fourcc, x, metric):
    x = np.ascontiguousarray(x, dtype="<f4")
    n, d = x.shape
    b = fourcc + _header(d, n, True, metric)
    b += struct.pack("<Q", (n * d * 4) // 4)          # WRITEXBVECTOR: size of the byte vector / 4
    return b + x.tobytes()


def _ivf_header(d, ntotal, metric, nlist, nprobe, centroids):
    b = _header(d, ntotal, True, metric)
    b += struct.pack("<Q", nlist) + struct.pack("<Q", nprobe)            # size_t nlist, nprobe
    b += _flat_bytes(b"IxFI", centroids, 0)                              # the quantizer: an IndexFlatIP written in place
    b += struct.pack("<b", 0)                                            # direct map: char type = NoMap
    b += struct.pack("<Q", 0)                                            # ... and its (empty) array
    return b


def _invlists(nlist, code_size, lists, sparse):
    b = b"ilar" + struct.pack("<Q", nlist) + struct.pack("<Q", code_size)
    sizes = [len(ids) for _, ids in lists]
    if sparse:
        nz = [(l, s) for l, s in enumerate(sizes) if s]
        b += b"sprs" + struct.pack("<Q", 2 * len(nz))
        for l, s in nz:
            b += struct.pack("<QQ", l, s)
    else:
        b += b"full" + struct.pack("<Q", nlist)
        for s in sizes:
            b += struct.pack("<Q", s)
    for codes, ids in lists:
        if len(ids):
            b += np.ascontiguousarray(codes).tobytes() + np.ascontiguousarray(ids, dtype="<i8").tobytes()
    return b


def _lists(assign, payload, nlist, keep=None):
    out = []
    for l in range(nlist):
        sel = np.nonzero(assign == l)[0] if (keep is None or l in keep) else np.zeros(0, np.int64)
        out.append((payload[sel], sel.astype(np.int64)))
    return out


def test_hand_packed_ixfi(gpu, tmp_path):
    g = load_golden("flat_ip_d100")
    x, q = regen_gpu(gpu, g)
    path = tmp_path / "index_Flat.faiss"
    path.write_bytes(_flat_bytes(b"IxFI", x.astype(np.float32), 0))
    ix = gpu.read_index(str(path))
    assert ix.ntotal == g["n"] and ix.d == g["d"]
    D, I = ix.search(q, g["k"])
    assert_same_results(D, I, g["D"], g["I"], "hand-packed IxFI")


def test_hand_packed_iwfl_full(gpu, orc, tmp_path):
    g = load_golden("ivfflat_d768")
    x, q = regen_gpu(gpu, g)
    x32 = x.astype(np.float32)
    a, _ = orc.assign_ip(g["centroids"], x32)
    lists = _lists(a, x32, g["nlist"])
    b = b"IwFl" + _ivf_header(g["d"], g["n"], 0, g["nlist"], 1, g["centroids"]) + _invlists(g["nlist"], g["d"] * 4, lists, sparse=False)
    path = tmp_path / "index_IVFFlat.faiss"
    path.write_bytes(b)
    ix = gpu.read_index(str(path))
    assert ix.ntotal == g["n"] and ix.nlist == g["nlist"] and ix.nprobe == 1
    ix.nprobe = g["nprobe"]                          # the reference sets probe after read_index (ivf_flat.py:73)
    D, I = ix.search(q, g["k"])
    assert_same_results(D, I, g["D"], g["I"], "hand-packed IwFl / full")


@pytest.mark.parametrize("sparse", [False, True])
def test_hand_packed_iwpq(gpu, orc, tmp_path, sparse):
    g = load_golden("ivfpq_d64_m16")
    x, q = regen_gpu(gpu, g)
    x32 = x.astype(np.float32)
    cen, cb, M, nlist = g["centroids"], g["codebooks"], g["M"], g["nlist"]
    a, _ = orc.assign_ip(cen, x32)
    codes = orc.pq_encode(cb, orc.residuals(cen, x32, a))
    keep = {3, 7, 20} if sparse else None            # "sprs" is written when at most half of the lists are non-empty
    lists = _lists(a, codes, nlist, keep)
    ntotal = sum(len(i) for _, i in lists)
    b = b"IwPQ" + _ivf_header(g["d"], ntotal, 0, nlist, g["nprobe"], cen)
    b += struct.pack("<?", True)                      # by_residual
    b += struct.pack("<Q", M)                         # code_size
    b += struct.pack("<QQQ", g["d"], M, 8)            # ProductQuantizer: d, M, nbits
    b += struct.pack("<Q", cb.size) + np.ascontiguousarray(cb, dtype="<f4").tobytes()
    b += _invlists(nlist, M, lists, sparse)
    path = tmp_path / "index_IVFPQ.faiss"
    path.write_bytes(b)
    ix = gpu.read_index(str(path))
    assert ix.ntotal == ntotal and ix.nprobe == g["nprobe"] and ix.M == M
    D, I = ix.search(q, g["k"])
    if not sparse:
        assert_same_results(D, I, g["D"], g["I"], "hand-packed IwPQ / full")
    else:
        sel = np.concatenate([i for _, i in lists])
        lm = orc.ListMajor(a[sel], sel, codes[sel], nlist)
        Dr, Ir = orc.ivfpq_search(cen, cb, lm, q.astype(np.float32), g["nprobe"], g["k"])
        assert_same_results(D, I, Dr, Ir, "hand-packed IwPQ / sprs")
    # the same bytes spread over two shards while loading (a FAISS hand-over served by several GPUs) answer identically
    n = gpu.get_num_gpus()
    sh = gpu.read_index(str(path), devices=[0, 1 % n])
    assert sh.nshards == 2 and sh.ntotal == ntotal
    D2, I2 = sh.search(q, g["k"])
    assert_same_results(D2, I2, D, I, "hand-packed IwPQ re-sharded on load")
    if not sparse:
        # ... and keep growing like the single index: rows added WITHOUT ids continue the id sequence at ntotal on both (the
        # re-sharded handle used to restart at 0 and collide with the imported ids — ADVICE r3)
        extra = x[:50] * np.float16(1.0)
        ix.add(extra); sh.add(extra)
        assert ix.ntotal == sh.ntotal == ntotal + 50
        qe = extra[:8]
        ix.nprobe = sh.nprobe = nlist
        De, Ie = ix.search(qe, 2048)
        Ds, Is = sh.search(qe, 2048)
        assert_same_results(Ds, Is, De, Ie, "add after a re-sharded load")
        # half of the index comes back per query: the new rows are in there under NEW ids (ntotal + j), never under recycled ones
        assert (Is >= ntotal).any(1).all() and Is.max() < ntotal + 50
