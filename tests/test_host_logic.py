"""CPU: the host-side mirror of the reference interface (paths, file contract, id maps, ctxs, merge),
driven through a test-double engine (tests/fake_engine.py) so no GPU is needed."""
import json
import os
import pickle
import sys

import numpy as np
import pytest

from conftest import REPO
from util import GOLDEN


class NS(dict):
    """attribute + .get access, like an OmegaConf node"""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def make_cfg(tmp, index_type, shard_ids, **index_kw):
    index = NS(index_type=index_type, index_shard_ids=shard_ids, projection_size=32, sample_train_size=600,
               ncentroids=4, probe=4, n_subquantizers=4, n_bits=8)
    index.update(index_kw)
    return NS(
        datastore=NS(domain="unit", embedding=NS(embedding_dir=os.path.join(tmp, "emb"), prefix="passages",
                                                 passages_dir=os.path.join(tmp, "psg")), index=index),
        evaluation=NS(eval_output_dir=os.path.join(tmp, "out"), data=NS(eval_data=os.path.join(tmp, "q.jsonl")),
                      search=NS(n_docs=3, overwrite=False)),
        model=NS(),
    )


def write_datastore(tmp, orc, n_shards=2, per=400, d=32):
    os.makedirs(os.path.join(tmp, "emb"), exist_ok=True)
    os.makedirs(os.path.join(tmp, "psg"), exist_ok=True)
    embs = []
    for s in range(n_shards):
        e = orc.synth_vectors(d, 6, 11, 100 + s, 0.5, 0, per)
        embs.append(e)
        with open(os.path.join(tmp, "emb", f"passages_{s:02d}.pkl"), "wb") as f:
            pickle.dump((list(range(per)), e), f)          # (ids, fp16 embeddings) — src/embed.py:155-156
        with open(os.path.join(tmp, "psg", f"raw_passages-{s}-of-{n_shards}.pkl"), "wb") as f:
            pickle.dump([{"text": f"shard {s} chunk {c} é", "id": c} for c in range(per)], f)
    return embs


@pytest.fixture()
def fake(monkeypatch, orc):
    import fake_engine
    import rsx
    for name in ("IndexFlatIP", "IndexIVFFlat", "IndexIVFPQ", "read_index", "write_index"):
        monkeypatch.setattr(rsx, name, getattr(fake_engine, name))
    return fake_engine


def test_paths_match_reference_golden():
    from src.indicies.index_utils import get_index_dir_and_embedding_paths
    with open(os.path.join(GOLDEN, "paths_golden.json")) as f:
        gold = json.load(f)
    for c in gold["cases"]:
        cfg = NS(datastore=NS(embedding=NS(embedding_dir=c["embedding_dir"], prefix=c["prefix"]),
                              index=NS(index_type=c["index_type"], index_shard_ids=c["index_shard_ids"])))
        index_dir, paths = get_index_dir_and_embedding_paths(cfg)
        assert index_dir == c["index_dir"] and paths == c["embedding_paths"]


def test_index_file_names():
    # reference src/indicies/base.py:23-30
    from src.indicies.base import index_file_names
    a = NS(index_type="IVFPQ", sample_train_size=1000000, projection_size=768, ncentroids=4096)
    assert index_file_names(a) == ("index_IVFPQ.1000000.768.4096.faiss", True)
    a.index_type = "IVFFlat"
    assert index_file_names(a) == ("index_IVFFlat.1000000.768.4096.faiss", True)
    assert index_file_names(NS(index_type="Flat")) == ("index_Flat.faiss", False)


@pytest.mark.parametrize("index_type", ["Flat", "IVFFlat", "IVFPQ"])
def test_indexer_facade_contract(tmp_path, orc, fake, index_type):
    from src.indicies.base import Indexer
    tmp = str(tmp_path)
    embs = write_datastore(tmp, orc)
    cfg = make_cfg(tmp, index_type, [1, 0])
    ix = Indexer(cfg)
    index_dir = os.path.join(tmp, "emb", f"index_{index_type}", "0_1")
    name = "index_Flat.faiss" if index_type == "Flat" else f"index_{index_type}.600.32.4.faiss"
    assert os.path.exists(os.path.join(index_dir, name)) and os.path.exists(os.path.join(index_dir, name + ".meta"))
    assert os.path.exists(os.path.join(index_dir, "passage_pos_id_map.pkl"))
    if index_type != "Flat":
        assert os.path.exists(os.path.join(index_dir, name + ".trained"))
    ds = ix.datastore
    assert ds.index.ntotal == 800 and len(ds.index_id_to_db_id) == 800
    assert ds.index_id_to_db_id[0] == [0, 0] and ds.index_id_to_db_id[400] == [1, 0]   # shard order, sequential ids
    q = np.concatenate([embs[0][5:6], embs[1][7:8]], 0)
    scores, passages, db_ids = ix.search(q, k=3)
    assert isinstance(scores, list) and len(scores) == 2 and len(scores[0]) == 3 and isinstance(scores[0][0], float)
    assert isinstance(passages[0][0], str) and len(db_ids[0]) == 3
    if index_type == "Flat":   # inner product: compare with the brute-force oracle over both shards
        D, I = orc.flat_search(q.astype(np.float32), np.concatenate(embs, 0).astype(np.float32), 3, 0)
        assert db_ids == [[[int(i) // 400, int(i) % 400] for i in row] for row in I]
        assert scores == D.tolist()
        assert passages[0][0] == f"shard {I[0, 0] // 400} chunk {I[0, 0] % 400} é"
    # second construction loads instead of building (file-existence contract, flat.py:37 / ivf_flat.py:69)
    mtime = os.path.getmtime(os.path.join(index_dir, name))
    ix2 = Indexer(cfg)
    assert os.path.getmtime(os.path.join(index_dir, name)) == mtime
    s2, p2, d2 = ix2.search(q, k=3)
    assert d2 == db_ids and s2 == scores
    if index_type != "Flat":
        assert ix2.datastore.index.nprobe == 4     # probe applied at load (ivf_flat.py:73)


def test_passage_fetch_matches_reference_reads(tmp_path, orc, fake, monkeypatch):
    """get_retrieved_passages resolves ids through integer arrays and reads each passage with ONE os.pread on a cached
    descriptor instead of the reference's open() per id (flat.py:115-121): the strings must be what
    open/seek/readline/json.loads gives, non-ASCII text included, every passage file is opened once, and a repeated id is
    read once."""
    import json
    from src.indicies.base import Indexer
    tmp = str(tmp_path)
    write_datastore(tmp, orc)
    ds = Indexer(make_cfg(tmp, "Flat", [0, 1])).datastore
    ids = [[0, 399, 400, 799, 5], [5, 5, 400, 0, 799]]
    opened, preads = [], []
    real_open, real_pread = os.open, os.pread
    monkeypatch.setattr(os, "open", lambda f, *a, **k: (opened.append(f), real_open(f, *a, **k))[1])
    monkeypatch.setattr(os, "pread", lambda fd, n, off: (preads.append((fd, off)), real_pread(fd, n, off))[1])
    passages, db_ids = ds.get_retrieved_passages(ids)
    monkeypatch.undo()
    assert len([f for f in opened if str(f).endswith(".jsonl")]) == 2          # one descriptor per passage file
    assert len(preads) == 5                                                     # 5 distinct passages, each read once
    for row_ids, row_txt, row_db in zip(ids, passages, db_ids):
        for i, txt, db in zip(row_ids, row_txt, row_db):
            shard, chunk = ds.index_id_to_db_id[i]
            fname, pos = ds.psg_pos_id_map[shard][chunk]
            with open(fname, "r") as f:                                           # the reference's read
                f.seek(pos)
                assert txt == json.loads(f.readline())["text"]
            assert db == [shard, chunk]
    # the reference's -1 quirk (an unfilled slot indexes the LAST passage) is kept, explicitly
    p_neg, d_neg = ds.get_retrieved_passages(np.array([[-1, 0]]))
    assert p_neg[0][0] == ds._get_passage(len(ds.index_id_to_db_id) - 1)["text"] and d_neg[0][0] == ds.index_id_to_db_id[-1]
    # ragged input takes the per-id path and gives the same strings
    assert ds.get_retrieved_passages([[0, 399], [5]])[0] == [passages[0][:2], [passages[0][4]]]
    ds.close_passage_files()
    ds._MAX_OPEN_PASSAGE_FILES = 1                                                # LRU bound holds
    assert ds.get_retrieved_passages(ids) == (passages, db_ids)
    assert len(ds._psg_fds) == 1
    ds.close_passage_files()
    assert "_psg_fds" not in ds.__dict__ and "_psg_files" not in ds.__dict__


def test_unknown_index_type_raises(tmp_path, orc, fake):
    from src.indicies.base import Indexer
    write_datastore(str(tmp_path), orc)
    with pytest.raises(NotImplementedError):
        Indexer(make_cfg(str(tmp_path), "PQ", [0]))     # stale configs in the reference hit this (base.py:72)


def test_search_driver_and_merge(tmp_path, orc, fake):
    import src.search as S
    tmp = str(tmp_path)
    embs = write_datastore(tmp, orc)
    data = [{"raw_query": ""}, {"raw_query": "a"}, {"raw_query": "b"}]
    q = np.concatenate([embs[0][1:2], embs[1][2:3]], 0)
    outs = []
    for shard in ([0], [1]):
        cfg = make_cfg(tmp, "Flat", shard)
        S.search_dense_topk(cfg, data=data, questions_embedding=q)
        path = S.get_search_output_path(cfg, shard)
        assert path == os.path.join(tmp, "out", str(shard[0]), "q_retrieved_results.jsonl")
        rows = [json.loads(l) for l in open(path)]
        assert rows[0]["ctxs"] == [None]
        assert set(rows[1]["ctxs"][0]) == {"id", "source", "retrieval text", "retrieval score"}
        assert isinstance(rows[1]["ctxs"][0]["retrieval score"], str) and rows[1]["ctxs"][0]["source"] == "unit"
        outs.append(rows)
    # multi-index merge == brute force over both shards (scores as the reference serialises them)
    cfg = make_cfg(tmp, "Flat", [[0], [1]])
    merged_path = S.post_hoc_merge_topk(cfg)
    assert merged_path == os.path.join(tmp, "out", "0-1", "q_retrieved_results.jsonl")
    merged = [json.loads(l) for l in open(merged_path)]
    allx = np.concatenate(embs, 0).astype(np.float32)
    D, I = orc.flat_search(q.astype(np.float32), allx, 3, 0)
    for qi, row in enumerate(merged[1:]):
        got = [(c["id"], float(c["retrieval score"])) for c in row["ctxs"]]
        want = [([int(i) // 400, int(i) % 400], float(str(float(s)))) for i, s in zip(I[qi], D[qi])]
        assert got == want
    # skip-if-exists contract
    before = os.path.getmtime(merged_path)
    S.post_hoc_merge_topk(cfg)
    assert os.path.getmtime(merged_path) == before


def test_flat_indexer_matches_reference_class_golden(tmp_path, orc, fake):
    """tests/golden/flat_indexer_golden.json records what the reference's own FlatIndexer class does on this datastore
    (run with the same test-double engine standing in for faiss — tests/golden/make_golden.py).  The mirror class must
    write the same files, build the same id map and position map, and return the same search() triple."""
    import json
    from src.indicies.flat import FlatIndexer
    with open(os.path.join(GOLDEN, "flat_indexer_golden.json")) as f:
        g = json.load(f)
    tmp = str(tmp_path)
    ds = g["datastore"]
    embs = write_datastore(tmp, orc, n_shards=ds["n_shards"], per=ds["per"], d=ds["d"])
    os.makedirs(os.path.join(tmp, "index"))
    kw = dict(embed_paths=[os.path.join(tmp, "emb", f"passages_{s:02d}.pkl") for s in ds["embed_order"]],
              index_path=os.path.join(tmp, "index", "index_Flat.faiss"),
              meta_file=os.path.join(tmp, "index", "index_Flat.faiss.meta"),
              passage_dir=os.path.join(tmp, "psg"),
              pos_map_save_path=os.path.join(tmp, "index", "passage_pos_id_map.pkl"), dimension=ds["d"])
    ix = FlatIndexer(**kw)
    assert sorted(os.listdir(os.path.join(tmp, "index"))) == g["index_dir_files"]
    assert sorted(os.listdir(os.path.join(tmp, "psg"))) == g["passage_dir_files"]
    with open(kw["meta_file"], "rb") as f:
        meta = pickle.load(f)
    assert len(meta) == g["meta_len"] and meta[:3] == g["meta_head"] and meta[400] == g["meta_at_400"]
    from util import sha
    assert sha(np.asarray(meta, dtype=np.int64)) == g["meta_sha"]
    with open(kw["pos_map_save_path"], "rb") as f:
        pos = pickle.load(f)
    for sh, m in g["pos_map_sample"].items():
        for c, (rel, off) in m.items():
            fn, got_off = pos[int(sh)][int(c)]
            assert os.path.relpath(fn, tmp) == rel and got_off == off
    q = np.concatenate([embs[s][r:r + 1] for s, r in ds["query_rows"]], 0)
    scores, passages, db_ids = ix.search(q, k=g["k"])
    assert passages == g["passages"] and db_ids == g["db_ids"]
    assert np.allclose(scores, g["scores"], rtol=0, atol=1e-5)
    assert int(ix.index.ntotal) == g["attrs"]["ntotal"]
    mtime = os.path.getmtime(kw["index_path"])
    ix2 = FlatIndexer(**kw)                                # loads, does not rebuild
    assert os.path.getmtime(kw["index_path"]) == mtime
    assert ix2.search(q, k=g["k"])[1:] == (passages, db_ids)


@pytest.mark.parametrize("case", [0, 1, 2])
def test_indexer_matches_reference_facade_golden(tmp_path, orc, fake, case):
    """tests/golden/indexer_facade_golden.json = the reference's own Indexer facade + FlatIndexer / IVFFlatIndexer /
    IVFPQIndexer run on this datastore with the same test-double engine as `faiss` and numpy's RNG seeded (so the
    unseeded training sample of ivf_flat.py:132 is reproducible).  The mirror, seeded the same way, must derive the same
    index directory and file names, id map, engine state and search() triple."""
    import json
    from src.indicies.base import Indexer
    with open(os.path.join(GOLDEN, "indexer_facade_golden.json")) as f:
        g = json.load(f)
    c = g["cases"][case]
    tmp = str(tmp_path)
    ds = g["datastore"]
    embs = write_datastore(tmp, orc, n_shards=ds["n_shards"], per=ds["per"], d=ds["d"])
    cfg = make_cfg(tmp, c["index_type"], c["index_shard_ids"], **g["index_args"])
    np.random.seed(g["np_random_seed"])
    ix = Indexer(cfg)
    index_dir = os.path.join(tmp, c["index_dir"])
    assert sorted(os.listdir(index_dir)) == c["index_dir_files"]
    meta_name = [f for f in c["index_dir_files"] if f.endswith(".meta")][0]
    with open(os.path.join(index_dir, meta_name), "rb") as f:
        meta = pickle.load(f)
    from util import sha
    assert len(meta) == c["meta_len"] and meta[:2] == c["meta_head"] and meta[400] == c["meta_at_400"]
    assert sha(np.asarray(meta, dtype=np.int64)) == c["meta_sha"]
    d_ = ix.datastore
    assert int(d_.index.ntotal) == c["ntotal"]
    if c["index_type"] != "Flat":
        assert int(d_.index.nprobe) == c["nprobe"] and d_.probe == c["probe_attr"]
    q = np.concatenate([embs[s][r:r + 1] for s, r in ds["query_rows"]], 0)
    scores, passages, db_ids = ix.search(q, k=g["k"])
    assert db_ids == c["db_ids"] and passages == c["passages"]
    assert np.allclose(scores, c["scores"], rtol=0, atol=1e-5)


def test_search_driver_matches_reference_golden(tmp_path, fake):
    """tests/golden/search_golden.json was produced by RUNNING the reference's own add_passages_to_eval_data,
    get_search_output_path, get_merged_search_output_path and post_hoc_merge_topk (tests/golden/make_golden.py):
    the host mirror must give the same records, paths and merged files (ties: earlier shard first)."""
    import copy, json
    from src import search as S
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "search_golden.json")) as f:
        g = json.load(f)
    a = g["add_passages"]
    data = copy.deepcopy(a["in"]["data"])
    S.add_passages_to_eval_data(data, a["in"]["passages"], a["in"]["scores"], a["in"]["db_ids"], a["in"]["valid_query_idx"],
                                domain=a["in"]["domain"])
    assert data == a["out"]
    for c in g["paths"]:
        cfg = NS(datastore=NS(index=NS(index_shard_ids=c["index_shard_ids"])),
                 evaluation=NS(eval_output_dir="/out/eval", data=NS(eval_data="/data/eval/nq_open.jsonl")))
        first = c["index_shard_ids"][0] if isinstance(c["index_shard_ids"][0], list) else c["index_shard_ids"]
        assert S.get_search_output_path(cfg, first) == c["per_index"]
        assert S.get_merged_search_output_path(cfg) == c["merged"]
    m = g["merge"]
    cfg = NS(datastore=NS(index=NS(index_shard_ids=m["index_shard_ids"])),
             evaluation=NS(eval_output_dir=os.path.join(str(tmp_path), "eval"), data=NS(eval_data="/data/eval/unit.jsonl"),
                           search=NS(overwrite=True, n_docs=m["n_docs"])))
    for sid, exs in m["shard_results"].items():
        pth = S.get_search_output_path(cfg, [int(sid)])
        os.makedirs(os.path.dirname(pth), exist_ok=True)
        with open(pth, "w") as f:
            for ex in exs:
                f.write(json.dumps(ex) + "\n")
    out = S.post_hoc_merge_topk(cfg)
    assert os.path.relpath(out, str(tmp_path)) == m["merged_relpath"]
    with open(out) as f:
        assert [json.loads(line) for line in f] == m["merged"]


def test_merge_matches_reference_rerank_elements_golden(orc):
    """search_golden.json["rerank_elements"] = output of the reference's own api/serve_main_node.py::rerank_elements on
    3 shards x 4 queries x 5 results full of equal scores.  The oracle merge (what the GPU kernel is tested against) and
    the host merge must give exactly that order."""
    import json
    sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd"))
    from sharded import merge_topk_host
    with open(os.path.join(GOLDEN, "search_golden.json")) as f:
        g = json.load(f)["rerank_elements"]
    D, I = np.asarray(g["D"], np.float32), np.asarray(g["I"], np.int64)
    for fn in (lambda: orc.merge_topk(D, I, 0), lambda: merge_topk_host(D, I, 0)):
        Dm, Im = fn()
        assert Im.tolist() == g["IDs"] and Dm.tolist() == g["scores"]


def test_merge_ctxs_is_stable():
    from src.search import merge_ctxs
    mk = lambda tag, s: {"id": tag, "retrieval score": str(s)}
    out = merge_ctxs([[mk("a0", 5.0), mk("a1", 3.0)], [mk("b0", 5.0), mk("b1", 4.0)], [mk("c0", 5.0), mk("c1", 4.0)]], 4)
    assert [c["id"] for c in out] == ["a0", "b0", "c0", "b1"]   # ties keep earlier shard first


def test_safe_write_jsonl_removes_partial(tmp_path):
    from src.search import safe_write_jsonl
    p = str(tmp_path / "x.jsonl")
    safe_write_jsonl([{"a": 1}, {"b": {1, 2}}], p)    # a set is not JSON-serialisable
    assert not os.path.exists(p)
    safe_write_jsonl([{"a": 1}], p)
    assert open(p).read() == '{"a": 1}\n'


def test_shard_range():
    from sharded import shard_range
    spans = [shard_range(103, r, 8) for r in range(8)]
    assert spans[0][0] == 0 and spans[-1][1] == 103
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_device_list_resolution(monkeypatch):
    """cfg.datastore.index.devices / RSX_DEVICES -> rsx._resolve_devices: explicit arguments win, then the facade's default,
    then the environment; strings are comma lists; one entry names the device of an ordinary handle."""
    import rsx
    monkeypatch.delenv("RSX_DEVICES", raising=False)
    rsx.set_default_devices(None)
    try:
        assert rsx._resolve_devices(None, None) is None
        assert rsx._resolve_devices(3, None) is None                       # an explicit device: never sharded
        assert rsx._resolve_devices(None, [0, 1, 2]) == [0, 1, 2]
        assert rsx._resolve_devices(None, "1, 3") == [1, 3]
        assert rsx._resolve_devices(None, [2]) == [2]
        assert rsx._resolve_devices(None, []) is None
        monkeypatch.setenv("RSX_DEVICES", "0,1")
        assert rsx._resolve_devices(None, None) == [0, 1]
        assert rsx._resolve_devices(5, None) is None                       # ... the environment does not override it either
        rsx.set_default_devices([4, 5, 6])
        assert rsx._resolve_devices(None, None) == [4, 5, 6]               # facade default before the environment
        assert rsx._resolve_devices(None, [7, 8]) == [7, 8]
    finally:
        rsx.set_default_devices(None)


def test_backend_and_storage_dtype_keys(tmp_path, orc, monkeypatch):
    """SURVEY 8b optional config keys: `backend` selects the engine module the backends call — "faiss" resolves to the
    `faiss` package (here the test double registered under that name; nothing is patched into rsx) and fails loudly when
    the package is missing; `storage_dtype` is validated against what the index reports."""
    import sys
    import fake_engine
    from src.indicies import engine as sel
    from src.indicies.base import Indexer
    tmp = str(tmp_path)
    embs = write_datastore(tmp, orc)
    q = np.concatenate([embs[0][5:6], embs[1][7:8]], 0)
    try:
        monkeypatch.setitem(sys.modules, "faiss", fake_engine)
        cfg = make_cfg(tmp, "Flat", [0, 1])
        cfg.datastore.index["backend"] = "faiss"
        ix = Indexer(cfg)
        assert ix.backend == "faiss" and isinstance(ix.datastore.index, fake_engine.IndexFlatIP)
        # the selection is scoped to that Indexer's construction: a later Indexer without the key gets the default engine
        assert sel.backend_name() == "mi355x"
        import rsx as _rsx
        assert _rsx.get_default_devices() is None
        D, I = orc.flat_search(q.astype(np.float32), np.concatenate(embs, 0).astype(np.float32), 3, 0)
        scores, passages, db_ids = ix.search(q, k=3)
        assert scores == D.tolist() and db_ids == [[[int(i) // 400, int(i) % 400] for i in row] for row in I]
        # no faiss package -> a clear error, never a silent fallback to another engine
        monkeypatch.delitem(sys.modules, "faiss")
        monkeypatch.setattr("builtins.__import__", _no_faiss_import(__import__))
        with pytest.raises(RuntimeError, match="faiss package is not importable"):
            sel.set_backend("faiss")
    finally:
        monkeypatch.undo()
        sel.set_backend("mi355x")
    import rsx
    assert sel.engine() is rsx and sel.backend_name() == "mi355x"
    with pytest.raises(ValueError):
        sel.set_backend("cuda")

    class Ix:                       # what check_storage_dtype looks at
        def __init__(self, sd):
            self.storage_dtype = sd
    assert sel.check_storage_dtype(Ix("float16"), "auto") == "float16"
    assert sel.check_storage_dtype(Ix("float32"), None) == "float32"          # auto: data forced fp32 rows — fine
    assert sel.check_storage_dtype(Ix("float16"), "float16") == "float16"
    with pytest.raises(RuntimeError):
        sel.check_storage_dtype(Ix("float32"), "float16")
    with pytest.raises(NotImplementedError):
        sel.check_storage_dtype(Ix("float16"), "float32")
    with pytest.raises(ValueError):
        sel.check_storage_dtype(Ix("float16"), "bf16")


def _no_faiss_import(real):
    def imp(name, *a, **k):
        if name == "faiss":
            raise ImportError("No module named 'faiss'")
        return real(name, *a, **k)
    return imp


def test_datastore_api_twin(tmp_path, orc, fake):
    """api/api_index.py: shard selection, the {'scores','passages','IDs'} record, str / list queries, the reference's error
    for other types, and the 30-call latency protocol — with an injected encoder (a lookup into the embeddings)."""
    from api.api_index import DatastoreAPI, get_datastore, profile_time
    tmp = str(tmp_path)
    embs = write_datastore(tmp, orc)
    table = {"q five": embs[0][5], "q seven": embs[1][7]}
    calls = []

    def enc(queries):
        calls.append(list(queries))
        return np.stack([table[s] for s in queries], 0)
    ds = get_datastore(make_cfg(tmp, "Flat", [0, 1]), query_encoder_fn=enc)
    r = ds.search("q five", 3)
    assert set(r) == {"scores", "passages", "IDs"} and len(r["scores"]) == 1 and len(r["IDs"][0]) == 3
    D, I = orc.flat_search(np.stack([table["q five"], table["q seven"]]).astype(np.float32),
                           np.concatenate(embs, 0).astype(np.float32), 3, 0)
    assert r["IDs"] == [[[int(i) // 400, int(i) % 400] for i in I[0]]] and r["scores"] == [D[0].tolist()]
    r2 = ds.search(["q five", "q seven"], 3)
    assert r2["scores"] == D.tolist() and r2["passages"][1][0] == f"shard {I[1, 0] // 400} chunk {I[1, 0] % 400} é"
    with pytest.raises(AttributeError):
        ds.search(5)
    # one worker = one shard (api_index.py:23-27): ids are then relative to that shard only
    one = DatastoreAPI(make_cfg(tmp, "Flat", [0, 1]), shard_id=1, query_encoder_fn=enc)
    assert one.cfg.datastore.index.index_shard_ids == [1] and one.index.index.ntotal == 400
    assert one.search("q seven", 1)["IDs"] == [[[1, 7]]]
    n0 = len(calls)
    per_query = profile_time(ds, "q five", 3)
    assert len(calls) - n0 == 30 and per_query >= 0.0
    with pytest.raises(RuntimeError):
        DatastoreAPI(make_cfg(tmp, "Flat", [0, 1])).search("q five")


def test_search_topk_is_callable_as_main_ric_calls_it(tmp_path, orc, fake, monkeypatch):
    """ric/main_ric.py:27-29 calls `search_topk(cfg)` with the config alone: the evaluation examples come from the host
    application's `src.data.load_eval_data`, the query encoder from its Contriever loader (reference src/search.py:236-281),
    both imported on first use.  Stand-ins for the two host modules; the result must equal the injected-arguments call."""
    import types
    import torch
    import src.search as S
    tmp = str(tmp_path)
    embs = write_datastore(tmp, orc)
    data = [{"raw_query": "Alpha"}, {"raw_query": ""}, {"raw_query": "beta"}, {"raw_query": "Gamma delta"}]
    table = {"alpha": embs[0][3], "beta": embs[1][5], "gamma delta": embs[0][9]}

    class Tok:
        def batch_encode_plus(self, batch, return_tensors=None, max_length=None, padding=None, truncation=None):
            assert return_tensors == "pt" and padding and truncation and max_length == 77
            self.seen = getattr(self, "seen", []) + [list(batch)]
            return {"input_ids": torch.tensor([[sorted(table).index(t)] for t in batch])}

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor(np.stack([table[t] for t in sorted(table)]).astype(np.float32)), requires_grad=False)
        def forward(self, input_ids):
            return self.w[input_ids[:, 0]]           # contriever models return the pooled embedding itself

    tok = Tok()
    loaded = []
    def load_retriever(name):
        loaded.append(name)
        return Enc(), tok, None
    host_data = types.ModuleType("src.data"); host_data.load_eval_data = lambda cfg: [dict(ex) for ex in data]
    c0 = types.ModuleType("contriever"); c1 = types.ModuleType("contriever.src"); c2 = types.ModuleType("contriever.src.contriever")
    c2.load_retriever = load_retriever; c0.src = c1; c1.contriever = c2
    for name, mod in (("src.data", host_data), ("contriever", c0), ("contriever.src", c1), ("contriever.src.contriever", c2)):
        monkeypatch.setitem(sys.modules, name, mod)
    cfg = make_cfg(tmp, "Flat", [0, 1])
    cfg.model = NS(query_encoder="facebook/contriever-msmarco", query_tokenizer="facebook/contriever-msmarco")
    cfg.datastore.index["no_fp16"] = True       # the stand-in runs on the CPU here
    cfg.evaluation.search.update(per_gpu_batch_size=2, question_maxlength=77, lowercase=True, normalize_text=False,
                                 cache_query_embedding=True, query_embedding_save_path=os.path.join(tmp, "qemb.pkl"))
    assert S.search_topk(cfg) is None           # ONE positional argument
    assert loaded == ["facebook/contriever-msmarco"]
    assert tok.seen == [["alpha", "beta"], ["gamma delta"]]          # lowercased, batches of per_gpu_batch_size, empty query skipped
    out = S.get_search_output_path(cfg, [0, 1])
    rows = [json.loads(l) for l in open(out)]
    assert rows[1]["ctxs"] == [None] and [len(r["ctxs"]) for r in (rows[0], rows[2], rows[3])] == [3, 3, 3]
    qe = np.stack([table["alpha"], table["beta"], table["gamma delta"]]).astype(np.float32)
    Dw, Iw = orc.flat_search(qe, np.concatenate(embs, 0).astype(np.float32), 3, 0)
    for r, iw, dw in zip((rows[0], rows[2], rows[3]), Iw, Dw):
        assert [c["id"] for c in r["ctxs"]] == [[int(i) // 400, int(i) % 400] for i in iw]
        assert [c["retrieval score"] for c in r["ctxs"]] == [str(float(x)) for x in dw]
    cached = pickle.load(open(os.path.join(tmp, "qemb.pkl"), "rb"))
    assert cached.shape == (3, 32)
    # the same through the injection arguments (what the tests and the serving twin use)
    cfg2 = make_cfg(tmp, "Flat", [0, 1]); cfg2.evaluation.eval_output_dir = os.path.join(tmp, "out2")
    S.search_topk(cfg2, data=[dict(ex) for ex in data], questions_embedding=cached)
    assert [json.loads(l) for l in open(S.get_search_output_path(cfg2, [0, 1]))] == rows
    # second call: results exist -> nothing is loaded (reference :225-232)
    loaded.clear(); S.search_topk(cfg); assert loaded == []
    # outside the host application the missing module is named
    monkeypatch.delitem(sys.modules, "src.data")
    cfg3 = make_cfg(tmp, "Flat", [0, 1]); cfg3.evaluation.eval_output_dir = os.path.join(tmp, "out3")
    with pytest.raises(ImportError, match="src.data"):
        S.search_topk(cfg3)
    # build_index(cfg) is the other one-argument call of main_ric.py (:23-25)
    from src.index import build_index
    assert len(build_index(make_cfg(tmp, "Flat", [0, 1]))) == 1
