"""Generates tests/golden/*.npz and paths_golden.json.  Run from the repo root:

    python tests/golden/make_golden.py

Inputs of the numeric cases are NOT stored: they are regenerated from seeds by the deterministic
synthetic generator (oracle/rsx_oracle.c orc_synth_*, bit-identical to the GPU generator) and
pinned by a SHA-256 of their bytes; trained parameters (centroids, codebooks) and the expected
outputs ARE stored.  Expected outputs come from the CPU oracle (Tier 1 C restatement,
cross-checked here against the numpy fp64 Tier 0).  PARITY UNPINNED: no FAISS output exists to pin
against (see oracle/rsx_oracle.c header).

paths_golden.json is produced by importing the reference's own pure-Python
src/indicies/index_utils.py from /root/reference (it has no third-party imports) and recording
get_index_dir_and_embedding_paths() outputs for a few configs — data, not source.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import oracle as o  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEED_C, SEED_X, SEED_Q = 1234, 10000, 999


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def data(d, ncent, n, nq, sigma=0.5, sigma_q=0.1):
    x = o.synth_vectors(d, ncent, SEED_C, SEED_X, sigma, 0, n)
    q = o.synth_queries(d, ncent, SEED_C, SEED_X, sigma, n, SEED_Q, sigma_q, 0, nq)
    return x, q


def recall(I, Igt):
    return float(np.mean([len(set(a.tolist()) & set(b.tolist())) / len(b) for a, b in zip(I, Igt)]))


def flat_case(name, d, ncent, n, nq, k, metric):
    x, q = data(d, ncent, n, nq)
    D, I = o.flat_search(q.astype(np.float32), x.astype(np.float32), k, metric)
    D0, I0 = o.np_flat_search(q, x, k, "ip" if metric == 0 else "l2")
    assert (I == I0).all() and np.array_equal(D, D0), "Tier 1 != Tier 0"
    np.savez_compressed(os.path.join(OUT, name), d=d, ncent=ncent, n=n, nq=nq, k=k, metric=metric,
                        x_sha=sha(x), q_sha=sha(q), D=D, I=I)
    print(name, "ok")


def ivf_case(name, d, ncent, n, nq, k, nlist, nprobe, M=None):
    x, q = data(d, ncent, n, nq)
    x32, q32 = x.astype(np.float32), q.astype(np.float32)
    cen = o.kmeans(0, x32, nlist, 10, 1234)
    assign, _ = o.assign_ip(cen, x32)
    ids = np.arange(n, dtype=np.int64)
    Dgt, Igt = o.flat_search(q32, x32, k, 0)
    if M is None:
        lm = o.ListMajor(assign, ids, x32, nlist)
        D, I = o.ivfflat_search(0, cen, lm, q32, nprobe, k)
        # nprobe = nlist must reproduce the exhaustive result
        Dall, Iall = o.ivfflat_search(0, cen, lm, q32, nlist, k)
        assert (Iall == Igt).all() and np.array_equal(Dall, Dgt)
        np.savez_compressed(os.path.join(OUT, name), d=d, ncent=ncent, n=n, nq=nq, k=k, nlist=nlist, nprobe=nprobe,
                            x_sha=sha(x), q_sha=sha(q), centroids=cen, assign_sha=sha(assign), D=D, I=I,
                            recall=recall(I, Igt))
    else:
        sel = o.kmeans_sample(n, 256, 256, 1234)
        xt = x32[sel]
        at, _ = o.assign_ip(cen, xt)
        cb = o.pq_train(o.residuals(cen, xt, at), M, 25, 1234)
        codes = o.pq_encode(cb, o.residuals(cen, x32, assign))
        lm = o.ListMajor(assign, ids, codes, nlist)
        D, I = o.ivfpq_search(cen, cb, lm, q32, nprobe, k)
        Dh, Ih = o.ivfpq_search(cen, cb, lm, q32, nprobe, k, heap=True)
        assert np.array_equal(D, Dh), "heap variant scores differ"
        np.savez_compressed(os.path.join(OUT, name), d=d, ncent=ncent, n=n, nq=nq, k=k, nlist=nlist, nprobe=nprobe,
                            M=M, x_sha=sha(x), q_sha=sha(q), centroids=cen, codebooks=cb.astype(np.float32),
                            assign_sha=sha(assign), codes_sha=sha(codes), D=D, I=I, recall=recall(I, Igt))
    print(name, "ok; recall@k vs exact =", recall(I, Igt))


def edge_cases():
    rng = np.random.RandomState(7)
    d = 32
    x = (rng.randn(40, d)).astype(np.float16)
    x[10] = x[3]; x[25] = x[3]; x[39] = x[3]        # duplicated vectors -> exact score ties
    q = np.concatenate([x[3:4], (rng.randn(3, d)).astype(np.float16)], 0)
    D, I = o.flat_search(q.astype(np.float32), x.astype(np.float32), 8, 0)
    Dk, Ik = o.flat_search(q.astype(np.float32), x[:5].astype(np.float32), 8, 0)  # k > ntotal -> -1 padding
    # IVF with empty lists: 8 centroids, only 3 used
    cen = (rng.randn(8, d)).astype(np.float32)
    cen[3:] = -100.0 * np.abs(cen[3:])  # never the argmax for these data? keep deterministic via oracle anyway
    a, _ = o.assign_ip(cen, x.astype(np.float32))
    lm = o.ListMajor(a, np.arange(40, dtype=np.int64), x.astype(np.float32), 8)
    Div, Iiv = o.ivfflat_search(0, cen, lm, q.astype(np.float32), 8, 8)     # nprobe = nlist
    # merge: 3 shards, cross-shard equal scores -> earlier shard first
    Dm = np.array([[[5.0, 3.0, 1.0]], [[5.0, 3.0, 2.0]], [[4.0, 3.0, -np.inf]]], dtype=np.float32)
    Im = np.array([[[10, 11, 12]], [[20, 21, 22]], [[30, 31, -1]]], dtype=np.int64)
    Dmo, Imo = o.merge_topk(Dm, Im, 0)
    np.savez_compressed(os.path.join(OUT, "edge_cases"), x=x, q=q, D=D, I=I, Dk=Dk, Ik=Ik, cen=cen, assign=a,
                        Div=Div, Iiv=Iiv, Dm=Dm, Im=Im, Dmo=Dmo, Imo=Imo)
    print("edge_cases ok", I[0], Imo)


def paths_golden():
    ref = "/root/reference"
    if not os.path.isdir(ref):
        print("reference absent: paths_golden.json left as is")
        return
    sys.path.insert(0, ref)
    for m in ("src", "src.indicies"):
        sys.modules.pop(m, None)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_index_utils", os.path.join(ref, "src/indicies/index_utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    class NS(dict):
        __getattr__ = dict.__getitem__

    cases = []
    for index_type, shard_ids in [("Flat", [0]), ("IVFFlat", [3, 1, 2]), ("IVFPQ", [0, 1, 10, 7]), ("Flat", [5, 4])]:
        cfg = NS(datastore=NS(embedding=NS(embedding_dir="/data/emb/ds-256", prefix="passages"),
                              index=NS(index_type=index_type, index_shard_ids=shard_ids)))
        index_dir, paths = mod.get_index_dir_and_embedding_paths(cfg)
        cases.append({"index_type": index_type, "index_shard_ids": shard_ids, "embedding_dir": "/data/emb/ds-256",
                      "prefix": "passages", "index_dir": index_dir, "embedding_paths": paths})
    with open(os.path.join(OUT, "paths_golden.json"), "w") as f:
        json.dump({"generated_by": "tests/golden/make_golden.py importing /root/reference/src/indicies/index_utils.py",
                   "cases": cases}, f, indent=1)
    print("paths_golden ok")


def search_golden():
    """Golden vectors for the pure-Python driver pieces the host mirror restates, produced by
    RUNNING the reference's own function definitions: the five functions below are taken from
    /root/reference/src/search.py with `ast` (the module itself cannot be imported here: faiss, omegaconf,
    sentence_transformers, pyserini are not installed) and executed against small inputs.  The only name supplied
    from outside is `ListConfig = list` (the reference uses it solely in isinstance tests on `index_shard_ids`)."""
    import ast, logging, tempfile
    ref = "/root/reference/src/search.py"
    if not os.path.exists(ref):
        print("reference absent: search_golden.json left as is")
        return
    want = {"add_passages_to_eval_data", "get_search_output_path", "get_merged_search_output_path",
            "post_hoc_merge_topk", "safe_write_jsonl"}
    tree = ast.parse(open(ref).read())
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want], type_ignores=[])
    ns = {"os": os, "json": json, "logging": logging, "ListConfig": list}
    exec(compile(mod, ref, "exec"), ns)
    assert want <= set(ns)

    class NS(dict):
        __getattr__ = dict.__getitem__

    out = {"generated_by": "tests/golden/make_golden.py executing the function definitions of /root/reference/src/search.py"}

    # 1. ctxs records (src/search.py:126-146)
    data = [{"raw_query": f"q{i}"} for i in range(4)]
    passages = [["p a", "p b é"], ["p c", "p d"]]
    scores = [[0.75, 0.5], [1.25, -0.125]]
    db_ids = [[[0, 3], [1, 9]], [[2, 1], [0, 0]]]
    inp = {"data": json.loads(json.dumps(data)), "passages": passages, "scores": scores, "db_ids": db_ids,
           "valid_query_idx": [1, 3], "domain": "unit"}
    ns["add_passages_to_eval_data"](data, passages, scores, db_ids, [1, 3], domain="unit")
    out["add_passages"] = {"in": inp, "out": data}

    # 2. output paths (src/search.py:156-183)
    paths = []
    for shard_ids in ([0], [2, 0, 1], [[1], [0]], [[3, 4], [0, 1, 2]]):
        cfg = NS(datastore=NS(index=NS(index_shard_ids=shard_ids)),
                 evaluation=NS(eval_output_dir="/out/eval", data=NS(eval_data="/data/eval/nq_open.jsonl")))
        first = shard_ids[0] if isinstance(shard_ids[0], list) else shard_ids
        paths.append({"index_shard_ids": shard_ids, "per_index": ns["get_search_output_path"](cfg, first),
                      "merged": ns["get_merged_search_output_path"](cfg)})
    out["paths"] = paths

    # 3. multi-index merge (src/search.py:312-373): string scores, cross-shard ties, a query-less first example
    def ctx(shard, j, score):
        return {"id": [shard, j], "source": "unit", "retrieval text": f"s{shard} c{j}", "retrieval score": score}
    shard_results = {
        0: [{"raw_query": "", "ctxs": [None]},
            {"raw_query": "a", "ctxs": [ctx(0, 0, "0.9"), ctx(0, 1, "0.5"), ctx(0, 2, "0.25")]},
            {"raw_query": "b", "ctxs": [ctx(0, 3, "1.5"), ctx(0, 4, "1.5"), ctx(0, 5, "-2.0")]}],
        1: [{"raw_query": "", "ctxs": [None]},
            {"raw_query": "a", "ctxs": [ctx(1, 0, "0.9"), ctx(1, 1, "0.6"), ctx(1, 2, "0.1")]},
            {"raw_query": "b", "ctxs": [ctx(1, 3, "1.5"), ctx(1, 4, "1e-3"), ctx(1, 5, "-3")]}],
        2: [{"raw_query": "", "ctxs": [None]},
            {"raw_query": "a", "ctxs": [ctx(2, 0, "0.95"), ctx(2, 1, "0.5"), ctx(2, 2, "0.5")]},
            {"raw_query": "b", "ctxs": [ctx(2, 3, "2"), ctx(2, 4, "1.5"), ctx(2, 5, "1.5")]}],
    }
    with tempfile.TemporaryDirectory() as tmp:
        cfg = NS(datastore=NS(index=NS(index_shard_ids=[[0], [1], [2]])),
                 evaluation=NS(eval_output_dir=os.path.join(tmp, "eval"), data=NS(eval_data="/data/eval/unit.jsonl"),
                               search=NS(overwrite=True, n_docs=3)))
        for sid, exs in shard_results.items():
            pth = ns["get_search_output_path"](cfg, [sid])
            os.makedirs(os.path.dirname(pth), exist_ok=True)
            with open(pth, "w") as f:
                for ex in exs:
                    f.write(json.dumps(ex) + "\n")
        ns["post_hoc_merge_topk"](cfg)
        merged_path = ns["get_merged_search_output_path"](cfg)
        merged = [json.loads(l) for l in open(merged_path)]
        out["merge"] = {"n_docs": 3, "index_shard_ids": [[0], [1], [2]], "shard_results": {str(k): v for k, v in shard_results.items()},
                        "merged_relpath": os.path.relpath(merged_path, tmp), "merged": merged}
    # 4. the serving twin of the merge: api/serve_main_node.py:109-165 rerank_elements (pure Python), 3 shards x 4 queries
    #    x 5 results with many equal scores: the order the GPU merge kernel / merge_topk_host must reproduce
    api = "/root/reference/api/serve_main_node.py"
    tree2 = ast.parse(open(api).read())
    ns2 = {}
    exec(compile(ast.Module(body=[n for n in tree2.body if isinstance(n, ast.FunctionDef) and n.name == "rerank_elements"],
                            type_ignores=[]), api, "exec"), ns2)
    rng = np.random.RandomState(7)
    D = np.sort(rng.randint(0, 6, size=(3, 4, 5)).astype(np.float32) * 0.5, axis=2)[:, :, ::-1].copy()
    I = rng.randint(0, 10 ** 6, size=(3, 4, 5)).astype(np.int64)
    elems = [{"IDs": I[s_].tolist(), "passages": [[f"p{i}" for i in row] for row in I[s_].tolist()], "scores": D[s_].tolist()}
             for s_ in range(3)]
    rr = ns2["rerank_elements"](elems, k=5)
    out["rerank_elements"] = {"D": D.tolist(), "I": I.tolist(), "k": 5, "IDs": rr["IDs"], "scores": rr["scores"],
                              "passages_first_row": rr["passages"][0]}
    with open(os.path.join(OUT, "search_golden.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("search_golden ok:", [c["id"] for c in out["merge"]["merged"][2]["ctxs"]], rr["scores"][0])


def flat_indexer_golden():
    """Host behaviour of the reference's OWN FlatIndexer class (src/indicies/flat.py), executed here with the test
    double tests/fake_engine.py registered as the `faiss` module (the one import the file needs that the image
    lacks; its surface is the subset of FAISS the class calls: IndexFlatIP / add / search / read_index / write_index,
    arithmetic by the oracle).  Recorded: files written, the id map, the passage position map, search() returns, and
    that a second construction loads instead of rebuilding.  Pins SURVEY 8 rows a1, a2, a5, a8 (Flat) of the host
    mirror to the reference's behaviour, not to a reading of it."""
    import pickle, tempfile
    ref = "/root/reference"
    if not os.path.isdir(ref):
        print("reference absent: flat_indexer_golden.json left as is")
        return
    tests_dir = os.path.dirname(OUT)
    sys.path.insert(0, tests_dir)
    import fake_engine
    sys.modules["faiss"] = fake_engine
    for m in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        sys.modules.pop(m)
    sys.path.insert(0, ref)
    import importlib
    ref_flat = importlib.import_module("src.indicies.flat")
    assert ref_flat.__file__.startswith(ref)
    n_shards, per, d = 2, 400, 32
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "emb")); os.makedirs(os.path.join(tmp, "psg")); os.makedirs(os.path.join(tmp, "index"))
        embs = []
        for sh in range(n_shards):   # the same datastore tests/test_host_logic.py::write_datastore builds
            e = o.synth_vectors(d, 6, 11, 100 + sh, 0.5, 0, per)
            embs.append(e)
            with open(os.path.join(tmp, "emb", f"passages_{sh:02d}.pkl"), "wb") as f:
                pickle.dump((list(range(per)), e), f)
            with open(os.path.join(tmp, "psg", f"raw_passages-{sh}-of-{n_shards}.pkl"), "wb") as f:
                pickle.dump([{"text": f"shard {sh} chunk {c} é", "id": c} for c in range(per)], f)
        kw = dict(embed_paths=[os.path.join(tmp, "emb", f"passages_{sh:02d}.pkl") for sh in (1, 0)],   # given order = id order
                  index_path=os.path.join(tmp, "index", "index_Flat.faiss"),
                  meta_file=os.path.join(tmp, "index", "index_Flat.faiss.meta"),
                  passage_dir=os.path.join(tmp, "psg"),
                  pos_map_save_path=os.path.join(tmp, "index", "passage_pos_id_map.pkl"), dimension=d)
        ix = ref_flat.FlatIndexer(**kw)
        q = np.concatenate([embs[0][5:6], embs[1][7:8], embs[1][399:400]], 0)
        scores, passages, db_ids = ix.search(q, k=3)
        files = sorted(os.listdir(os.path.join(tmp, "index")))
        meta = pickle.load(open(kw["meta_file"], "rb"))
        pos = pickle.load(open(kw["pos_map_save_path"], "rb"))
        pos_rel = {str(sh): {str(c): [os.path.relpath(v[0], tmp), v[1]] for c, v in m.items() if c in (0, 1, 2, 399)} for sh, m in pos.items()}
        mtime = os.path.getmtime(kw["index_path"])
        ix2 = ref_flat.FlatIndexer(**kw)
        s2, p2, d2 = ix2.search(q, k=3)
        assert os.path.getmtime(kw["index_path"]) == mtime and (s2, p2, d2) == (scores, passages, db_ids)
        psg_files = sorted(os.listdir(os.path.join(tmp, "psg")))
    out = {"generated_by": "tests/golden/make_golden.py running /root/reference/src/indicies/flat.py::FlatIndexer with tests/fake_engine.py as `faiss`",
           "datastore": {"n_shards": n_shards, "per": per, "d": d, "embed_order": [1, 0], "query_rows": [[0, 5], [1, 7], [1, 399]]},
           "index_dir_files": files, "passage_dir_files": psg_files,
           "meta_len": len(meta), "meta_head": meta[:3], "meta_at_400": meta[400], "meta_sha": sha(np.asarray(meta, dtype=np.int64)),
           "pos_map_sample": pos_rel, "k": 3, "scores": scores, "passages": passages, "db_ids": db_ids,
           "attrs": {"cuda": ix.cuda, "ntotal": int(ix.index.ntotal)}}
    with open(os.path.join(OUT, "flat_indexer_golden.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("flat_indexer_golden ok:", db_ids[0], files)


def indexer_facade_golden():
    """The reference's OWN facade and backends — src/indicies/base.py::Indexer dispatching to FlatIndexer,
    IVFFlatIndexer and IVFPQIndexer — executed on a small datastore with tests/fake_engine.py registered as `faiss`
    and a two-line `omegaconf` module (ListConfig = list; the backends import the name, the code paths used here
    never touch it).  numpy's global RNG is seeded before each construction so that the reference's
    np.random.choice training sample (ivf_flat.py:132, ivf_pq.py:135) is reproducible; the mirror draws the same
    sample under the same seed, so the whole pipeline — sample, train, add, file names, id maps, search() returns —
    must coincide.  Pins SURVEY 8 rows a1-a5, a8, a9."""
    import pickle, tempfile, types
    ref = "/root/reference"
    if not os.path.isdir(ref):
        print("reference absent: indexer_facade_golden.json left as is")
        return
    sys.path.insert(0, os.path.dirname(OUT))
    import fake_engine
    sys.modules["faiss"] = fake_engine
    om = types.ModuleType("omegaconf"); om.ListConfig = list
    sys.modules.setdefault("omegaconf", om)
    for m in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        sys.modules.pop(m)
    if ref in sys.path:
        sys.path.remove(ref)
    sys.path.insert(0, ref)
    import importlib
    ref_base = importlib.import_module("src.indicies.base")
    assert ref_base.__file__.startswith(ref)

    class NS(dict):
        __getattr__ = dict.__getitem__

    n_shards, per, d = 2, 400, 32
    out = {"generated_by": "tests/golden/make_golden.py running /root/reference/src/indicies/base.py::Indexer (+ flat.py, ivf_flat.py, "
                           "ivf_pq.py) with tests/fake_engine.py as `faiss`",
           "datastore": {"n_shards": n_shards, "per": per, "d": d, "query_rows": [[0, 5], [1, 7], [1, 399]]},
           "index_args": {"projection_size": d, "sample_train_size": 600, "ncentroids": 4, "probe": 4, "n_subquantizers": 4, "n_bits": 8},
           "np_random_seed": 4242, "k": 3, "cases": []}
    for index_type, shard_ids in (("Flat", [1, 0]), ("IVFFlat", [1, 0]), ("IVFPQ", [0, 1])):
        with tempfile.TemporaryDirectory() as tmp:
            os.makedirs(os.path.join(tmp, "emb")); os.makedirs(os.path.join(tmp, "psg"))
            embs = []
            for sh in range(n_shards):   # = tests/test_host_logic.py::write_datastore
                e = o.synth_vectors(d, 6, 11, 100 + sh, 0.5, 0, per)
                embs.append(e)
                with open(os.path.join(tmp, "emb", f"passages_{sh:02d}.pkl"), "wb") as f:
                    pickle.dump((list(range(per)), e), f)
                with open(os.path.join(tmp, "psg", f"raw_passages-{sh}-of-{n_shards}.pkl"), "wb") as f:
                    pickle.dump([{"text": f"shard {sh} chunk {c} é", "id": c} for c in range(per)], f)
            cfg = NS(datastore=NS(domain="unit",
                                  embedding=NS(embedding_dir=os.path.join(tmp, "emb"), prefix="passages", passages_dir=os.path.join(tmp, "psg")),
                                  index=NS(index_type=index_type, index_shard_ids=shard_ids, **out["index_args"])))
            np.random.seed(out["np_random_seed"])
            ix = ref_base.Indexer(cfg)
            q = np.concatenate([embs[0][5:6], embs[1][7:8], embs[1][399:400]], 0)
            scores, passages, db_ids = ix.search(q, k=3)
            index_dir = None
            for root, dirs, files in os.walk(os.path.join(tmp, "emb")):
                if any(f.endswith(".faiss") for f in files):
                    index_dir = root
            meta_name = [f for f in os.listdir(index_dir) if f.endswith(".meta")][0]
            meta = pickle.load(open(os.path.join(index_dir, meta_name), "rb"))
            ds = ix.datastore
            out["cases"].append({
                "index_type": index_type, "index_shard_ids": shard_ids,
                "index_dir": os.path.relpath(index_dir, tmp), "index_dir_files": sorted(os.listdir(index_dir)),
                "meta_len": len(meta), "meta_head": meta[:2], "meta_at_400": meta[400], "meta_sha": sha(np.asarray(meta, dtype=np.int64)),
                "ntotal": int(ds.index.ntotal), "nprobe": int(getattr(ds.index, "nprobe", 0)), "probe_attr": getattr(ds, "probe", None),
                "scores": scores, "passages": passages, "db_ids": db_ids})
            print("indexer_facade_golden", index_type, db_ids[0], sorted(os.listdir(index_dir)))
    with open(os.path.join(OUT, "indexer_facade_golden.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)


if __name__ == "__main__":
    flat_case("flat_ip_d768", 768, 16, 4096, 32, 10, 0)
    flat_case("flat_l2_d64", 64, 16, 2048, 16, 5, 1)
    flat_case("flat_ip_d100", 100, 8, 1000, 40, 7, 0)
    ivf_case("ivfflat_d768", 768, 16, 4096, 32, 10, 16, 4)
    ivf_case("ivfpq_d768_m96", 768, 16, 4096, 32, 10, 16, 4, M=96)
    ivf_case("ivfpq_d64_m16", 64, 16, 8192, 32, 10, 32, 8, M=16)
    edge_cases()
    paths_golden()
    search_golden()
    flat_indexer_golden()
    indexer_facade_golden()
