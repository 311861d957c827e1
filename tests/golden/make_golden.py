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
    with open(os.path.join(OUT, "search_golden.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("search_golden ok:", [c["id"] for c in out["merge"]["merged"][2]["ctxs"]])


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
