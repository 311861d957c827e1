"""Opportunistic FAISS leg of the bench (last row): when `import faiss` works on the box, hand the SAME index
to the real FAISS CPU path — the reference's engine at src/indicies/flat.py:139, ivf_flat.py:225, ivf_pq.py:230 — through a
FAISS-format file (rsx_faiss_io.write_faiss_index -> faiss.read_index), search the same queries, and record FAISS's
ids / scores / timing beside the GPU's.  This is the only route to a reference-pinned parity verdict; FAISS is absent from
this image and from the GPU boxes seen so far, so the usual outcome is {"available": false}.

Nothing here is required for the bench line and nothing in the product path imports it.
"""
import os
import tempfile
import time

import numpy as np


def faiss_leg(index, queries, k, nprobe, D_gpu, I_gpu, log=print, repeats=3, keep_file=None):
    """index: an rsx index (single-GPU handle); queries: numpy [nq, d] (any float dtype); D_gpu / I_gpu: numpy results of
    rsx_search on the same queries.  -> dict for the bench line."""
    try:
        import faiss
    except Exception as e:   # ImportError, or a broken wheel
        return {"available": False, "note": f"import faiss failed on this box: {type(e).__name__}: {e}"}
    import rsx_faiss_io as fio
    q32 = np.ascontiguousarray(queries, dtype=np.float32)
    out = {"available": True, "faiss_version": getattr(faiss, "__version__", "?")}
    path = keep_file or os.path.join(tempfile.mkdtemp(prefix="rsx_faiss_"), "index.faiss")
    try:
        t0 = time.perf_counter()
        fio.write_faiss_index(index, path)
        out["write_s"] = round(time.perf_counter() - t0, 2)
        out["file_bytes"] = os.path.getsize(path)
        t0 = time.perf_counter()
        fx = faiss.read_index(path)           # proves the writer against the real reader
        out["faiss_read_index_s"] = round(time.perf_counter() - t0, 2)
        out["faiss_ntotal"] = int(fx.ntotal)
        if hasattr(fx, "nprobe"):
            fx.nprobe = int(nprobe)
        try:
            out["threads"] = int(faiss.omp_get_max_threads())
        except Exception:
            out["threads"] = None
        fx.search(q32[:min(len(q32), 64)], k)      # page in
        times = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            Df, If = fx.search(q32, k)
            times.append(time.perf_counter() - t0)
        Df, If = np.asarray(Df), np.asarray(If)
        out["value"] = round(len(q32) / float(np.mean(times)), 3)
        out["unit"] = "queries/s"
        out["repeats_s"] = [round(t, 3) for t in times]
        out.update(compare(Df, If, np.asarray(D_gpu), np.asarray(I_gpu)))
        log(f"faiss leg: {out}")
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {e}"
    finally:
        if keep_file is None:
            try:
                os.remove(path); os.rmdir(os.path.dirname(path))
            except OSError:
                pass
    return out


def compare(Df, If, Dg, Ig):
    """FAISS (Df, If) vs this engine (Dg, Ig): the parity figures `north_star` names — ids (bit-exact for Flat), scores within
    an fp32 tolerance, and set overlap (same recall@k for IVF-PQ).  Differences confined to runs of EQUAL FAISS scores are
    counted separately: FAISS's order among exact ties is heap mechanics."""
    nq, k = If.shape
    same_rows = (If == Ig).all(1)
    set_overlap = float(np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(If, Ig)]))
    scale = np.maximum(np.abs(Df), 1e-30)
    fin = np.isfinite(Df) & np.isfinite(Dg)
    rel = np.where(fin, np.abs(Df - Dg) / scale, 0.0)
    tie_only = 0
    for r in np.nonzero(~same_rows)[0]:
        # a row differs only by ties if the id multisets agree inside every run of equal FAISS scores
        ok = np.array_equal(np.sort(Df[r]), np.sort(Dg[r])) or np.allclose(Df[r], Dg[r], rtol=1e-6, atol=0)
        if ok and sorted(If[r].tolist()) == sorted(Ig[r].tolist()):
            tie_only += 1
    return {"queries": int(nq), "ids_identical_queries": int(same_rows.sum()),
            "ids_differ_only_in_tie_order_queries": int(tie_only),
            "ids_identical": bool(same_rows.all()), "id_set_overlap_at_k": round(set_overlap, 6),
            "scores_bit_identical": bool(np.array_equal(Df, Dg)),
            "max_rel_score_diff": float(rel.max()) if rel.size else 0.0,
            "parity": "green" if (same_rows.sum() + tie_only == nq and (rel.max() if rel.size else 0.0) <= 1e-5) else "differs"}
