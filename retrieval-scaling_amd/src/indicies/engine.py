"""Engine selection for the index backends (optional config keys with behaviour-preserving defaults).

cfg.datastore.index.backend
    "mi355x" (default; also "rsx", "hip")  the HIP engine behind include/rsx.h (retrieval-scaling_amd/rsx.py)
    "faiss"                                the reference's own engine — every call the backends make has the same name and
                                           arguments in both modules — if the faiss package is importable
                                           (it is not in this image: a clear error instead of a silent fallback)

The backends call `engine().IndexFlatIP(...)`, `engine().read_index(...)`, ... instead of naming a module.
"""
import rsx as _rsx

_engine = _rsx
_name = "mi355x"


def engine():
    return _engine


def backend_name():
    return _name


def set_backend(name):
    """Select the engine module for indexes created from now on; returns the module."""
    global _engine, _name
    key = (name or "mi355x").strip().lower()
    if key in ("mi355x", "rsx", "hip"):
        _engine, _name = _rsx, "mi355x"
    elif key == "faiss":
        try:
            import faiss
        except ImportError as e:
            raise RuntimeError("cfg.datastore.index.backend = 'faiss', but the faiss package is not importable here "
                               "(the reference pins faiss 1.8.0: environment.yml:11); use backend 'mi355x'") from e
        _engine, _name = faiss, "faiss"
    else:
        raise ValueError(f"cfg.datastore.index.backend = {name!r}: expected 'mi355x' or 'faiss'")
    return _engine


def check_storage_dtype(index, want):
    """cfg.datastore.index.storage_dtype: "auto" (default) keeps vectors as fp16 rows while every value added is
    fp16-representable (the reference's embeddings are) and widens them to fp32 otherwise; "float16" additionally REQUIRES
    that outcome (raises if the data forced fp32 rows: twice the HBM); "float32" is what FAISS stores and is not offered
    as a forced mode (100M x 768 fp32 does not fit one GPU) — lossless fp16 storage returns the same results."""
    key = (want or "auto").strip().lower()
    if key not in ("auto", "float16", "fp16", "half", "float32", "fp32"):
        raise ValueError(f"cfg.datastore.index.storage_dtype = {want!r}: expected 'auto', 'float16' or 'float32'")
    if key in ("float32", "fp32"):
        raise NotImplementedError("storage_dtype 'float32' cannot be forced: rows are widened to fp32 automatically when a value "
                                  "is not fp16-representable, and stay fp16 (lossless, half the HBM) otherwise")
    have = getattr(index, "storage_dtype", None)
    if key in ("float16", "fp16", "half") and have == "float32":
        raise RuntimeError("storage_dtype 'float16' was requested but the added vectors hold values fp16 cannot represent: "
                           "the index keeps fp32 rows")
    return have
