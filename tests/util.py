"""Shared helpers for the tests (golden loading, data regeneration, comparisons)."""
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED_C, SEED_X, SEED_Q = 1234, 10000, 999


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: (g[k].item() if g[k].shape == () else g[k]) for k in g.files}


def regen(orc, g, sigma=0.5, sigma_q=0.1):
    """Regenerate a golden case's inputs with the CPU generator and check their pinned hashes."""
    x = orc.synth_vectors(g["d"], g["ncent"], SEED_C, SEED_X, sigma, 0, g["n"])
    q = orc.synth_queries(g["d"], g["ncent"], SEED_C, SEED_X, sigma, g["n"], SEED_Q, sigma_q, 0, g["nq"])
    assert sha(x) == g["x_sha"] and sha(q) == g["q_sha"], "synthetic generator drifted from the golden inputs"
    return x, q


def regen_gpu(rsx, g, sigma=0.5, sigma_q=0.1):
    """Same inputs from the HIP generator (must be bit-identical to the CPU generator)."""
    x = rsx.synth_vectors(g["d"], g["ncent"], SEED_C, SEED_X, sigma, 0, g["n"])
    q = rsx.synth_queries(g["d"], g["ncent"], SEED_C, SEED_X, sigma, g["n"], SEED_Q, sigma_q, 0, g["nq"])
    assert sha(x) == g["x_sha"] and sha(q) == g["q_sha"], "HIP synthetic generator differs from the oracle's"
    return x, q


def assert_same_results(D, I, Dref, Iref, what=""):
    """Bit-exact ids and scores (the bar for integer/index work and for canonical scores)."""
    assert D.shape == Dref.shape and I.shape == Iref.shape, what
    bad = np.nonzero((I != Iref).any(axis=1))[0]
    assert len(bad) == 0, f"{what}: ids differ for {len(bad)} queries, first q={bad[:3]}: got {I[bad[0]]} want {Iref[bad[0]]} (D {D[bad[0]]} vs {Dref[bad[0]]})"
    same = (D == Dref) | (np.isinf(D) & np.isinf(Dref) & (np.sign(D) == np.sign(Dref)))
    assert same.all(), f"{what}: scores differ, max abs {np.nanmax(np.abs(np.where(same, 0, D - Dref)))}"
