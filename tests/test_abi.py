"""CPU: the C-ABI library loads and exports exactly what include/rsx.h declares (no compute)."""
import os
import re
import subprocess

from conftest import PKG, REPO


def header_symbols():
    text = open(os.path.join(REPO, "include", "rsx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsx_[a-z_0-9]+)\s*\(", text)))


def test_header_matches_binding(rsxlib):
    assert header_symbols() == sorted(rsxlib.ABI_SYMBOLS)


def test_library_exports_every_symbol(rsxlib):
    so = os.path.join(PKG, "csrc", "librsx.so")
    out = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    exported = set(re.findall(r" T (rsx_[a-z_0-9]+)", out))
    missing = [s for s in header_symbols() if s not in exported]
    assert not missing, missing
    L = rsxlib.lib()
    assert L.rsx_version() >= 1000
    for s in header_symbols():
        assert getattr(L, s) is not None


def test_no_cpu_fallback(rsxlib):
    """Without a GPU every constructor fails loudly; with one this test is vacuous."""
    if rsxlib.get_num_gpus() > 0:
        return
    import pytest
    with pytest.raises(RuntimeError, match="no HIP device"):
        rsxlib.IndexFlatIP(16)
    with pytest.raises(RuntimeError, match="no HIP device"):
        rsxlib.IndexIVFPQ(None, 16, 4, 4, 8, rsxlib.METRIC_INNER_PRODUCT)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the package may import, load or call it."""
    bad = []
    for root, _, files in os.walk(PKG):
        for f in files:
            if not (f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile"):
                continue
            txt = open(os.path.join(root, f), errors="replace").read()
            if re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M) or "liborc" in txt or "orc_" in txt.replace("orc_synth", ""):
                bad.append(os.path.join(root, f))
    assert not bad, bad


def test_shipped_library_has_no_measurement_switches(rsxlib):
    """RSX_ROT_VARIANT / RSX_SCAN8_VARIANT / RSX_*_V1 used to sit in the production kernels' launchers — an environment
    variable could make rsx_search skip part of its work.  They now exist only in the -DRSX_MEASURE build (librsx_measure.so,
    loaded by tools/ through RSX_LIB): the shipped librsx.so must not even contain the variable names.  RSX_PQ_LAYOUT (which
    of two result-identical code layouts a NEW index uses) is the one documented environment knob."""
    import subprocess
    lib = os.path.join(PKG, "csrc", "librsx.so")
    names = set(re.findall(rb"RSX_[A-Z0-9_]+", open(lib, "rb").read()))
    assert names <= {b"RSX_PQ_LAYOUT", b"RSX_OK"} | {n for n in names if n.startswith(b"RSX_ERR_") or n.startswith(b"RSX_METRIC") or n in (b"RSX_F16", b"RSX_F32")}, names
    # the measurement build is a separate target, never loaded by the product
    assert "librsx_measure.so" in open(os.path.join(PKG, "csrc", "Makefile")).read()
    assert "librsx_measure" not in open(os.path.join(PKG, "rsx.py")).read()
