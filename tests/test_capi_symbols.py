"""CPU checks of the boundary: the C-ABI library loads here (hipcc cross-compiles
without a GPU), exports every function include/gs_amd.h declares, and fails
loudly -- not silently on a CPU path -- when no GPU is present."""
import os
import re

import pytest

from gsutil import REPO


def declared_functions():
    src = open(os.path.join(REPO, "include", "gs_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd.capi import SYMBOLS

    lib = gs.load_library()
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert sorted(SYMBOLS) == names  # the ctypes binding lists exactly the header's functions
    assert b"gfx950" in lib.gs_version()


def test_sizes_match_survey_layout():
    import ctypes

    import groth_sahai_rs_amd as gs

    lib = gs.load_library()
    sz = (ctypes.c_size_t * 6)()
    assert lib.gs_sizes(0, sz) == 0
    assert list(sz) == [48, 32, 96, 192, 576, 2016]  # Fq, Fr, G1, G2, GT, CRS (BLS12-381)
    assert lib.gs_sizes(1, sz) == 0
    assert list(sz) == [32, 32, 64, 128, 384, 1344]  # BN254
    assert lib.gs_sizes(7, sz) != 0


def test_no_cpu_fallback():
    """Without a GPU the product must refuse to compute (GS_ERR_DEVICE), never fall back."""
    import torch

    import groth_sahai_rs_amd as gs

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(gs.GsError) as ei:
        gs.Engine(0, 0)
    assert ei.value.code == 2


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under groth_sahai_rs_amd/ may import, include, link or open it."""
    pkg = os.path.join(REPO, "groth_sahai_rs_amd")
    bad = re.compile(r'^\s*(import|from)\s+(gs_oracle|gs_ref\w*|oracle)\b|#include\s+"[^"]*(oracle|gs_ref)[^"]*"|'
                     r'["\']oracle["\']|libgs_ref|CDLL\([^)]*ref', re.M)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cuh", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(root, f)).read()
                m = bad.search(txt)
                assert m is None, (root, f, m.group(0))
