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


def test_capi_rejects_short_buffers_before_the_c_abi():
    """The C ABI takes bare pointers; proof lengths come from the wire.  The ctypes layer must refuse any array whose
    byte length disagrees with (N, m, n, type) -- GS_ERR_SHAPE, where the reference panics in pairing_sum / left_mul
    (data_structures.rs:495,705) -- before a pointer is handed over.  No GPU needed: the checks run first."""
    import numpy as np

    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd.capi import Engine

    e = object.__new__(Engine)  # sizes only: no context (there is no GPU here)
    e.FQ, e.FR, e.G1, e.G2, e.GT, e.CRS = 48, 32, 96, 192, 576, 2016
    e.COM1, e.COM2 = 192, 384
    z = lambda n: np.zeros(n, dtype=np.uint8)
    m, n = 2, 1
    good = dict(A=z(n * 96), B=z(m * 192), Gamma=z(m * n * 32), target=z(576), xcoms=z(m * 192), ycoms=z(n * 384),
                pi=z(2 * 384), theta=z(2 * 192))
    e._check_verify("t", 0, 1, m, n, **good)  # consistent: passes
    for name, short in (("pi", 384), ("theta", 192), ("A", 0), ("B", 192), ("Gamma", 32), ("target", 575),
                        ("xcoms", 192), ("ycoms", 385)):
        bad = dict(good)
        bad[name] = z(short)
        with pytest.raises(gs.GsError) as ei:
            e._check_verify("t", 0, 1, m, n, **bad)
        assert ei.value.code == 1 and name in str(ei.value)
    with pytest.raises(gs.GsError):
        e._check_verify("t", 0, 1, 0, 1, **good)  # empty variable list
    with pytest.raises(gs.GsError):
        e._check_verify("t", 9, 1, m, n, **good)  # unknown equation type
    pv = dict(X=z(m * 96), Y=z(n * 192), A=z(n * 96), B=z(m * 192), Gamma=z(m * n * 32), R=z(m * 2 * 32),
              S=z(n * 2 * 32), T=z(4 * 32))
    e._check_prove("t", 0, 1, m, n, **pv)
    for name in pv:
        bad = dict(pv)
        bad[name] = z(pv[name].size - 32)
        with pytest.raises(gs.GsError) as ei:
            e._check_prove("t", 0, 1, m, n, **bad)
        assert ei.value.code == 1
    # MSMEG2: x side scalars (kx = 1), target in G2
    e._check_verify("t", 2, 3, m, n, A=z(3 * n * 32), B=z(3 * m * 192), Gamma=z(3 * m * n * 32), target=z(3 * 192),
                    xcoms=z(3 * m * 192), ycoms=z(3 * n * 384), pi=z(3 * 384), theta=z(3 * 2 * 192))
    e.ctx = None  # nothing to destroy
