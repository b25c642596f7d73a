"""MI355X-native Groth-Sahai prove/verify engine (hot path of jdwhite48/groth-sahai-rs).

The product is the C-ABI shared library built from csrc/ (include/gs_amd.h);
this package is the thin ctypes binding plus a host-side mirror of the
reference's Provable / Verifiable / batch_commit_* interface.  There is no CPU
fallback: without the HIP library (or without a GPU) every compute call fails.
"""
from .capi import Engine, GsError, MultiEngine, load_library, lib_path  # noqa: F401
from .capi import GS_PPE, GS_MSMEG1, GS_MSMEG2, GS_QUAD, CURVE_BLS12_381, CURVE_BN254  # noqa: F401
