"""ctypes loader for oracle/gs_ref.c (CPU restatement of the reference path).
TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def lib(curve="bls12_381"):
    if curve not in _LIBS:
        p = os.path.join(HERE, "libgs_ref_%s.so" % curve)
        src = os.path.join(HERE, "gs_ref.c")
        if not os.path.exists(p) or os.path.getmtime(src) > os.path.getmtime(p):
            subprocess.check_call(["make", "-C", HERE])
        l = ctypes.CDLL(p)
        l.ref_bench_ppe.restype = ctypes.c_double
        l.ref_fpmul_count.restype = ctypes.c_uint64
        _LIBS[curve] = l
    return _LIBS[curve]


def _p(a):
    if a is None:
        return ctypes.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


def _u8(a):
    return np.ascontiguousarray(a).view(np.uint8).reshape(-1)


def sizes(curve="bls12_381"):
    out = (ctypes.c_int * 6)()
    lib(curve).ref_sizes(out)
    return [int(x) for x in out]


def commit_and_prove(curve, ty, m, n, X, Y, A, B, G, R, S, T, crs, want_coms=True):
    FQ, FR, G1, G2, GT, CRS = sizes(curve)
    kx = 2 if ty in (0, 1) else 1
    ky = 2 if ty in (0, 2) else 1
    xc = np.zeros(m * 2 * G1, np.uint8) if want_coms else None
    yc = np.zeros(n * 2 * G2, np.uint8) if want_coms else None
    pi = np.zeros(kx * 2 * G2, np.uint8)
    th = np.zeros(ky * 2 * G1, np.uint8)
    lib(curve).ref_commit_and_prove(ty, m, n, _p(_u8(X)), _p(_u8(Y)), _p(_u8(A)), _p(_u8(B)), _p(_u8(G)), _p(_u8(R)),
                                    _p(_u8(S)), _p(_u8(T)), _p(_u8(crs)), _p(xc), _p(yc), _p(pi), _p(th))
    return dict(xcoms=xc, ycoms=yc, pi=pi, theta=th)


def verify(curve, ty, m, n, A, B, G, target, xc, yc, pi, theta, crs):
    return int(lib(curve).ref_verify(ty, m, n, _p(_u8(A)), _p(_u8(B)), _p(_u8(G)), _p(_u8(target)), _p(_u8(xc)),
                                     _p(_u8(yc)), _p(_u8(pi)), _p(_u8(theta)), _p(_u8(crs))))


def g_mul(curve, group, p, k):
    FQ, FR, G1, G2, GT, CRS = sizes(curve)
    out = np.zeros(G1 if group == 1 else G2, np.uint8)
    (lib(curve).ref_g1_mul if group == 1 else lib(curve).ref_g2_mul)(_p(_u8(p)), _p(_u8(k)), _p(out))
    return out


def multi_pairing(curve, n, ps, qs):
    out = np.zeros(sizes(curve)[4], np.uint8)
    lib(curve).ref_multi_pairing(n, _p(_u8(ps)), _p(_u8(qs)), _p(out))
    return out


def pairing_sum(curve, n, xs, ys):
    out = np.zeros(4 * sizes(curve)[4], np.uint8)
    lib(curve).ref_pairing_sum(n, _p(_u8(xs)), _p(_u8(ys)), _p(out))
    return out


def left_mul(curve, group, rows, cols, lhs, col):
    FQ, FR, G1, G2, GT, CRS = sizes(curve)
    out = np.zeros(rows * 2 * (G1 if group == 1 else G2), np.uint8)
    (lib(curve).ref_left_mul_com1 if group == 1 else lib(curve).ref_left_mul_com2)(rows, cols, _p(_u8(lhs)),
                                                                                    _p(_u8(col)), _p(out))
    return out


def bench_ppe(units, m=4, n=4, threads=1, seed=20241220, curve="bls12_381"):
    ok = ctypes.c_int()
    fpm = ctypes.c_uint64()
    t = lib(curve).ref_bench_ppe(units, m, n, threads, ctypes.c_uint64(seed), ctypes.byref(ok), ctypes.byref(fpm))
    bench_ppe.last_fpmuls = int(fpm.value)
    return float(t), units, bool(ok.value)


def fr_matmul(curve, ar, ac, bc, a, b):
    out = np.zeros(ar * bc * 32, np.uint8)
    lib(curve).ref_fr_matmul(ar, ac, bc, _p(_u8(a)), _p(_u8(b)), _p(out))
    return out
