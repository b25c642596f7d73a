"""Shared helpers for the tests: boundary (Montgomery-limb) encodings of the
canonical integers held in tests/golden/*.json.  Layout = SURVEY.md 8a-3:
little-endian u64 limbs of a*2^(64N) mod p; G1 affine = x||y (identity =
all-zero); G2 = x.c0||x.c1||y.c0||y.c1; Fp12 = 12 Fq in arkworks order."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


class CurveInfo:
    def __init__(self, name):
        self.name = name
        with open(os.path.join(HERE, "golden", name + ".json")) as f:
            self.golden = json.load(f)
        self.p = int(self.golden["p"], 16)
        self.r = int(self.golden["r"], 16)
        self.nq = (self.p.bit_length() + 63) // 64  # u64 limbs
        self.nr = (self.r.bit_length() + 63) // 64
        self.Rq = 1 << (64 * self.nq)
        self.Rr = 1 << (64 * self.nr)
        self.curve_id = 0 if name == "bls12_381" else 1

    # ---- scalars ----
    def _limbs(self, v, n):
        return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]

    def fq(self, v):
        return np.array(self._limbs(v * self.Rq % self.p, self.nq), dtype=np.uint64)

    def fr(self, v):
        return np.array(self._limbs(v * self.Rr % self.r, self.nr), dtype=np.uint64)

    def fq_dec(self, a):
        a = np.asarray(a, dtype=np.uint64).reshape(-1)
        v = sum(int(x) << (64 * i) for i, x in enumerate(a))
        return v * pow(self.Rq, -1, self.p) % self.p

    def fr_dec(self, a):
        a = np.asarray(a, dtype=np.uint64).reshape(-1)
        v = sum(int(x) << (64 * i) for i, x in enumerate(a))
        return v * pow(self.Rr, -1, self.r) % self.r

    # ---- points (from golden hex lists; None = identity) ----
    def g1(self, h):
        if h is None:
            return np.zeros(2 * self.nq, dtype=np.uint64)
        return np.concatenate([self.fq(int(h[0], 16)), self.fq(int(h[1], 16))])

    def g2(self, h):
        if h is None:
            return np.zeros(4 * self.nq, dtype=np.uint64)
        return np.concatenate([self.fq(int(s, 16)) for s in h])

    def f12(self, h):
        return np.concatenate([self.fq(int(s, 16)) for s in h])

    def com1(self, h):
        return np.concatenate([self.g1(h[0]), self.g1(h[1])])

    def com2(self, h):
        return np.concatenate([self.g2(h[0]), self.g2(h[1])])

    def fr_hex(self, s):
        return self.fr(int(s, 16))

    def fr_mat(self, m):
        if not m:
            return np.zeros((0, self.nr), dtype=np.uint64)
        return np.stack([self.fr_hex(s) for row in m for s in row])

    # ---- decoders back to hex lists (for readable assertion messages) ----
    def g1_dec(self, a):
        a = np.asarray(a, dtype=np.uint64).reshape(2, self.nq)
        if not a.any():
            return None
        return ["%x" % self.fq_dec(a[0]), "%x" % self.fq_dec(a[1])]

    def g2_dec(self, a):
        a = np.asarray(a, dtype=np.uint64).reshape(4, self.nq)
        if not a.any():
            return None
        return ["%x" % self.fq_dec(x) for x in a]

    def com1_dec(self, a):
        a = np.asarray(a, dtype=np.uint64).reshape(2, 2 * self.nq)
        return [self.g1_dec(a[0]), self.g1_dec(a[1])]

    def com2_dec(self, a):
        a = np.asarray(a, dtype=np.uint64).reshape(2, 4 * self.nq)
        return [self.g2_dec(a[0]), self.g2_dec(a[1])]

    def f12_dec(self, a):
        a = np.asarray(a, dtype=np.uint64).reshape(12, self.nq)
        return ["%x" % self.fq_dec(x) for x in a]


_CACHE = {}


def curve(name):
    if name not in _CACHE:
        _CACHE[name] = CurveInfo(name)
    return _CACHE[name]


def ptr(a):
    import ctypes

    return a.ctypes.data_as(ctypes.c_void_p)
