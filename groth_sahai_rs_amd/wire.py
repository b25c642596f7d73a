"""Canonical (de)serialisation of the reference's serialisable types, byte-compatible with what
`#[derive(CanonicalSerialize, CanonicalDeserialize)]` produces for them (ark-serialize ^0.5):

    Com1 / Com2                 src/data_structures.rs:128-133
    Commit1 / Commit2           src/prover/commit.rs:18-28      {coms: Vec<Com>, rand: Matrix<Fr>}
    EquProof                    src/prover/prove.rs:55-61       {pi, theta, equ_type, rand}
    EquType                     src/statement.rs:61-97          one byte 0..3
    PPE / MSMEG1 / MSMEG2 / QuadEqu   src/statement.rs:117-192  {a_consts, b_consts, gamma, target}
    CRS                         src/generator.rs:35-42          {u, v, g1_gen, g2_gen, gt_gen}

This module owns the framing only (Vec<T> = u64-LE length + items, fields in declaration order);
every group / field element goes through the GPU library's array converters (gs_wire_* in
include/gs_amd.h), batched per kind.  `serialize_compressed` / `serialize_uncompressed` /
`deserialize_compressed` / `deserialize_uncompressed` follow ark-serialize's names;
deserialisation validates like Validate::Yes and raises SerializationError("InvalidData")
where arkworks returns that error.  Values are the mirror's (numpy uint64 limb arrays)."""
import struct

import numpy as np

from . import mirror
from .capi import GS_MSMEG1, GS_MSMEG2, GS_PPE, GS_QUAD


class SerializationError(ValueError):
    pass


# ---- schema: a tree of ("kind", value) leaves in wire order ------------------------------------------
def _vec(items, f):
    return [("len", len(items))] + [t for x in items for t in f(x)]


def _mat_fr(m):
    return _vec(m, lambda row: _vec(row, lambda e: [("fr", e)]))


def _com(kind):
    return lambda c: [(kind, np.asarray(c, dtype=np.uint64).reshape(2, -1)[0]),
                      (kind, np.asarray(c, dtype=np.uint64).reshape(2, -1)[1])]


def _leaves(obj):
    if isinstance(obj, mirror.Commit2):
        return _vec(obj.coms, _com("g2")) + _mat_fr(obj.rand)
    if isinstance(obj, mirror.Commit1):
        return _vec(obj.coms, _com("g1")) + _mat_fr(obj.rand)
    if isinstance(obj, mirror.EquProof):
        return _vec(obj.pi, _com("g2")) + _vec(obj.theta, _com("g1")) + [("u8", obj.equ_type)] + _mat_fr(obj.rand)
    if isinstance(obj, mirror.CRS):
        return (_vec(obj.u, _com("g1")) + _vec(obj.v, _com("g2")) + [("g1", obj.g1_gen), ("g2", obj.g2_gen),
                                                                  ("gt", obj.gt_gen)])
    if isinstance(obj, mirror._Equation):
        ka, kb, kt = _EQU_KINDS[obj.TYPE]
        return (_vec(obj.a_consts, lambda e: [(ka, e)]) + _vec(obj.b_consts, lambda e: [(kb, e)]) + _mat_fr(obj.gamma)
                + [(kt, obj.target)])
    raise TypeError("not a serialisable Groth-Sahai type: %r" % type(obj))


_EQU_KINDS = {GS_PPE: ("g1", "g2", "gt"), GS_MSMEG1: ("g1", "fr", "g1"), GS_MSMEG2: ("fr", "g2", "g2"),
              GS_QUAD: ("fr", "fr", "fr")}


def _serialize(obj, engine, compressed):
    leaves = _leaves(obj)
    enc = {}
    for kind in ("g1", "g2", "fr", "gt"):
        vals = [np.asarray(v, dtype=np.uint64).reshape(-1) for k, v in leaves if k == kind]
        if vals:
            enc[kind] = iter(engine.wire_encode(kind, np.stack(vals), compressed))
    out = []
    for k, v in leaves:
        if k == "len":
            out.append(struct.pack("<Q", v))
        elif k == "u8":
            out.append(bytes([v]))
        else:
            out.append(next(enc[k]).tobytes())
    return b"".join(out)


def serialize_compressed(obj, engine=None):
    return _serialize(obj, engine or obj.engine, True)


def serialize_uncompressed(obj, engine=None):
    return _serialize(obj, engine or obj.engine, False)


# ---- deserialisation: a cursor that records element slices, then one batched decode per kind ---------
class _Reader:
    def __init__(self, data, engine, compressed):
        self.b, self.o, self.eng, self.c = memoryview(data), 0, engine, compressed
        ws = engine.wire_sizes()
        self.sz = {"g1": ws["g1c" if compressed else "g1u"], "g2": ws["g2c" if compressed else "g2u"], "fr": ws["fr"],
                   "gt": ws["gt"]}
        self.req = {k: [] for k in self.sz}

    def take(self, n):
        if self.o + n > len(self.b):
            raise SerializationError("InvalidData: truncated input")
        s = self.b[self.o:self.o + n]
        self.o += n
        return s

    def length(self):
        n = struct.unpack("<Q", self.take(8))[0]
        if n > len(self.b):  # every element is at least one byte: reject absurd lengths before allocating
            raise SerializationError("InvalidData: length prefix exceeds the input")
        return n

    def u8(self):
        return self.take(1)[0]

    def elem(self, kind):
        """Reserve one element; returns a handle resolved by finish()."""
        self.req[kind].append(bytes(self.take(self.sz[kind])))
        return (kind, len(self.req[kind]) - 1)

    def vec(self, f):
        return [f() for _ in range(self.length())]

    def mat_fr(self):
        return self.vec(lambda: self.vec(lambda: self.elem("fr")))

    def com(self, kind):
        return (self.elem(kind), self.elem(kind))

    def finish(self, validate):
        if self.o != len(self.b):
            raise SerializationError("InvalidData: trailing bytes")
        self.val = {}
        for kind, items in self.req.items():
            if not items:
                continue
            buf = np.frombuffer(b"".join(items), dtype=np.uint8).reshape(len(items), -1)
            vals, ok = self.eng.wire_decode(kind, buf, self.c, validate)
            if not ok.all():
                raise SerializationError("InvalidData: %s element %d" % (kind, int(np.argmin(ok))))
            self.val[kind] = vals.view(np.uint64)

    def get(self, h):
        return self.val[h[0]][h[1]].copy()

    def getcom(self, h):
        return np.concatenate([self.get(h[0]), self.get(h[1])])

    def getmat(self, m):
        return [[self.get(e) for e in row] for row in m]


def _deserialize(cls, data, engine, compressed, validate=True, curve=0, device=0):
    if cls is mirror.CRS and engine is None:
        from .capi import Engine
        engine = Engine(curve, device)
    r = _Reader(data, engine, compressed)
    if cls in (mirror.Commit1, mirror.Commit2):
        kind = "g1" if cls is mirror.Commit1 else "g2"
        coms, rand = r.vec(lambda: r.com(kind)), r.mat_fr()
        r.finish(validate)
        return cls([r.getcom(c) for c in coms], r.getmat(rand))
    if cls is mirror.EquProof:
        pi, theta = r.vec(lambda: r.com("g2")), r.vec(lambda: r.com("g1"))
        ty = r.u8()
        if ty > 3:
            raise SerializationError("InvalidData: EquType %d" % ty)
        rand = r.mat_fr()
        r.finish(validate)
        return cls([r.getcom(c) for c in pi], [r.getcom(c) for c in theta], ty, r.getmat(rand))
    if cls is mirror.CRS:
        u, v = r.vec(lambda: r.com("g1")), r.vec(lambda: r.com("g2"))
        g1, g2, gt = r.elem("g1"), r.elem("g2"), r.elem("gt")
        r.finish(validate)
        if len(u) != 2 or len(v) != 2:
            raise SerializationError("InvalidData: the SXDH CRS has two keys per group")
        return cls([r.getcom(c) for c in u], [r.getcom(c) for c in v], r.get(g1), r.get(g2), r.get(gt), engine.curve,
                   device)
    if issubclass(cls, mirror._Equation):
        ka, kb, kt = _EQU_KINDS[cls.TYPE]
        a, b = r.vec(lambda: r.elem(ka)), r.vec(lambda: r.elem(kb))
        gamma, t = r.mat_fr(), r.elem(kt)
        r.finish(validate)
        return cls([r.get(h) for h in a], [r.get(h) for h in b], r.getmat(gamma), r.get(t))
    raise TypeError("not a deserialisable Groth-Sahai type: %r" % cls)


def deserialize_compressed(cls, data, engine=None, validate=True, **kw):
    return _deserialize(cls, data, engine, True, validate, **kw)


def deserialize_uncompressed(cls, data, engine=None, validate=True, **kw):
    return _deserialize(cls, data, engine, False, validate, **kw)
