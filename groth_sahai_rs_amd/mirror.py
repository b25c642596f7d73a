"""Host-side mirror of the reference's operator interface for the hot path, on
top of the C ABI (same names, argument meaning and error behaviour):

    CRS, generate_crs                     src/generator.rs:35-42, 81-118 (generators supplied by the caller)
    Commit1 / Commit2 {coms, rand}        src/prover/commit.rs:18-28
    batch_commit_G1 / _G2 / _scalar_to_B1 / _scalar_to_B2, commit_G1 ...   commit.rs:59-256
    PPE / MSMEG1 / MSMEG2 / QuadEqu {a_consts, b_consts, gamma, target}     src/statement.rs:117-192
        .commit_and_prove(xvars, yvars, crs, rng) -> CProof                src/prover/prove.rs:29-52
        .prove(xvars, yvars, xcoms, ycoms, crs, rng) -> EquProof
        .verify(com_proof, crs) -> bool                                    src/verifier.rs:18-21
    EquProof {pi, theta, equ_type, rand}, CProof {xcoms, ycoms, equ_proofs} prove.rs:55-69

Values are numpy uint64 limb arrays in the boundary layout of include/gs_amd.h
(G1 = 2*NQ limbs, G2 = 4*NQ, Fr = 4, GT = 12*NQ); lists of them where the
reference has Vec<..>.  `rng` is any object with a method fr() returning one
Montgomery-form scalar (4 u64 limbs); draws happen in the reference's order
(R row-major, then S, then T: commit.rs:85-88,185-188; prove.rs:123-126).
Shape mismatches raise AssertionError where the reference panics via
assert_eq! (prove.rs:106-113; verifier.rs:25-26).

The batched entry points (`prove_many` / `verify_many`) are the build's
addition: N independent equations of one shape over the shared CRS -- the
reference has no batch API.
"""
import numpy as np

from .capi import GS_MSMEG1, GS_MSMEG2, GS_PPE, GS_QUAD, Engine


class CRS:
    """u: [Com1;2], v: [Com2;2], g1_gen, g2_gen, gt_gen  (generator.rs:35-42)."""

    def __init__(self, u, v, g1_gen, g2_gen, gt_gen, curve=0, device=0):
        self.u, self.v, self.g1_gen, self.g2_gen, self.gt_gen = u, v, g1_gen, g2_gen, gt_gen
        self.engine = Engine(curve, device)
        flat = np.concatenate([np.asarray(x, dtype=np.uint64).reshape(-1) for x in (u[0], u[1], v[0], v[1], g1_gen,
                                                                                   g2_gen, gt_gen)])
        self.engine.set_crs(flat)


def generate_crs(p1, p2, rng, curve=0, device=0, hiding=False):
    """AbstractCrs::generate_crs (generator.rs:81-118) with the group generators supplied by the caller
    (the reference draws them with G1::rand / G2::rand, which has no counterpart outside arkworks);
    the four scalars a1, a2, t1, t2 are drawn from `rng` in the reference's order (generator.rs:90-93)."""
    eng = Engine(curve, device)
    sc = np.concatenate([rng.fr() for _ in range(4)])
    raw = eng.crs_generate(p1, p2, sc, hiding=hiding).view(np.uint64)  # hiding: generator.rs:65-77
    eng.close()
    g1, g2 = eng.G1 // 8, eng.G2 // 8
    o = 0
    parts = []
    for sz in (2 * g1, 2 * g1, 2 * g2, 2 * g2, g1, g2, eng.GT // 8):
        parts.append(raw[o:o + sz].copy())
        o += sz
    return CRS([parts[0], parts[1]], [parts[2], parts[3]], parts[4], parts[5], parts[6], curve, device)


class Commit1:
    def __init__(self, coms, rand):
        self.coms, self.rand = coms, rand  # coms: list of Com1 (4*NQ limbs); rand: Matrix<Fr>

    def __eq__(self, o):
        return all((a == b).all() for a, b in zip(self.coms, o.coms)) and _mat_eq(self.rand, o.rand)

    def append(self, other):  # commit.rs:43-51
        assert len(self.coms) == len(self.rand) and len(other.coms) == len(other.rand)
        self.coms += other.coms
        self.rand += other.rand
        other.coms, other.rand = [], []


class Commit2(Commit1):
    pass


def _mat_eq(a, b):
    return len(a) == len(b) and all(len(x) == len(y) and all((p == q).all() for p, q in zip(x, y)) for x, y in zip(a, b))


def _cat(xs, width):
    if len(xs) == 0:
        return np.zeros(0, dtype=np.uint64)
    return np.concatenate([np.asarray(x, dtype=np.uint64).reshape(-1) for x in xs])


def _flat_mat(m):
    return _cat([e for row in m for e in row], 4)


def _split(buf, n):
    a = np.asarray(buf).view(np.uint64)
    return [a[i * (a.size // n):(i + 1) * (a.size // n)].copy() for i in range(n)] if n else []


def _batch_commit(kind, vars_, key, rng, cols):
    n = len(vars_)
    rand = [[rng.fr() for _ in range(cols)] for _ in range(n)]
    if n == 0:
        return [], rand
    out = key.engine.commit(kind, _cat(vars_, 0), _flat_mat(rand))
    return [out[i].view(np.uint64).copy() for i in range(n)], rand


def batch_commit_G1(xvars, key, rng):  # commit.rs:78-100
    return Commit1(*_batch_commit("g1", xvars, key, rng, 2))


def batch_commit_G2(yvars, key, rng):  # commit.rs:178-200
    return Commit2(*_batch_commit("g2", yvars, key, rng, 2))


def batch_commit_scalar_to_B1(scalar_xvars, key, rng):  # commit.rs:125-156
    return Commit1(*_batch_commit("fr_b1", scalar_xvars, key, rng, 1))


def batch_commit_scalar_to_B2(scalar_yvars, key, rng):  # commit.rs:225-256
    return Commit2(*_batch_commit("fr_b2", scalar_yvars, key, rng, 1))


def commit_G1(xvar, key, rng):  # commit.rs:59-75
    return batch_commit_G1([xvar], key, rng)


def commit_G2(yvar, key, rng):
    return batch_commit_G2([yvar], key, rng)


def commit_scalar_to_B1(x, key, rng):
    return batch_commit_scalar_to_B1([x], key, rng)


def commit_scalar_to_B2(y, key, rng):
    return batch_commit_scalar_to_B2([y], key, rng)


class EquProof:
    def __init__(self, pi, theta, equ_type, rand):
        self.pi, self.theta, self.equ_type, self.rand = pi, theta, equ_type, rand


class CProof:
    def __init__(self, xcoms, ycoms, equ_proofs):
        self.xcoms, self.ycoms, self.equ_proofs = xcoms, ycoms, equ_proofs


class _Equation:
    TYPE = None

    def __init__(self, a_consts, b_consts, gamma, target):
        self.a_consts, self.b_consts, self.gamma, self.target = a_consts, b_consts, gamma, target

    def get_type(self):
        return self.TYPE

    def _check_statement_shape(self, m, n):
        """a_consts pair with the n Y variables, b_consts with the m X variables, Gamma is m x n."""
        assert len(self.a_consts) == n, "a_consts.len() == yvars.len()"
        assert len(self.b_consts) == m, "b_consts.len() == xvars.len()"
        assert len(self.gamma) == m and all(len(row) == n for row in self.gamma), "gamma is m x n"

    # -- Provable ---------------------------------------------------------
    def _kxky(self):
        return (2 if self.TYPE in (GS_PPE, GS_MSMEG1) else 1), (2 if self.TYPE in (GS_PPE, GS_MSMEG2) else 1)

    def commit_and_prove(self, xvars, yvars, crs, rng):  # prove.rs:72-90 etc.
        kx, ky = self._kxky()
        xcoms = (batch_commit_G1 if kx == 2 else batch_commit_scalar_to_B1)(xvars, crs, rng)
        ycoms = (batch_commit_G2 if ky == 2 else batch_commit_scalar_to_B2)(yvars, crs, rng)
        return CProof(xcoms, ycoms, [self.prove(xvars, yvars, xcoms, ycoms, crs, rng)])

    def prove(self, xvars, yvars, xcoms, ycoms, crs, rng):  # prove.rs:92-171 etc.
        kx, ky = self._kxky()
        # the reference's shape asserts (prove.rs:106-113)
        assert len(xvars) == len(xcoms.rand)
        assert len(self.gamma) == len(xcoms.rand)
        assert len(xcoms.rand[0]) == kx
        assert len(yvars) == len(ycoms.rand)
        assert len(self.gamma[0]) == len(ycoms.rand)
        assert len(ycoms.rand[0]) == ky
        m, n = len(xvars), len(yvars)
        self._check_statement_shape(m, n)
        assert all(len(r) == kx for r in xcoms.rand) and all(len(r) == ky for r in ycoms.rand)
        T = [[rng.fr() for _ in range(kx)] for _ in range(ky)]
        out = crs.engine.prove_batch(self.TYPE, 1, m, n, _cat(xvars, 0), _cat(yvars, 0), _cat(self.a_consts, 0),
                                     _cat(self.b_consts, 0), _flat_mat(self.gamma), _flat_mat(xcoms.rand),
                                     _flat_mat(ycoms.rand), _flat_mat(T), want_coms=False)
        pi, theta = _split(out["pi"], kx), _split(out["theta"], ky)
        assert len(pi) == kx and len(theta) == ky
        return EquProof(pi, theta, self.TYPE, T)

    # -- Verifiable --------------------------------------------------------
    def verify(self, com_proof, crs):  # verifier.rs:23-157
        assert len(com_proof.equ_proofs) == 1
        assert self.get_type() == com_proof.equ_proofs[0].equ_type
        pf = com_proof.equ_proofs[0]
        m, n = len(com_proof.xcoms.coms), len(com_proof.ycoms.coms)
        # proof lengths come from the wire: check them where the reference panics (pairing_sum / left_mul,
        # data_structures.rs:495,705) instead of letting the C ABI read past a short buffer
        kx, ky = self._kxky()
        assert m >= 1 and n >= 1
        self._check_statement_shape(m, n)
        assert len(pf.pi) == kx and len(pf.theta) == ky
        ok = crs.engine.verify_batch(self.TYPE, 1, m, n, _cat(self.a_consts, 0), _cat(self.b_consts, 0),
                                     _flat_mat(self.gamma), np.asarray(self.target, dtype=np.uint64),
                                     _cat(com_proof.xcoms.coms, 0), _cat(com_proof.ycoms.coms, 0), _cat(pf.pi, 0),
                                     _cat(pf.theta, 0))
        return bool(ok[0])


class PPE(_Equation):
    TYPE = GS_PPE


class MSMEG1(_Equation):
    TYPE = GS_MSMEG1


class MSMEG2(_Equation):
    TYPE = GS_MSMEG2


class QuadEqu(_Equation):
    TYPE = GS_QUAD


class Statement:
    """`pub type Statement = Vec<dyn Equ>` (statement.rs:109) made usable: a list of equations of ONE type over the same
    variables (statement.rs:24-28: "each equation is defined with respect to the list of variables that span across
    ALL equations").  The variables are committed once; every equation gets its own EquProof against those
    commitments -- exactly what batch_commit_* followed by `equ.prove(..)` per equation does in the reference, in
    one call of the engine (gs_prove_statement / gs_verify_statement).  RNG draw order: R, S, then T of equation 0, 1, ..
    """

    def __init__(self, equations):
        assert len(equations) >= 1 and all(e.TYPE == equations[0].TYPE for e in equations)
        self.equations = list(equations)
        self.TYPE = equations[0].TYPE

    def _kxky(self):
        return self.equations[0]._kxky()

    def commit_and_prove(self, xvars, yvars, crs, rng):
        kx, ky = self._kxky()
        m, n, E = len(xvars), len(yvars), len(self.equations)
        assert m >= 1 and n >= 1
        for equ in self.equations:
            equ._check_statement_shape(m, n)
        R = [[rng.fr() for _ in range(kx)] for _ in range(m)]
        S = [[rng.fr() for _ in range(ky)] for _ in range(n)]
        Ts = [[[rng.fr() for _ in range(kx)] for _ in range(ky)] for _ in range(E)]
        cat = np.concatenate
        out = crs.engine.prove_statement(
            self.TYPE, E, m, n, _cat(xvars, 0), _cat(yvars, 0), cat([_cat(e.a_consts, 0) for e in self.equations]),
            cat([_cat(e.b_consts, 0) for e in self.equations]), cat([_flat_mat(e.gamma) for e in self.equations]),
            _flat_mat(R), _flat_mat(S), cat([_flat_mat(T) for T in Ts]))
        xcoms = Commit1(_split(out["xcoms"], m), R)
        ycoms = Commit2(_split(out["ycoms"], n), S)
        pis, ths = _split(out["pi"], E * kx), _split(out["theta"], E * ky)
        proofs = [EquProof(pis[e * kx:(e + 1) * kx], ths[e * ky:(e + 1) * ky], self.TYPE, Ts[e]) for e in range(E)]
        return CProof(xcoms, ycoms, proofs)

    def verify(self, com_proof, crs):
        """[bool per equation]; the statement holds iff all are true"""
        kx, ky = self._kxky()
        E = len(self.equations)
        assert len(com_proof.equ_proofs) == E
        m, n = len(com_proof.xcoms.coms), len(com_proof.ycoms.coms)
        assert m >= 1 and n >= 1
        for equ, pf in zip(self.equations, com_proof.equ_proofs):
            equ._check_statement_shape(m, n)
            assert pf.equ_type == self.TYPE and len(pf.pi) == kx and len(pf.theta) == ky
        cat = np.concatenate
        ok = crs.engine.verify_statement(
            self.TYPE, E, m, n, cat([_cat(e.a_consts, 0) for e in self.equations]),
            cat([_cat(e.b_consts, 0) for e in self.equations]), cat([_flat_mat(e.gamma) for e in self.equations]),
            cat([np.asarray(e.target, dtype=np.uint64).reshape(-1) for e in self.equations]),
            _cat(com_proof.xcoms.coms, 0), _cat(com_proof.ycoms.coms, 0),
            cat([_cat(pf.pi, 0) for pf in com_proof.equ_proofs]), cat([_cat(pf.theta, 0) for pf in com_proof.equ_proofs]))
        return [bool(v) for v in ok]


class MixedProof:
    """Commitments of the four variable groups of a mixed-type Statement and one EquProof per equation."""

    def __init__(self, com_xg, com_yg, com_xs, com_ys, equ_proofs):
        self.com_xg, self.com_yg, self.com_xs, self.com_ys, self.equ_proofs = com_xg, com_yg, com_xs, com_ys, equ_proofs


class MixedStatement:
    """`Statement = Vec<dyn Equ>` (statement.rs:24-28,109) with equations of ANY type over one list of variables:
    G1 variables xg, G2 variables yg and scalar variables xs (committed into B1), ys (into B2).  A PPE is over
    (xg, yg), an MSMEG1 over (xg, ys), an MSMEG2 over (xs, yg), a QuadEqu over (xs, ys)  (statement.rs:117-192: which
    side of each type is a group element).  Every group is committed ONCE (batch_commit_G1 / _G2 / _scalar_to_B1 /
    _scalar_to_B2), every equation gets its own EquProof against those commitments -- what the reference does when
    one calls `equ.prove(..)` per equation with shared Commit1 / Commit2 -- and all equations of all types go to the
    engine in ONE call (gs_prove_mixed / gs_verify_mixed with shared_vars parts).  RNG draw order: the four commit
    randomness matrices (xg, yg, xs, ys), then T of equation 0, 1, ..."""

    def __init__(self, equations):
        assert len(equations) >= 1
        self.equations = list(equations)

    @staticmethod
    def _groups(ty):
        return ("xg" if ty in (GS_PPE, GS_MSMEG1) else "xs"), ("yg" if ty in (GS_PPE, GS_MSMEG2) else "ys")

    def _by_type(self):
        order = []
        for e in self.equations:
            if e.TYPE not in order:
                order.append(e.TYPE)
        return [(ty, [i for i, e in enumerate(self.equations) if e.TYPE == ty]) for ty in order]

    def commit_and_prove(self, xg, yg, xs, ys, crs, rng):
        vars_ = dict(xg=xg, yg=yg, xs=xs, ys=ys)
        coms = dict(xg=batch_commit_G1(xg, crs, rng), yg=batch_commit_G2(yg, crs, rng),
                    xs=batch_commit_scalar_to_B1(xs, crs, rng), ys=batch_commit_scalar_to_B2(ys, crs, rng))
        Ts = []
        for e in self.equations:
            kx, ky = e._kxky()
            Ts.append([[rng.fr() for _ in range(kx)] for _ in range(ky)])
        cat = np.concatenate
        parts = []
        for ty, idx in self._by_type():
            gx, gy = self._groups(ty)
            m, n = len(vars_[gx]), len(vars_[gy])
            assert m >= 1 and n >= 1
            eqs = [self.equations[i] for i in idx]
            for e in eqs:
                e._check_statement_shape(m, n)
            parts.append(dict(ty=ty, N=len(idx), m=m, n=n, shared=True, want_coms=False, X=_cat(vars_[gx], 0),
                              Y=_cat(vars_[gy], 0), A=cat([_cat(e.a_consts, 0) for e in eqs]),
                              B=cat([_cat(e.b_consts, 0) for e in eqs]), Gamma=cat([_flat_mat(e.gamma) for e in eqs]),
                              R=_flat_mat(coms[gx].rand), S=_flat_mat(coms[gy].rand),
                              T=cat([_flat_mat(Ts[i]) for i in idx])))
        outs = crs.engine.prove_mixed(parts)
        proofs = [None] * len(self.equations)
        for (ty, idx), o in zip(self._by_type(), outs):
            kx, ky = self.equations[idx[0]]._kxky()
            pis, ths = _split(o["pi"], len(idx) * kx), _split(o["theta"], len(idx) * ky)
            for k, i in enumerate(idx):
                proofs[i] = EquProof(pis[k * kx:(k + 1) * kx], ths[k * ky:(k + 1) * ky], ty, Ts[i])
        return MixedProof(coms["xg"], coms["yg"], coms["xs"], coms["ys"], proofs)

    def verify(self, proof, crs):
        """[bool per equation], in the order of the statement's equations"""
        coms = dict(xg=proof.com_xg, yg=proof.com_yg, xs=proof.com_xs, ys=proof.com_ys)
        assert len(proof.equ_proofs) == len(self.equations)
        cat = np.concatenate
        parts = []
        for ty, idx in self._by_type():
            gx, gy = self._groups(ty)
            m, n = len(coms[gx].coms), len(coms[gy].coms)
            assert m >= 1 and n >= 1
            eqs = [self.equations[i] for i in idx]
            pfs = [proof.equ_proofs[i] for i in idx]
            kx, ky = eqs[0]._kxky()
            for e, pf in zip(eqs, pfs):
                e._check_statement_shape(m, n)
                assert pf.equ_type == ty and len(pf.pi) == kx and len(pf.theta) == ky
            parts.append(dict(ty=ty, N=len(idx), m=m, n=n, shared=True, A=cat([_cat(e.a_consts, 0) for e in eqs]),
                              B=cat([_cat(e.b_consts, 0) for e in eqs]), Gamma=cat([_flat_mat(e.gamma) for e in eqs]),
                              target=cat([np.asarray(e.target, dtype=np.uint64).reshape(-1) for e in eqs]),
                              xcoms=_cat(coms[gx].coms, 0), ycoms=_cat(coms[gy].coms, 0),
                              pi=cat([_cat(pf.pi, 0) for pf in pfs]), theta=cat([_cat(pf.theta, 0) for pf in pfs])))
        oks = crs.engine.verify_mixed(parts)
        out = [None] * len(self.equations)
        for (ty, idx), ok in zip(self._by_type(), oks):
            for k, i in enumerate(idx):
                out[i] = bool(ok[k])
        return out
