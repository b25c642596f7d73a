"""Several sub-batches in ONE call (gs_prove_mixed / gs_verify_mixed, include/gs_amd.h).

  * configs[2] of the baseline: a batch that mixes PPE, MSMEG1 and MSMEG2 equations (here also of different shapes)
    proved and verified in one call -- the parts' launches recorded and MERGED into segmented launches (k_seg: one
    launch per kernel body with a segment per part) -- gives byte-for-byte what one gs_prove_batch / gs_verify_batch
    call per sub-batch gives, through host pointers and through device pointers, merged and (mixed_merge = 0) with the
    parts one after the other, and with the library's kernel profile on (it times the MERGED launches themselves:
    round 4 -- the per-kernel times of a mixed step then add up to the step bench.py times);
  * a mixed-type STATEMENT (statement.rs:24-28,109: equations of any type over ONE list of variables): the variables are
    committed once per group, 70 equations of the four types (two waves of PPEs) get their proofs in one call, and EVERY
    equation's pi / theta and the shared commitments equal the C oracle's commit_and_prove with the same X, Y, R, S and
    that equation's A, B, Gamma, T (oracle/gs_ref.c), every verdict equals the oracle's verdict, equations whose target
    was moved are rejected by both.  Both curves."""
import os
import sys

import numpy as np
import pytest

from gsutil import REPO

sys.path.insert(0, os.path.join(REPO, "oracle"))

pytestmark = pytest.mark.gpu

IN_P = ("X", "Y", "A", "B", "Gamma", "R", "S", "T")
OUT_P = ("xcoms", "ycoms", "pi", "theta")
IN_V = ("A", "B", "Gamma", "target", "xcoms", "ycoms", "pi", "theta")


def test_mixed_batch_equals_one_call_per_sub_batch():
    import torch

    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd.workload import Workload

    eng = gs.Engine(0, 0)
    shapes = [(0, 40, 4, 4), (1, 23, 3, 2), (2, 70, 2, 5), (3, 9, 1, 1)]
    wls = [Workload(eng, ty=ty, N=N, m=m, n=n, seed=6000, corrupt_every=0) for ty, N, m, n in shapes]  # one CRS
    host = lambda t: t.cpu().numpy()
    want, hostin = [], []
    for wl in wls:
        wl.prove()
        eng.sync()
        want.append({k: host(getattr(wl, k)) for k in OUT_P})
        hostin.append({k: host(getattr(wl, k)) for k in IN_P + ("target",)})
    parts = [dict(ty=ty, N=N, m=m, n=n, **{k: h[k] for k in IN_P}) for (ty, N, m, n), h in zip(shapes, hostin)]
    for prof, merge in ((False, 1), (True, 1), (False, 0)):  # merged launches, the same under the profile, in sequence
        eng.prof_enable(prof)
        eng.prof_reset()
        eng.set_option("mixed_merge", merge)
        got = eng.prove_mixed(parts)
        if prof:  # the profile saw the merged launches: ONE k_prep_prove launch for the four parts, not four
            launches = {n: k for n, _, k in eng.prof_get()}
            assert launches.get("k_prep_prove") == 1, launches
        for g, w in zip(got, want):
            for k in OUT_P:
                assert (g[k] == w[k]).all(), (k, prof)
        vparts = [dict(ty=ty, N=N, m=m, n=n, A=h["A"], B=h["B"], Gamma=h["Gamma"], target=h["target"], **g)
                  for (ty, N, m, n), h, g in zip(shapes, hostin, got)]
        oks = eng.verify_mixed(vparts)
        assert all(ok.all() for ok in oks)
        vparts[2]["pi"] = vparts[2]["pi"].copy()
        vparts[2]["pi"][69 * (len(vparts[2]["pi"]) // 70) + 4] ^= 1
        oks = eng.verify_mixed(vparts)
        assert oks[0].all() and oks[1].all() and oks[3].all() and oks[2][:69].all() and oks[2][69] == 0
    eng.prof_enable(False)
    eng.set_option("mixed_merge", -1)
    # six parts in one call: more parts than a merged launch has segments (4), so every body goes out in two launches
    got6 = eng.prove_mixed((parts * 2)[:6])
    for g, w in zip(got6, (want * 2)[:6]):
        for k in OUT_P:
            assert (g[k] == w[k]).all(), ("six parts", k)
    # device pointers: the workloads' own tensors, outputs zeroed first
    for wl in wls:
        for k in OUT_P:
            getattr(wl, k).zero_()
        wl.ok.zero_()
    dparts = [dict(ty=wl.ty, N=wl.N, m=wl.m, n=wl.n, X=wl.X, Y=wl.Y, A=wl.A, B=wl.B, Gamma=wl.Gamma, R=wl.R, S=wl.S,
                   T=wl.T, xcoms=wl.xcoms, ycoms=wl.ycoms, pi=wl.pi, theta=wl.theta) for wl in wls]
    eng.prove_mixed_dev(dparts)
    eng.verify_mixed_dev([dict(ty=wl.ty, N=wl.N, m=wl.m, n=wl.n, A=wl.A, B=wl.B, Gamma=wl.Gamma, target=wl.target,
                               xcoms=wl.xcoms, ycoms=wl.ycoms, pi=wl.pi, theta=wl.theta, ok=wl.ok) for wl in wls])
    eng.sync()
    torch.cuda.synchronize()
    for wl, w in zip(wls, want):
        for k in OUT_P:
            assert (host(getattr(wl, k)) == w[k]).all(), k
        assert host(wl.ok).all()
    with pytest.raises(gs.GsError):
        eng.prove_mixed(parts * 3)  # more than GS_MIXED_MAX parts
    eng.close()


@pytest.mark.parametrize("cname,cid", [("bls12_381", 0), ("bn254", 1)])
def test_mixed_type_statement_against_the_oracle(cname, cid):
    import groth_sahai_rs_amd as gs
    import gs_ref_py as ref
    from gpubatch import pool
    from stmtutil import StatementInputs

    eng = gs.Engine(cid, 0)
    si = StatementInputs(eng, mg=3, ng=2, ms=2, ns=3, seed=5150 + cid)
    counts = {0: 66, 1: 2, 2: 1, 3: 1}  # 70 equations; the PPEs alone are two waves of Miller lanes
    parts = [si.part(ty, E) for ty, E in counts.items()]
    # commitments once per variable group (commit.rs:78-100,125-156,178-200,225-256)
    com = dict(xg=eng.commit("g1", si.vars["xg"], si.rand["xg"]), yg=eng.commit("g2", si.vars["yg"], si.rand["yg"]),
               xs=eng.commit("fr_b1", si.vars["xs"], si.rand["xs"]), ys=eng.commit("fr_b2", si.vars["ys"], si.rand["ys"]))
    outs = eng.prove_mixed([dict(p, want_coms=False) for p in parts])
    per = lambda a, N: len(a) // N

    def check(job):
        p, o, e = job
        ty, N, m, n = p["ty"], p["N"], p["m"], p["n"]
        gx, gy = si.groups(ty)
        cut = lambda a: a[e * per(a, N):(e + 1) * per(a, N)]
        want = ref.commit_and_prove(cname, ty, m, n, p["X"], p["Y"], cut(p["A"]), cut(p["B"]), cut(p["Gamma"]), p["R"],
                                    p["S"], cut(p["T"]), si.crs)
        if not ((want["xcoms"] == com[gx].reshape(-1)).all() and (want["ycoms"] == com[gy].reshape(-1)).all()):
            return (ty, e, "shared commitments")
        if not ((want["pi"] == cut(o["pi"])).all() and (want["theta"] == cut(o["theta"])).all()):
            return (ty, e, "proof")
        v = ref.verify(cname, ty, m, n, cut(p["A"]), cut(p["B"]), cut(p["Gamma"]), cut(p["target"]), want["xcoms"],
                       want["ycoms"], want["pi"], want["theta"], si.crs)
        return None if v == 1 else (ty, e, "oracle verdict")

    jobs = [(p, o, e) for p, o in zip(parts, outs) for e in range(p["N"])]
    bad = [r for r in pool().map(check, jobs) if r is not None]
    assert not bad, bad[:4]
    vparts = [dict(ty=p["ty"], N=p["N"], m=p["m"], n=p["n"], shared=True, A=p["A"], B=p["B"], Gamma=p["Gamma"],
                   target=p["target"], xcoms=com[si.groups(p["ty"])[0]].reshape(-1),
                   ycoms=com[si.groups(p["ty"])[1]].reshape(-1), pi=o["pi"], theta=o["theta"])
              for p, o in zip(parts, outs)]
    oks = eng.verify_mixed(vparts)
    assert all(ok.all() for ok in oks)
    # equations that do NOT hold (targets moved), proved honestly: rejected here and by the oracle
    wrong = [si.part(ty, 2, satisfied=False) for ty in (0, 1, 2, 3)]
    wouts = eng.prove_mixed([dict(p, want_coms=False) for p in wrong])
    wv = [dict(ty=p["ty"], N=2, m=p["m"], n=p["n"], shared=True, A=p["A"], B=p["B"], Gamma=p["Gamma"], target=p["target"],
               xcoms=com[si.groups(p["ty"])[0]].reshape(-1), ycoms=com[si.groups(p["ty"])[1]].reshape(-1), pi=o["pi"],
               theta=o["theta"]) for p, o in zip(wrong, wouts)]
    assert not any(ok.any() for ok in eng.verify_mixed(wv))
    p, o = wrong[0], wouts[0]
    cut = lambda a: a[:per(a, 2)]
    assert ref.verify(cname, 0, p["m"], p["n"], cut(p["A"]), cut(p["B"]), cut(p["Gamma"]), cut(p["target"]),
                      com["xg"].reshape(-1), com["yg"].reshape(-1), cut(o["pi"]), cut(o["theta"]), si.crs) == 0
    eng.close()


def test_mixed_statement_mirror_draw_order_and_verdicts():
    """mirror.MixedStatement: commitments drawn in the order xg, yg, xs, ys, then T per equation; the proofs equal
    Provable::prove per equation against the shared commitments (mirror's single-equation path on the same draws)."""
    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd import mirror
    from gsutil import curve
    from stmtutil import StatementInputs

    c = curve("bls12_381")
    g = c.golden["crs"]
    crs = mirror.CRS([c.com1(g["u"][0]), c.com1(g["u"][1])], [c.com2(g["v"][0]), c.com2(g["v"][1])], c.g1(g["g1"]),
                     c.g2(g["g2"]), c.f12(g["gt"]))
    cases = [c.golden["cases"][i] for i in (4, 5, 6, 7)]  # dense 2 x 2 of every type
    lim = lambda mat: [[c.fr_hex(s) for s in row] for row in mat]
    ex = lambda ty: (c.g1 if ty in (0, 1) else c.fr_hex)
    ey = lambda ty: (c.g2 if ty in (0, 2) else c.fr_hex)
    ppe, m1, m2, qd = cases
    xg, yg = [c.g1(v) for v in ppe["xvars"]], [c.g2(v) for v in ppe["yvars"]]
    xs, ys = [c.fr_hex(v) for v in qd["xvars"]], [c.fr_hex(v) for v in qd["yvars"]]
    cls = [mirror.PPE, mirror.MSMEG1, mirror.MSMEG2, mirror.QuadEqu]
    tgt = {0: c.f12, 1: c.g1, 2: c.g2, 3: c.fr_hex}
    equs = [cls[k["type"]]([ex(k["type"])(v) for v in k["a"]], [ey(k["type"])(v) for v in k["b"]], lim(k["gamma"]),
                           tgt[k["type"]](k["target"])) for k in (ppe, qd, m1, m2, ppe)]

    class Rng:
        def __init__(self, seed):
            self.r = np.random.default_rng(seed)

        def fr(self):
            v = self.r.integers(0, 1 << 62, size=4, dtype=np.uint64)
            v[3] &= np.uint64((1 << 60) - 1)  # < r as a Montgomery representative
            return v

    st = mirror.MixedStatement(equs)
    proof = st.commit_and_prove(xg, yg, xs, ys, crs, Rng(3))
    rng = Rng(3)
    cx = mirror.batch_commit_G1(xg, crs, rng)
    cy = mirror.batch_commit_G2(yg, crs, rng)
    csx = mirror.batch_commit_scalar_to_B1(xs, crs, rng)
    csy = mirror.batch_commit_scalar_to_B2(ys, crs, rng)
    assert proof.com_xg == cx and proof.com_yg == cy and proof.com_xs == csx and proof.com_ys == csy
    grp = {0: (xg, yg, cx, cy), 1: (xg, ys, cx, csy), 2: (xs, yg, csx, cy), 3: (xs, ys, csx, csy)}
    for e, pf in zip(equs, proof.equ_proofs):
        xv, yv, xc, yc = grp[e.TYPE]
        want = e.prove(xv, yv, xc, yc, crs, rng)
        assert all((a == b).all() for a, b in zip(pf.pi, want.pi)) and all((a == b).all() for a, b in zip(pf.theta, want.theta))
    verdicts = st.verify(proof, crs)
    each = [e.verify(mirror.CProof(grp[e.TYPE][2], grp[e.TYPE][3], [pf]), crs) for e, pf in zip(equs, proof.equ_proofs)]
    assert verdicts == each
