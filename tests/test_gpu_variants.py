"""Every kernel SHAPE the planner can pick is compared with the C oracle.

The planner (csrc/gs_amd.hip: miller_cost, pick_tm, coop_fe) switches kernels with the batch size: twin-accumulator
Miller lanes, 1..12 pairs per lane, Straus groups of 4 / 8 terms (one to four outputs per table build, 4- or 5-bit windows), one, two, four or eight outputs per inversion in the reductions, one-lane or 3-lane final exponentiation, table-reading
or stepping CRS pairs, side streams.  At the small N the oracle can follow, the planner alone would only ever choose
the small-batch shapes; here every shape is FORCED through gs_set_option, all four equation types, both curves, every
equation of the batch bit-exact (commitments, pi, theta) and verdict-exact against oracle/gs_ref.c, and the library's
kernel profile proves that the intended kernel is the one that ran.  Reference: src/prover/prove.rs:92-171,
src/verifier.rs:23-55."""
import numpy as np
import pytest

from gpubatch import run_batch

pytestmark = pytest.mark.gpu

# name -> options; chosen so that every value of every knob occurs with both values of its neighbours
SHAPES = {
    "twin6_straus8x2w5_lane": dict(red_k=4, miller_twin=1, miller_ch=6, var_tm=8, var_mo=2, var_w=5, coop_fe=0, line_tables=1,
                                   overlap=0),
    "twin2_straus4x4_coop_notab": dict(red_k=2, miller_twin=1, miller_ch=2, var_tm=4, var_mo=4, var_w=4, coop_fe=2,
                                       line_tables=0, overlap=0),
    "twin4_straus2_lane_overlap": dict(red_k=1, miller_twin=1, miller_ch=4, var_tm=2, var_mo=1, var_w=4, coop_fe=0,
                                       line_tables=1, overlap=1),
    "single6_straus8x4w5_lane": dict(red_k=2, miller_twin=0, miller_ch=6, var_tm=8, var_mo=4, var_w=5, coop_fe=0, line_tables=1,
                                     overlap=0),
    "single1_plain_coop_notab_overlap": dict(red_k=4, miller_twin=0, miller_ch=1, var_tm=1, var_mo=1, var_w=4, coop_fe=2,
                                             line_tables=0, overlap=1),
    "single9_straus4w5_coop": dict(red_k=1, miller_twin=0, miller_ch=9, var_tm=4, var_mo=1, var_w=5, coop_fe=2, line_tables=1,
                                   overlap=0),
    "twin12_straus8x2_lane": dict(red_k=4, miller_twin=1, miller_ch=12, var_tm=8, var_mo=2, var_w=4, coop_fe=0, line_tables=1,
                                 overlap=0),
    # the twin task list on PAIRS of lanes, one accumulator each, lines exchanged through LDS (2) or DPP (3):
    # k_miller_pair; 12 triples per task (6 twist points per lane), 5 (odd: lane 1 idles in the last round, no tables)
    # and 3 with tables (one stepping point against two)
    "pair12_straus8x2w5_lane": dict(red_k=4, miller_twin=2, miller_ch=12, var_tm=8, var_mo=2, var_w=5, coop_fe=0,
                                   line_tables=1, overlap=0),
    "pair5dpp_straus4_coop_notab": dict(red_k=2, miller_twin=3, miller_ch=5, var_tm=4, var_mo=1, var_w=4, coop_fe=2,
                                       line_tables=0, overlap=0),
    "pair3_straus2_lane_overlap": dict(red_k=1, miller_twin=2, miller_ch=3, var_tm=2, var_mo=1, var_w=4, coop_fe=0,
                                       line_tables=1, overlap=1),
    # the G1 Straus lanes in the kernels BUILT FOR TWO WAVES PER SIMD (round 4: register-only G1 point operations leave
    # room for it; planned for launches that put two waves on every SIMD, forced here on 66 equations), 4- and 8-term
    # groups, 4- and 5-bit windows
    "pair12_straus8x2w5_lane_w2": dict(red_k=4, miller_twin=2, miller_ch=12, var_tm=8, var_mo=2, var_w=5, coop_fe=0,
                                      line_tables=1, overlap=0, var_w2=1),
    "pair5dpp_straus4x4_coop_w2": dict(red_k=2, miller_twin=3, miller_ch=5, var_tm=4, var_mo=4, var_w=4, coop_fe=2,
                                      line_tables=1, overlap=0, var_w2=1),
    # the verifier's Gamma^T c on window tables of the commitment components shared by all outputs (what large arities
    # use; forced here at 4 x 4): k_tab_build + k_var_tab8
    # (red_k=8: every output of a side behind one inversion)
    "pair12_tab8_lane": dict(red_k=8, miller_twin=3, miller_ch=12, var_tm=8, var_mo=2, var_w=5, coop_fe=0, line_tables=1,
                             overlap=0, var_tab=1),
}
MILLER_KERNEL = {0: "k_miller", 1: "k_miller.twin", 2: "k_miller.pair", 3: "k_miller.pairdpp"}


def expected_kernels(ty, m, n, o):
    xg, yg = ty in (0, 1), ty in (0, 2)
    ex = [MILLER_KERNEL[o["miller_twin"]], "k_final.coop" if o["coop_fe"] == 2 else "k_final"]

    def var_name(terms, tag):  # what pick_tm's balanced group size selects (csrc/gs_amd.hip)
        tm = o["var_tm"]
        if terms < 2 or tm <= 1:
            return "k_var" + tag
        ng = (terms + tm - 1) // tm
        eff = (terms + ng - 1) // ng
        if eff <= 1:
            return "k_var" + tag
        # k_var_multi<group capacity>[w5][x<outputs per lane>]: the lanes share one table build between outputs
        return (("k_var_multi4" if eff <= 4 else "k_var_multi8") + ("w5" if o["var_w"] == 5 else "") +
                ("x%d" % o["var_mo"] if o["var_mo"] > 1 else "") + tag)

    if xg:
        ex.append(var_name(m + n, ".g1"))
    if yg:
        ex.append(var_name(m + n, ".g2"))
    if o.get("var_tab") == 1:
        ex += ["k_tab_build.vg1", "k_var_tab8.vg1"]
    else:
        ex.append(var_name(m, ".vg1"))
    return ex


@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("ty", [0, 1, 2, 3])
@pytest.mark.parametrize("cname,cid", [("bls12_381", 0), ("bn254", 1)])
def test_forced_kernel_shapes_match_oracle(cname, cid, ty, shape):
    o = SHAPES[shape]
    N, m, n = 66, 4, 4  # two waves per task, the second one ragged
    run_batch(cid, cname, ty, N, m, n, range(N), opts=o, expect=expected_kernels(ty, m, n, o), seed=9100 + ty)


def test_straus_workspace_smaller_than_the_batch():
    """The Straus lanes' table workspace is bounded (var_ws_lanes); a batch with more lanes runs as several launches
    over the same workspace: 66 equations x 2 groups in chunks of 64 lanes, ragged last chunk."""
    o = dict(SHAPES["twin6_straus8x2w5_lane"], var_ws_lanes=64)
    run_batch(0, "bls12_381", 0, 66, 4, 4, range(66), opts=o, expect=expected_kernels(0, 4, 4, o), seed=9300)


@pytest.mark.parametrize("cname,cid", [("bls12_381", 0), ("bn254", 1)])
def test_forced_shapes_wide_statement(cname, cid):
    """8 x 3 PPE: 8-term Straus groups on the verifier's G1 side too (k_var_multi8.vg1), 11 pairs in the b = 1 cells
    (two twin lanes of 6 + 5), uneven Miller lanes."""
    o = SHAPES["twin6_straus8x2w5_lane"]
    run_batch(cid, cname, 0, 40, 8, 3, range(40), opts=o, expect=expected_kernels(0, 8, 3, o) + ["k_var_multi8w5x2.vg1"],
              seed=9200)


def test_option_values_are_validated():
    """gs_set_option rejects unknown keys and out-of-range values (GS_ERR_ARG) and leaves the context usable."""
    import groth_sahai_rs_amd as gs

    eng = gs.Engine(0, 0)
    try:
        for key, bad in (("no_such_option", 1), ("miller_ch", 13), ("miller_twin", 4), ("var_tm", 9), ("var_mo", 3),
                         ("var_w", 6), ("red_k", 3), ("var_ws_lanes", 100), ("coop_fe", 3), ("var_tab", 2), ("mixed_merge", 2),
                         ("var_w2", 2)):
            with pytest.raises(gs.GsError):
                eng.set_option(key, bad)
        for key, good in (("miller_ch", 12), ("var_mo", 4), ("var_w", 5), ("red_k", 4), ("var_ws_lanes", 128),
                          ("miller_ch", 0), ("var_mo", 0), ("var_w", 0), ("red_k", 0), ("var_ws_lanes", 0), ("var_w2", 1),
                          ("var_w2", -1), ("endo", 0), ("endo", 1)):
            eng.set_option(key, good)
    finally:
        eng.close()


@pytest.mark.parametrize("cname,cid", [("bls12_381", 0), ("bn254", 1)])
@pytest.mark.parametrize("shape", ["pair12_straus8x2w5_lane", "twin2_straus4x4_coop_notab", "single1_plain_coop_notab_overlap"])
def test_identical_terms_in_one_lane_take_the_doubling_path(cname, cid, shape):
    """P = Q inside a generated addition.  The G1 / G2 point-operation subroutines (gen_pointops_asm.py) are the generic
    branch of the formulas; H = 0 (P = +-Q) is found by the 56-bit filter -- for G2 INSIDE the subroutine, which then
    returns at once with every operand untouched and sends the whole wave through the C++ addition.  Random batches never
    get there.  Here every equation carries two identical variables on both sides (X1 = X0, Y1 = Y0) and a Gamma of
    ones, so that the Straus lanes of the prover (two identical (base, scalar) terms in one lane: the second addition of
    the top window adds a point to itself) and, with one term per lane, the reductions meet P = Q; proofs and verdicts
    against the C oracle, all four types."""
    import torch

    import groth_sahai_rs_amd as gs
    import gs_ref_py as ref
    from groth_sahai_rs_amd.workload import CURVES, Workload

    o = SHAPES[shape]
    for ty in (0, 1, 2, 3):
        eng = gs.Engine(cid, 0)
        for k, v in o.items():
            eng.set_option(k, v)
        m, n, N = 4, 4, 66
        wl = Workload(eng, ty=ty, N=N, m=m, n=n, seed=777 + ty, corrupt_every=0)
        sh = wl.sh
        kx, ky, sx, sy, st = sh["kx"], sh["ky"], sh["sx"], sh["sy"], sh["st"]
        r = CURVES[cid]["r"]
        one = torch.from_numpy(np.array([((1 << 256) % r >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)],
                                        dtype=np.uint64).view(np.uint8).copy()).to(wl.X.device)
        X, Y, G = wl.X.view(N, m, sx), wl.Y.view(N, n, sy), wl.Gamma.view(N, m * n, 32)
        X[:, 1] = X[:, 0]
        X[:, 3] = X[:, 2]
        Y[:, 1] = Y[:, 0]
        Y[:, 3] = Y[:, 2]
        G[:, :] = one
        wl.prove()
        wl.verify()
        eng.sync()
        host = lambda t: t.cpu().numpy()
        Xh, Yh, A, B, Gh, R, S, T = map(host, (wl.X, wl.Y, wl.A, wl.B, wl.Gamma, wl.R, wl.S, wl.T))
        xc, yc, pi, th, tgt, ok = map(host, (wl.xcoms, wl.ycoms, wl.pi, wl.theta, wl.target, wl.ok))
        cut = lambda a, e, sz: a[e * sz:(e + 1) * sz]
        for e in (0, 1, 31, 64, 65):
            out = ref.commit_and_prove(cname, ty, m, n, cut(Xh, e, m * sx), cut(Yh, e, n * sy), cut(A, e, n * sx),
                                       cut(B, e, m * sy), cut(Gh, e, m * n * 32), cut(R, e, m * kx * 32),
                                       cut(S, e, n * ky * 32), cut(T, e, ky * kx * 32), wl.crs)
            for name, got, per in (("xcoms", xc, m * eng.COM1), ("ycoms", yc, n * eng.COM2), ("pi", pi, kx * eng.COM2),
                                   ("theta", th, ky * eng.COM1)):
                assert (out[name] == cut(got, e, per)).all(), (cname, shape, ty, e, name)
            want = ref.verify(cname, ty, m, n, cut(A, e, n * sx), cut(B, e, m * sy), cut(Gh, e, m * n * 32),
                              cut(tgt, e, st), out["xcoms"], out["ycoms"], out["pi"], out["theta"], wl.crs)
            assert int(ok[e]) == want, (cname, shape, ty, e, "verdict")
        eng.close()
