"""BASELINE.json's full-size configurations, at the size that selects their kernels:

  configs[2]  2^16 mixed PPE + MSME equations, BLS12-381 (50 % PPE, 25 % MSMEG1, 25 % MSMEG2 -- SURVEY.md 8d)
  configs[3]  the 2^15-equation shard one GPU of eight owns of the 2^18 batch
  configs[4]  2^16 PPE, BN254

Each batch is proved with the planner left alone (so the large-batch shapes run: twin Miller lanes, 8-term Straus
groups, one lane per final exponentiation), >= 16 sampled equations -- first / last lanes of the first / last waves and
a spread in between -- are compared bit for bit with the C oracle, and the whole batch goes through the
size-independent properties (all honest proofs accepted, exactly the corrupted ones rejected, batched verdict agrees).
Round 4 adds the shapes that are actually TIMED: the driver's default line (2^16 PPE BLS12-381: the
LDS lane-pair Miller kernel with the two-task plan) at its own size, and the mixed entry points (gs_prove_mixed_dev /
gs_verify_mixed_dev) at 2^12 and 2^14 with merged launches -- where the whole call's size picks every part's lane
shapes -- and at 2^16, where the parts run in sequence.
Reference: src/prover/prove.rs:92-171, src/verifier.rs:23-55, src/statement.rs:24-28,109."""
import fnmatch

import numpy as np
import pytest

from gpubatch import oracle_check, run_batch

pytestmark = pytest.mark.gpu


def spread(N, k=16):
    s = {0, 1, 63, 64, N // 2 - 1, N // 2, N - 65, N - 64, N - 2, N - 1}
    step = max(N // (k - len(s) + 1), 1)
    s.update(range(step // 2, N, step))
    return sorted(s)


# (k_var_multi*: the planner picks the group size together with the window width and the outputs per table build)
# (k_miller.pair*: two lanes per (equation, task) with one accumulator each -- what the planner picks at these sizes)
LARGE = ["k_miller.pair*", "k_final", "k_var_multi*.g1", "k_var_multi*.g2"]


def test_config2_mixed_2p16_bls12_381():
    N = 1 << 16
    run_batch(0, "bls12_381", 0, N // 2, 4, 4, spread(N // 2), seed=20241222, corrupt_every=1024, expect=LARGE)
    run_batch(0, "bls12_381", 1, N // 4, 4, 4, spread(N // 4), seed=20241222, corrupt_every=1024)
    run_batch(0, "bls12_381", 2, N // 4, 4, 4, spread(N // 4), seed=20241222, corrupt_every=1024)


def test_config3_shard_2p15_bls12_381():
    N = 1 << 15
    run_batch(0, "bls12_381", 0, N, 4, 4, spread(N), seed=20241223, corrupt_every=1024, expect=LARGE)


def test_config4_bn254_2p16():
    N = 1 << 16
    run_batch(1, "bn254", 0, N, 4, 4, spread(N), seed=20241224, corrupt_every=1024, expect=LARGE)


def test_config1_2p12_sampled_against_oracle():
    """configs[1] at its own size (the planner's mid-size shapes), 24 equations against the oracle."""
    N = 1 << 12
    run_batch(0, "bls12_381", 0, N, 4, 4, spread(N, 24), seed=20241221, corrupt_every=1024)


def test_headline_2p16_ppe_bls12_381():
    """The workload bench.py's default line times (BASELINE.json north_star: 2^16 PPE 4x4, BLS12-381, one GPU), planner
    left alone: `k_miller.pair` is the LDS-exchange lane-pair kernel (the DPP form is `k_miller.pairdpp`)."""
    N = 1 << 16
    run_batch(0, "bls12_381", 0, N, 4, 4, spread(N), seed=20241221, corrupt_every=1024,
              expect=["k_miller.pair", "k_final", "k_var_multi*.g1", "k_var_multi*.g2", "k_var_multi*.vg1", "k_fix.g1",
                      "k_fix.g2", "k_red.g1", "k_red.g2"])


def run_mixed(log2n, merge, seed, samples=16):
    """50 % PPE / 25 % MSMEG1 / 25 % MSMEG2 (BASELINE configs[2]) through ONE gs_prove_mixed_dev and ONE
    gs_verify_mixed_dev call: sampled equations of every part against the C oracle, whole-call verdicts, corrupted
    proofs found in every part; the kernel profile shows how the launches went out."""
    import groth_sahai_rs_amd as gs
    import gs_ref_py as ref
    from groth_sahai_rs_amd.workload import Workload

    eng = gs.Engine(0, 0)
    eng.set_option("mixed_merge", merge)
    N = 1 << log2n
    wls = [Workload(eng, ty=ty, N=n, m=4, n=4, seed=seed, corrupt_every=max(n // 8, 1))
           for ty, n in ((0, N // 2), (1, N // 4), (2, N // 4))]
    assert all((w.crs == wls[0].crs).all() for w in wls)
    pparts = [dict(ty=w.ty, N=w.N, m=w.m, n=w.n, X=w.X, Y=w.Y, A=w.A, B=w.B, Gamma=w.Gamma, R=w.R, S=w.S, T=w.T,
                   xcoms=w.xcoms, ycoms=w.ycoms, pi=w.pi, theta=w.theta) for w in wls]
    vparts = [dict(ty=w.ty, N=w.N, m=w.m, n=w.n, A=w.A, B=w.B, Gamma=w.Gamma, target=w.target, xcoms=w.xcoms,
                   ycoms=w.ycoms, pi=w.pi, theta=w.theta, ok=w.ok) for w in wls]
    eng.prove_mixed_dev(pparts)
    eng.sync()
    for w in wls:  # every part: >= `samples` equations, first / last lanes of first / last waves and a spread
        oracle_check(ref, "bls12_381", eng, w, spread(w.N, samples))
    eng.verify_mixed_dev(vparts)
    eng.sync()
    assert all(w.ok.cpu().numpy().all() for w in wls)
    # the same call under the kernel profile: identical bytes, and the launch counts show merged / sequential
    keep = [{k: getattr(w, k).clone() for k in ("xcoms", "ycoms", "pi", "theta")} for w in wls]
    eng.prof_enable(True)
    eng.prof_reset()
    eng.prove_mixed_dev(pparts)
    eng.verify_mixed_dev(vparts)
    eng.sync()
    prof = eng.prof_get()
    work = eng.prof_get_work()
    eng.prof_enable(False)
    for w, k0 in zip(wls, keep):
        for k, v in k0.items():
            assert (getattr(w, k) == v).all(), k
    launches = {}
    for name, _, cnt in prof:
        fam = name.split(".")[0]
        fam = "k_var_multi" if fam.startswith("k_var_multi") else fam
        launches[fam] = launches.get(fam, 0) + cnt
    names = [p[0] for p in prof]
    assert any(fnmatch.fnmatchcase(nm, "k_miller.pair*") for nm in names), names
    merged = merge == 1 or (merge == -1 and N <= (1 << 14))  # (planned: merged up to 2^14 equations per call)
    if merged:
        # three parts, one launch per body: the Miller lanes of all parts in ONE launch, one final-exponentiation
        # launch, one scalar-preparation launch per direction
        assert launches["k_miller"] == 1 and launches["k_prep_prove"] == 1 and launches["k_prep_verify"] == 1, launches
        assert launches.get("k_final", 0) == 1, launches  # (k_final and k_final.coop share the family name)
        mil = [nm for nm in names if nm.startswith("k_miller")][0]
        # lanes of the merged Miller launch = all parts' lanes: more than any single part could have launched
        assert work[mil][0] >= sum(w.N for w in wls) * 2, (mil, work[mil])
    else:
        assert launches["k_miller"] == 3 and launches["k_prep_prove"] == 3, launches
    # corrupted proofs are found in every part by the one verify call
    bad = [set(w.corrupt()) for w in wls]
    eng.verify_mixed_dev(vparts)
    eng.sync()
    for w, b in zip(wls, bad):
        assert b, "a part without a corrupted proof"
        want = np.ones(w.N, dtype=np.uint8)
        want[list(b)] = 0
        assert (w.ok.cpu().numpy() == want).all()
    eng.close()
    return launches


def test_mixed_merged_2p12():
    run_mixed(12, 1, seed=20241230)


def test_mixed_merged_2p14():
    run_mixed(14, 1, seed=20241231)


def test_mixed_entry_2p16_in_sequence():
    """configs[2] at its own size THROUGH the mixed entry points (planned: parts in sequence above 2^14)."""
    run_mixed(16, -1, seed=20241232)
