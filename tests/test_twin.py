"""CPU twin of the HIP arithmetic headers vs the golden fixtures (no GPU).

Validates the device ALGORITHMS (Montgomery CIOS, tower, Jacobian formulas,
projective Miller loop, arkworks-exponent final exponentiation) by compiling
the same .cuh sources for the host.  The GPU parity tests proper are in
test_gpu_parity.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from gsutil import HERE, REPO, curve, ptr

TWIN_SRC = os.path.join(HERE, "twin", "host_twin.cpp")
TWIN_SO = os.path.join(HERE, "twin", "libhost_twin.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def twin():
    srcs = [TWIN_SRC] + [
        os.path.join(REPO, "groth_sahai_rs_amd", "csrc", f)
        for f in os.listdir(os.path.join(REPO, "groth_sahai_rs_amd", "csrc"))
        if f.endswith((".cuh", ".h"))
    ]
    if not os.path.exists(TWIN_SO) or any(os.path.getmtime(s) > os.path.getmtime(TWIN_SO) for s in srcs):
        if not os.path.exists(CLANG):
            pytest.skip("no host clang++ for the CPU twin")
        subprocess.check_call([CLANG, "-O2", "-std=c++17", "-Wno-psabi", "-DGS_FQ28_CHECK", "-shared", "-fPIC", TWIN_SRC,
                               "-o", TWIN_SO])
    return ctypes.CDLL(TWIN_SO)


CURVES = ["bls12_381", "bn254"]


@pytest.mark.parametrize("cname", CURVES)
def test_field_ops(twin, cname):
    c = curve(cname)
    rng = np.random.default_rng(1)
    for _ in range(50):
        a = int.from_bytes(rng.bytes(48), "little") % c.p
        b = int.from_bytes(rng.bytes(48), "little") % c.p
        out = np.zeros(c.nq, dtype=np.uint64)
        getattr(twin, "twin_fp_mul_" + cname)(ptr(c.fq(a)), ptr(c.fq(b)), ptr(out))
        assert c.fq_dec(out) == a * b % c.p
        out4 = np.zeros(4 * c.nq, dtype=np.uint64)
        getattr(twin, "twin_fp_addsub_" + cname)(ptr(c.fq(a)), ptr(c.fq(b)), ptr(out4))
        o = out4.reshape(4, c.nq)
        assert c.fq_dec(o[0]) == (a + b) % c.p
        assert c.fq_dec(o[1]) == (a - b) % c.p
        assert c.fq_dec(o[2]) == (-a) % c.p
        assert c.fq_dec(o[3]) == (8 * a - 5 * b) % c.p
        assert getattr(twin, "twin_fp_is_zero_" + cname)(ptr(c.fq(a)), ptr(c.fq(a))) == 1
        assert getattr(twin, "twin_fp_is_zero_" + cname)(ptr(c.fq(a)), ptr(c.fq(b))) == (1 if a == b else 0)
        x = int.from_bytes(rng.bytes(32), "little") % c.r
        y = int.from_bytes(rng.bytes(32), "little") % c.r
        outr = np.zeros(c.nr, dtype=np.uint64)
        getattr(twin, "twin_fr_mul_" + cname)(ptr(c.fr(x)), ptr(c.fr(y)), ptr(outr))
        assert c.fr_dec(outr) == x * y % c.r
    for a in (1, 2, c.p - 1, 0x1234567):
        out = np.zeros(c.nq, dtype=np.uint64)
        getattr(twin, "twin_fp_inv_" + cname)(ptr(c.fq(a)), ptr(out))
        assert c.fq_dec(out) == pow(a, -1, c.p)
    # edge values: p-1 squared, zero
    out = np.zeros(c.nq, dtype=np.uint64)
    getattr(twin, "twin_fp_mul_" + cname)(ptr(c.fq(c.p - 1)), ptr(c.fq(c.p - 1)), ptr(out))
    assert c.fq_dec(out) == 1


@pytest.mark.parametrize("cname", CURVES)
def test_inversion_safegcd(twin, cname):
    """Round 4: inv() is the constant-time safegcd (divsteps in groups of 28 on signed 28-bit limbs, gs_fq28.cuh), not
    Fermat any more: 400 random values, powers of two and their neighbours, values next to 0 and p, and inv(0) = 0
    (what the reduction kernels rely on for identity slots) against Python's modular inverse."""
    c = curve(cname)
    rng = np.random.default_rng(2024)
    vals = [int.from_bytes(rng.bytes(64), "little") % c.p for _ in range(400)]
    vals += [1, 2, 3, c.p - 1, c.p - 2, (c.p - 1) // 2, (c.p + 1) // 2]
    for k in range(1, c.p.bit_length(), 7):
        vals += [(1 << k) % c.p, ((1 << k) - 1) % c.p, (c.p - (1 << k)) % c.p]
    inv = getattr(twin, "twin_fp_inv_" + cname)
    for a in vals:
        out = np.zeros(c.nq, dtype=np.uint64)
        inv(ptr(c.fq(a)), ptr(out))
        assert c.fq_dec(out) == pow(a, -1, c.p), hex(a)
    out = np.ones(c.nq, dtype=np.uint64)
    inv(ptr(c.fq(0)), ptr(out))
    assert c.fq_dec(out) == 0


@pytest.mark.parametrize("cname", CURVES)
def test_smul_golden(twin, cname):
    c = curve(cname)
    g = c.golden
    g1 = c.g1(g["g1_smul"][0]["out"])  # k = 1 -> generator
    g2 = c.g2(g["g2_smul"][0]["out"])
    for e in g["g1_smul"]:
        out = np.zeros(2 * c.nq, dtype=np.uint64)
        getattr(twin, "twin_g1_smul_" + cname)(ptr(g1), ptr(c.fr_hex(e["k"])), ptr(out))
        assert c.g1_dec(out) == e["out"], e["k"]
    for e in g["g2_smul"]:
        out = np.zeros(4 * c.nq, dtype=np.uint64)
        getattr(twin, "twin_g2_smul_" + cname)(ptr(g2), ptr(c.fr_hex(e["k"])), ptr(out))
        assert c.g2_dec(out) == e["out"], e["k"]
    # k = 0 and identity base
    out = np.ones(2 * c.nq, dtype=np.uint64)
    getattr(twin, "twin_g1_smul_" + cname)(ptr(g1), ptr(c.fr(0)), ptr(out))
    assert not out.any()
    out = np.ones(2 * c.nq, dtype=np.uint64)
    getattr(twin, "twin_g1_smul_" + cname)(ptr(c.g1(None)), ptr(c.fr(5)), ptr(out))
    assert not out.any()
    # P + (-P) = O, P + P = 2P through the generic add
    two = c.g1(g["g1_smul"][1]["out"])
    m1 = c.g1(g["g1_smul"][-1]["out"])  # (r-1) g = -g
    out = np.ones(2 * c.nq, dtype=np.uint64)
    getattr(twin, "twin_g1_add_" + cname)(ptr(g1), ptr(m1), ptr(out))
    assert not out.any()
    getattr(twin, "twin_g1_add_" + cname)(ptr(g1), ptr(g1), ptr(out))
    assert (out == two).all()
    # mixed add on G2: P + P (doubling branch), P + (-P) (identity), P + O
    two2 = c.g2(g["g2_smul"][1]["out"])
    m2 = c.g2(g["g2_smul"][-1]["out"])
    out2 = np.ones(4 * c.nq, dtype=np.uint64)
    getattr(twin, "twin_g2_madd_" + cname)(ptr(g2), ptr(g2), ptr(out2))
    assert (out2 == two2).all()
    getattr(twin, "twin_g2_madd_" + cname)(ptr(g2), ptr(m2), ptr(out2))
    assert not out2.any()
    getattr(twin, "twin_g2_madd_" + cname)(ptr(g2), ptr(c.g2(None)), ptr(out2))
    assert (out2 == g2).all()


@pytest.mark.parametrize("cname", CURVES)
def test_fp12_golden(twin, cname):
    c = curve(cname)
    t = c.golden["fp12"]
    a, b = c.f12(t["a"]), c.f12(t["b"])
    f = getattr(twin, "twin_fp12_op_" + cname)
    out = np.zeros(12 * c.nq, dtype=np.uint64)
    for op, key in [(0, "mul"), (1, "sqr"), (2, "inv"), (3, "conj"), (4, "frob1"), (5, "frob2")]:
        f(op, ptr(a), ptr(b), ptr(out))
        assert c.f12_dec(out) == t[key], key
    # a is a pairing value => in the cyclotomic subgroup: Granger-Scott == plain square
    f(7, ptr(a), None, ptr(out))
    assert c.f12_dec(out) == t["sqr"]
    # frob3 = frob1(frob2)
    tmp = np.zeros_like(out)
    f(5, ptr(a), None, ptr(tmp))
    f(4, ptr(tmp), None, ptr(out))
    f(6, ptr(a), None, ptr(tmp))
    assert (tmp == out).all()


@pytest.mark.parametrize("cname", CURVES)
def test_pairing_golden(twin, cname):
    c = curve(cname)
    f = getattr(twin, "twin_multi_pairing_" + cname)
    for e in c.golden["pairing"]:
        out = np.zeros(12 * c.nq, dtype=np.uint64)
        f(1, ptr(c.g1(e["p"])), ptr(c.g2(e["q"])), ptr(out), 1)
        assert c.f12_dec(out) == e["out"]
    # pairing_sum fixture: 4 cells, identity arguments skipped
    ps = c.golden["pairing_sum"]
    for cell, (ia, ib) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        P = np.concatenate([c.g1(x[ia]) for x in ps["x"]])
        Q = np.concatenate([c.g2(y[ib]) for y in ps["y"]])
        out = np.zeros(12 * c.nq, dtype=np.uint64)
        f(len(ps["x"]), ptr(P), ptr(Q), ptr(out), 1)
        assert c.f12_dec(out) == ps["out"][cell], cell


@pytest.mark.parametrize("cname", CURVES)
def test_straus_msm_equals_left_mul_fixture(twin, cname):
    """Joint MSM (Straus, shared doublings, endomorphism streams) against the golden
    Matrix<Com>::left_mul fixture: out[i].c = sum_k lhs[i][k] * col[k].c"""
    c = curve(cname)
    lm = c.golden["left_mul"]
    rows, k = len(lm["lhs"]), len(lm["lhs"][0])
    for i in range(rows):
        ks = np.concatenate([c.fr_hex(s) for s in lm["lhs"][i]])
        for comp in range(2):
            P1 = np.concatenate([c.g1(v[comp]) for v in lm["com1"]])
            out = np.zeros(2 * c.nq, dtype=np.uint64)
            getattr(twin, "twin_g1_msm_" + cname)(k, ptr(P1), ptr(ks), ptr(out))
            assert c.g1_dec(out) == lm["out1"][i][comp]
            P2 = np.concatenate([c.g2(v[comp]) for v in lm["com2"]])
            out2 = np.zeros(4 * c.nq, dtype=np.uint64)
            getattr(twin, "twin_g2_msm_" + cname)(k, ptr(P2), ptr(ks), ptr(out2))
            assert c.g2_dec(out2) == lm["out2"][i][comp]


@pytest.mark.parametrize("cname", CURVES)
def test_cooperative_final_exponentiation(twin, cname):
    """gs_coop.cuh: the 3-lane f^x and the final exponentiation built on it give, on every lane, exactly the
    one-lane results -- on cyclotomic inputs for f^x (Granger-Scott squaring) and on raw Miller outputs for
    the whole exponentiation (golden pairing values)."""
    c = curve(cname)
    mp = getattr(twin, "twin_multi_pairing_" + cname)
    coop = getattr(twin, "twin_coop_" + cname)
    expx = getattr(twin, "twin_exp_by_x_" + cname)
    for e in c.golden["pairing"][:3]:
        miller = np.zeros(12 * c.nq, dtype=np.uint64)
        mp(1, ptr(c.g1(e["p"])), ptr(c.g2(e["q"])), ptr(miller), 0)
        out3 = np.zeros(3 * 12 * c.nq, dtype=np.uint64)
        coop(1, ptr(miller), ptr(out3))
        for j in range(3):
            assert c.f12_dec(out3.reshape(3, -1)[j]) == e["out"], j
        # f^x on the (cyclotomic) pairing value itself
        gt = c.f12(e["out"])
        want = np.zeros(12 * c.nq, dtype=np.uint64)
        expx(ptr(gt), ptr(want))
        coop(0, ptr(gt), ptr(out3))
        for j in range(3):
            assert (out3.reshape(3, -1)[j] == want).all(), j


@pytest.mark.parametrize("cname", CURVES)
def test_endomorphism_scalar_multiplication_random(twin, cname):
    """GLV / GLS paths (BLS12-381: division by lambda / base-|x| digits; BN254: lattice decomposition with per-scalar
    signs) on random and extreme scalars against the big-integer oracle, both groups."""
    import os
    import sys

    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import gs_oracle as O

    c = curve(cname)
    oc = O.set_curve(O._bls12_381() if cname == "bls12_381" else O._bn254())
    rng = np.random.default_rng(99)
    ks = [0, 1, c.r - 1, c.r - 2, (1 << 128) - 1, 1 << 127, (1 << 64) + 1, c.r // 2]
    ks += [int.from_bytes(rng.bytes(40), "little") % c.r for _ in range(24)]
    g1, g2 = c.g1(c.golden["g1_smul"][2]["out"]), c.g2(c.golden["g2_smul"][2]["out"])  # 3 * generator
    P1, P2 = O.g1_mul(3, oc.g1), O.g2_mul(3, oc.g2)
    for k in ks:
        out = np.zeros(2 * c.nq, dtype=np.uint64)
        getattr(twin, "twin_g1_smul_" + cname)(ptr(g1), ptr(c.fr(k)), ptr(out))
        want = O.g1_mul(k, P1)
        assert c.g1_dec(out) == (None if want is None else ["%x" % want[0], "%x" % want[1]]), hex(k)
        out = np.zeros(4 * c.nq, dtype=np.uint64)
        getattr(twin, "twin_g2_smul_" + cname)(ptr(g2), ptr(c.fr(k)), ptr(out))
        want = O.g2_mul(k, P2)
        assert c.g2_dec(out) == (None if want is None else ["%x" % v for v in (want[0][0], want[0][1], want[1][0], want[1][1])]), hex(k)


@pytest.mark.parametrize("cname", CURVES)
def test_fixed_argument_line_tables(twin, cname):
    """Pairs whose G2 argument is tabulated (miller_line_table) give the same pairing product as stepping the twist
    point, in the single, the twin and the lane-pair Miller loop (the latter as two host threads with the device's put / get
    exchange discipline), for subsets of tabulated pairs that leave odd and even numbers of stepping triples."""
    c = curve(cname)
    ps = c.golden["pairing_sum"]
    P = np.concatenate([c.g1(x[1]) for x in ps["x"]])
    Q = np.concatenate([c.g2(y[1]) for y in ps["y"]])
    n = len(ps["x"])
    f = getattr(twin, "twin_multi_pairing_fixed_" + cname)
    for twin_mode in (0, 1, 2):  # single accumulator, twin lane, lane pair (two host threads, multi_miller_pair)
        outs = []
        for mask in (0, 1, (1 << n) - 1, 0b1010 & ((1 << n) - 1)):
            out = np.zeros(2 * 12 * c.nq, dtype=np.uint64)
            f(n, ptr(P), ptr(Q), mask, ptr(out), twin_mode)
            outs.append(out)
        assert c.f12_dec(outs[0][:12 * c.nq]) == ps["out"][3]  # cell (1, 1) of the fixture
        for o in outs[1:]:
            assert (o == outs[0]).all()
        if twin_mode:  # both accumulators were given the same G1 arguments
            assert (outs[0][12 * c.nq:] == outs[0][:12 * c.nq]).all()
    # longer lists (the fixture's pairs three times over: odd and even numbers of stepping triples, up to three rounds of
    # the lane pair, with and without tabulated pairs behind them): the lane pair equals the twin lane
    P3, Q3, n3 = np.tile(P, 3), np.tile(Q, 3), 3 * n
    for mask in (0, 1, 0b11 << (n3 - 2), (1 << n3) - 2):
        a, b = np.zeros(2 * 12 * c.nq, dtype=np.uint64), np.zeros(2 * 12 * c.nq, dtype=np.uint64)
        f(n3, ptr(P3), ptr(Q3), mask, ptr(a), 1)
        f(n3, ptr(P3), ptr(Q3), mask, ptr(b), 2)
        assert (a == b).all(), mask
