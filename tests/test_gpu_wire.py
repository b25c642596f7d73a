"""Wire format on the GPU (gs_wire_* + groth_sahai_rs_amd/wire.py): the reference's round-trip tests restated
(data_structures.rs:1270-1310 Com1/Com2; commit.rs:300-340 Commit1/2; prove.rs:600-640 EquProof; statement.rs:215-390
EquType and the four equation types; the CRS derive of generator.rs:35), plus byte-for-byte agreement with the
big-integer restatement (oracle/gs_wire_oracle.py) and rejection of invalid encodings."""
import os
import struct
import sys

import numpy as np
import pytest

from gsutil import REPO, curve
from test_gpu_mirror import ReplayRng, build

sys.path.insert(0, os.path.join(REPO, "oracle"))

pytestmark = pytest.mark.gpu


def hexpt(c, h, group):
    """golden hex point -> oracle value"""
    if h is None:
        return None
    if group == 1:
        return (int(h[0], 16), int(h[1], 16))
    return ((int(h[0], 16), int(h[1], 16)), (int(h[2], 16), int(h[3], 16)))


@pytest.fixture(scope="module", params=["bls12_381", "bn254"])
def env(request):
    from groth_sahai_rs_amd import mirror, wire

    import gs_oracle as O
    import gs_wire_oracle as W

    c = curve(request.param)
    O.set_curve(O._bls12_381() if request.param == "bls12_381" else O._bn254())
    g = c.golden["crs"]
    crs = mirror.CRS([c.com1(g["u"][0]), c.com1(g["u"][1])], [c.com2(g["v"][0]), c.com2(g["v"][1])], c.g1(g["g1"]),
                     c.g2(g["g2"]), c.f12(g["gt"]), curve=c.curve_id)
    return c, mirror, wire, crs, O, W


@pytest.mark.parametrize("idx", [0, 1, 2, 3, 8, 10])
def test_struct_round_trips_and_oracle_bytes(env, idx):
    c, mirror, wire, crs, O, W = env
    case = c.golden["cases"][idx]
    equ, xvars, yvars = build(c, mirror, case)
    proof = equ.commit_and_prove(xvars, yvars, crs, ReplayRng(c, [case["R"], case["S"], case["T"]]))
    ty = case["type"]
    gx, gy = 1 if ty in (0, 1) else 0, 2 if ty in (0, 2) else 0
    mat = lambda m: [[int(s, 16) for s in row] for row in m]
    com = lambda v, g: (hexpt(c, v[0], g), hexpt(c, v[1], g))
    for compressed, ser, de in ((True, wire.serialize_compressed, wire.deserialize_compressed),
                                (False, wire.serialize_uncompressed, wire.deserialize_uncompressed)):
        eng = crs.engine
        # Commit1 / Commit2
        b1 = ser(proof.xcoms, eng)
        assert b1 == W.enc_commit([com(v, 1) for v in case["xcoms"]], mat(case["R"]), 1, compressed)
        assert de(mirror.Commit1, b1, eng) == proof.xcoms
        b2 = ser(proof.ycoms, eng)
        assert b2 == W.enc_commit([com(v, 2) for v in case["ycoms"]], mat(case["S"]), 2, compressed)
        assert de(mirror.Commit2, b2, eng) == proof.ycoms
        # EquProof
        pf = proof.equ_proofs[0]
        b3 = ser(pf, eng)
        assert b3 == W.enc_equ_proof([com(v, 2) for v in case["pi"]], [com(v, 1) for v in case["theta"]], ty,
                                     mat(case["T"]), compressed)
        pf2 = de(mirror.EquProof, b3, eng)
        assert pf2.equ_type == ty and mirror._mat_eq(pf2.rand, pf.rand)
        assert all((a == b).all() for a, b in zip(pf2.pi + pf2.theta, pf.pi + pf.theta))
        # the equation itself
        ea = (lambda v: hexpt(c, v, 1)) if gx else (lambda s: int(s, 16))
        eb = (lambda v: hexpt(c, v, 2)) if gy else (lambda s: int(s, 16))
        tgt = {0: lambda t: O.f12_unflat([int(s, 16) for s in t]), 1: lambda t: hexpt(c, t, 1),
               2: lambda t: hexpt(c, t, 2), 3: lambda t: int(t, 16)}[ty](case["target"])
        b4 = ser(equ, eng)
        assert b4 == W.enc_equation(ty, [ea(v) for v in case["a"]], [eb(v) for v in case["b"]], mat(case["gamma"]), tgt,
                                    compressed)
        equ2 = de(type(equ), b4, eng)
        assert ser(equ2, eng) == b4
        # a deserialised proof still verifies against the deserialised equation
        assert equ2.verify(mirror.CProof(de(mirror.Commit1, b1, eng), de(mirror.Commit2, b2, eng), [pf2]), crs)


def test_crs_round_trip(env):
    c, mirror, wire, crs, O, W = env
    g = c.golden["crs"]
    com = lambda v, grp: (hexpt(c, v[0], grp), hexpt(c, v[1], grp))
    ocrs = {"u": [com(v, 1) for v in g["u"]], "v": [com(v, 2) for v in g["v"]], "g1": hexpt(c, g["g1"], 1),
            "g2": hexpt(c, g["g2"], 2), "gt": O.f12_unflat([int(s, 16) for s in g["gt"]])}
    for compressed, ser, de in ((True, wire.serialize_compressed, wire.deserialize_compressed),
                                (False, wire.serialize_uncompressed, wire.deserialize_uncompressed)):
        b = ser(crs)
        assert b == W.enc_crs(ocrs, compressed)
        crs2 = de(mirror.CRS, b, curve=c.curve_id)
        assert ser(crs2) == b


def test_equ_type_byte_and_invalid_data(env):
    c, mirror, wire, crs, O, W = env
    eng = crs.engine
    case = c.golden["cases"][0]
    equ, xvars, yvars = build(c, mirror, case)
    proof = equ.commit_and_prove(xvars, yvars, crs, ReplayRng(c, [case["R"], case["S"], case["T"]]))
    b = bytearray(wire.serialize_compressed(proof.equ_proofs[0], eng))
    ws = eng.wire_sizes()
    tpos = 8 + 2 * 2 * ws["g2c"] + 8 + 2 * 2 * ws["g1c"]  # pi (2 Com2), theta (2 Com1), then the type byte
    assert b[tpos] == 0  # EquType::PairingProduct (statement.rs:68-73)
    bad = bytearray(b)
    bad[tpos] = 4
    with pytest.raises(wire.SerializationError):
        wire.deserialize_compressed(mirror.EquProof, bytes(bad), eng)
    with pytest.raises(wire.SerializationError):  # truncated
        wire.deserialize_compressed(mirror.EquProof, bytes(b[:-1]), eng)
    with pytest.raises(wire.SerializationError):  # trailing byte
        wire.deserialize_compressed(mirror.EquProof, bytes(b) + b"\0", eng)
    with pytest.raises(wire.SerializationError):  # absurd length prefix
        wire.deserialize_compressed(mirror.EquProof, struct.pack("<Q", 1 << 40) + bytes(b[8:]), eng)
    bad = bytearray(b)
    bad[-1] = 0xFF  # last scalar of rand >= r
    bad[-2] = 0xFF
    with pytest.raises(wire.SerializationError):
        wire.deserialize_compressed(mirror.EquProof, bytes(bad), eng)
    # a curve point outside the prime-order subgroup passes only without validation (G2 has a cofactor on both curves)
    n = ws["g2c"]
    for x in range(1, 60):
        xs = [x, 1]
        if c.name == "bls12_381":
            enc = bytearray(b"".join(v.to_bytes(n // 2, "big") for v in reversed(xs)))
            enc[0] |= 0x80
        else:
            enc = bytearray(b"".join(v.to_bytes(n // 2, "little") for v in xs))
        vals, ok = eng.wire_decode("g2", np.frombuffer(bytes(enc), dtype=np.uint8), True, False)
        if ok[0]:
            _, ok2 = eng.wire_decode("g2", np.frombuffer(bytes(enc), dtype=np.uint8), True, True)
            assert ok2[0] == 0
            break
    else:
        pytest.fail("no curve point found")
    # GT: the generator's pairing value is in the r-torsion; a random Fp12 element is not
    gt = c.f12(c.golden["crs"]["gt"])
    enc = eng.wire_encode("gt", gt.reshape(1, -1))
    assert enc.tobytes() == W.enc_gt(O.f12_unflat([int(s, 16) for s in c.golden["crs"]["gt"]]))
    back, ok = eng.wire_decode("gt", enc, validate=True)
    assert ok[0] == 1 and (back.view(np.uint64).reshape(-1) == gt).all()
    junk = np.frombuffer(W.enc_gt(O.f12_unflat(list(range(2, 14)))), dtype=np.uint8)
    _, ok = eng.wire_decode("gt", junk, validate=True)
    assert ok[0] == 0
    _, ok = eng.wire_decode("gt", junk, validate=False)
    assert ok[0] == 1
