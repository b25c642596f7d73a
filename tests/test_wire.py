"""Wire format: the device encoders/decoders of csrc/gs_wire.cuh, compiled for the host by the CPU
twin, against the big-integer restatement oracle/gs_wire_oracle.py and against the published compressed encodings of
the BLS12-381 generators.  The reference's own tests for this path are round trips (data_structures.rs:1270-1310);
they are restated on the GPU in test_gpu_wire.py."""
import os
import sys

import numpy as np
import pytest

from gsutil import REPO, curve, ptr
from test_twin import twin  # noqa: F401  (fixture)

sys.path.insert(0, os.path.join(REPO, "oracle"))
import gs_oracle as O  # noqa: E402
import gs_wire_oracle as W  # noqa: E402

CURVES = ["bls12_381", "bn254"]
# the standard generators in the zcash / IETF BLS-signature compressed form (public constants)
G1_GEN_COMPRESSED = ("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac58"
                     "6c55e83ff97a1aeffb3af00adb22c6bb")
G2_GEN_COMPRESSED = ("93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
                     "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")


def setc(name):
    return O.set_curve(O._bls12_381() if name == "bls12_381" else O._bn254())


def sizes(c, group, compressed):
    return (1 if group == 1 else 2) * (1 if compressed else 2) * c.nq * 8


def to_boundary(c, pt, group):
    if pt is None:
        return np.zeros((2 if group == 1 else 4) * c.nq, dtype=np.uint64)
    if group == 1:
        return np.concatenate([c.fq(pt[0]), c.fq(pt[1])])
    return np.concatenate([c.fq(pt[0][0]), c.fq(pt[0][1]), c.fq(pt[1][0]), c.fq(pt[1][1])])


def test_oracle_reproduces_published_generator_encodings():
    oc = setc("bls12_381")
    assert W.enc_point(oc.g1, 1, True).hex() == G1_GEN_COMPRESSED
    assert W.enc_point(oc.g2, 2, True).hex() == G2_GEN_COMPRESSED
    assert W.dec_point(bytes.fromhex(G1_GEN_COMPRESSED), 1, True) == oc.g1
    assert W.dec_point(bytes.fromhex(G2_GEN_COMPRESSED), 2, True) == oc.g2
    # identity: 0xc0 then zeros (compressed), 0x40 then zeros (uncompressed)
    assert W.enc_point(None, 1, True) == bytes([0xC0]) + bytes(47)
    assert W.enc_point(None, 2, False) == bytes([0x40]) + bytes(191)


@pytest.mark.parametrize("cname", CURVES)
def test_device_point_codec_matches_oracle(twin, cname):  # noqa: F811
    c, oc = curve(cname), setc(cname)
    enc, dec = getattr(twin, "twin_wire_enc_" + cname), getattr(twin, "twin_wire_dec_" + cname)
    for group, gen, mul in ((1, oc.g1, O.g1_mul), (2, oc.g2, O.g2_mul)):
        pts = [None, gen] + [mul(k, gen) for k in (2, 3, 0xDEADBEEF, oc.r - 1, 0x1234567890ABCDEF1234567890ABCDEF)]
        for compressed in (True, False):
            n = sizes(c, group, compressed)
            for pt in pts:
                want = W.enc_point(pt, group, compressed)
                got = np.zeros(n, dtype=np.uint8)
                enc(group, int(compressed), ptr(to_boundary(c, pt, group)), ptr(got))
                assert got.tobytes() == want, (group, compressed, pt)
                back = np.zeros((2 if group == 1 else 4) * c.nq, dtype=np.uint64)
                assert dec(group, int(compressed), 1, ptr(got), ptr(back)) == 1
                assert (back == to_boundary(c, pt, group)).all()
                assert W.dec_point(want, group, compressed) == pt
    if cname == "bls12_381":
        got = np.zeros(48, dtype=np.uint8)
        enc(1, 1, ptr(to_boundary(c, oc.g1, 1)), ptr(got))
        assert got.tobytes().hex() == G1_GEN_COMPRESSED


@pytest.mark.parametrize("cname", CURVES)
def test_device_decoder_rejects_what_the_oracle_rejects(twin, cname):  # noqa: F811
    c, oc = curve(cname), setc(cname)
    dec = getattr(twin, "twin_wire_dec_" + cname)
    zc = cname == "bls12_381"

    def both(buf, group, compressed, validate=True):
        out = np.zeros((2 if group == 1 else 4) * c.nq, dtype=np.uint64)
        a = np.frombuffer(bytes(buf), dtype=np.uint8).copy()
        got = dec(group, int(compressed), int(validate), ptr(a), ptr(out))
        try:
            W.dec_point(bytes(buf), group, compressed, validate)
            want = 1
        except ValueError:
            want = 0
        assert got == want, (group, compressed, validate, bytes(buf).hex())
        return got

    for group, gen in ((1, oc.g1), (2, oc.g2)):
        good_c, good_u = bytearray(W.enc_point(gen, group, True)), bytearray(W.enc_point(gen, group, False))
        assert both(good_c, group, True) == 1 and both(good_u, group, False) == 1
        # y replaced by y + 1: off the curve
        bad = bytearray(good_u)
        bad[-1 if zc else len(bad) // 2] ^= 1
        assert both(bad, group, False) == 0
        # wrong compression flag / infinity flag with non-zero x
        bad = bytearray(good_c)
        if zc:
            bad[0] &= 0x7F
            assert both(bad, group, True) == 0
            bad = bytearray(good_c)
            bad[0] |= 0x40
        else:
            bad[-1] |= 0x40
        assert both(bad, group, True) == 0
        # the identity: exactly one encoding (flag + all-zero payload); the sort flag together with the infinity flag is
        # malformed (ark-bls12-381 EncodingFlags), and so is a non-zero payload under the infinity flag
        ident = bytearray(W.enc_point(None, group, True))
        assert both(ident, group, True) == 1
        bad = bytearray(ident)
        if zc:
            bad[0] |= 0x20
        else:
            bad[-1] |= 0x80
        assert both(bad, group, True) == 0
        bad = bytearray(ident)
        bad[len(bad) // 2] ^= 1
        assert both(bad, group, True) == 0
        # non-canonical x (= p): rejected before any arithmetic
        n = c.nq * 8
        pbytes = c.p.to_bytes(n, "big" if zc else "little")
        bad = bytearray(good_c)
        if zc:
            bad[:n] = pbytes
            bad[0] |= 0x80
        else:
            bad[(len(bad) - n):] = pbytes
        assert both(bad, group, True) == 0
        # x values with and without a point above them: the sqrt-existence branch, and for curve points that are
        # not in the prime-order subgroup the validate switch (G1 of BN254 has cofactor 1: always accepted)
        seen = set()
        for x in range(1, 40):
            xs = [x] if group == 1 else [x, 1]
            if zc:
                buf = bytearray(b"".join(v.to_bytes(n, "big") for v in reversed(xs)))
                buf[0] |= 0x80
            else:
                buf = bytearray(b"".join(v.to_bytes(n, "little") for v in xs))
            seen.add((both(buf, group, True, validate=False), both(buf, group, True, validate=True)))
        assert (0, 0) in seen and ((1, 0) in seen or (cname == "bn254" and group == 1 and (1, 1) in seen))
