"""C++ host layer (include/gs_amd.hpp -- the compiled-language mirror of the reference's
Provable / Verifiable / commit API) driven by tests/cpp/test_prover.cpp, the reference's
tests/prover.rs restated in C++.  The golden cases are serialised to a small blob, the
program is built with g++ against the in-tree libgs_amd.so and run on the GPU."""
import os
import struct
import subprocess

import sys

import numpy as np
import pytest

from gsutil import HERE, REPO, curve

sys.path.insert(0, os.path.join(REPO, "oracle"))

BUILD = os.path.join(HERE, "cpp", "_build")
LIBDIR = os.path.join(REPO, "groth_sahai_rs_amd", "lib")


def build_program():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "test_prover")
    src = os.path.join(HERE, "cpp", "test_prover.cpp")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(REPO, "include"), src, "-o", exe,
           "-L" + LIBDIR, "-lgs_amd", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def write_case(c, case, path):
    ty = case["type"]
    ex = c.g1 if ty in (0, 1) else c.fr_hex
    ey = c.g2 if ty in (0, 2) else c.fr_hex
    g = c.golden["crs"]
    cat = lambda xs: np.concatenate([np.asarray(x, dtype=np.uint64).reshape(-1) for x in xs]).tobytes()
    tgt = {0: c.f12, 1: c.g1, 2: c.g2, 3: c.fr_hex}[ty](case["target"])
    secs = [c.com1(g["u"][0]).tobytes(), c.com1(g["u"][1]).tobytes(), c.com2(g["v"][0]).tobytes(),
            c.com2(g["v"][1]).tobytes(), c.g1(g["g1"]).tobytes(), c.g2(g["g2"]).tobytes(), c.f12(g["gt"]).tobytes(),
            cat([ex(v) for v in case["xvars"]]), cat([ey(v) for v in case["yvars"]]),
            cat([ex(v) for v in case["a"]]), cat([ey(v) for v in case["b"]]), c.fr_mat(case["gamma"]).tobytes(),
            tgt.tobytes(), c.fr_mat(case["R"]).tobytes(), c.fr_mat(case["S"]).tobytes(), c.fr_mat(case["T"]).tobytes(),
            cat([c.com1(v) for v in case["xcoms"]]), cat([c.com2(v) for v in case["ycoms"]]),
            cat([c.com2(v) for v in case["pi"]]), cat([c.com1(v) for v in case["theta"]])]
    secs += wire_expectations(c, case)
    with open(path, "wb") as f:
        f.write(struct.pack("<4I", c.curve_id, ty, case["m"], case["n"]))
        for s in secs:
            f.write(struct.pack("<Q", len(s)))
            f.write(s)


def wire_expectations(c, case):
    """Commit1 (compressed), EquProof (compressed), equation (uncompressed), CRS (compressed) from the oracle."""
    import gs_oracle as O
    import gs_wire_oracle as W

    O.set_curve(O._bls12_381() if c.name == "bls12_381" else O._bn254())
    ty = case["type"]

    def pt(h, g):
        if h is None:
            return None
        return (int(h[0], 16), int(h[1], 16)) if g == 1 else ((int(h[0], 16), int(h[1], 16)), (int(h[2], 16), int(h[3], 16)))

    fr = lambda s: int(s, 16)
    mat = lambda m: [[fr(s) for s in row] for row in m]
    com = lambda v, g: (pt(v[0], g), pt(v[1], g))
    gx, gy = ty in (0, 1), ty in (0, 2)
    a = [pt(v, 1) if gx else fr(v) for v in case["a"]]
    b = [pt(v, 2) if gy else fr(v) for v in case["b"]]
    tgt = {0: lambda t: O.f12_unflat([fr(s) for s in t]), 1: lambda t: pt(t, 1), 2: lambda t: pt(t, 2), 3: fr}[ty](
        case["target"])
    g = c.golden["crs"]
    ocrs = {"u": [com(v, 1) for v in g["u"]], "v": [com(v, 2) for v in g["v"]], "g1": pt(g["g1"], 1),
            "g2": pt(g["g2"], 2), "gt": O.f12_unflat([fr(s) for s in g["gt"]])}
    return [W.enc_commit([com(v, 1) for v in case["xcoms"]], mat(case["R"]), 1, True),
            W.enc_equ_proof([com(v, 2) for v in case["pi"]], [com(v, 1) for v in case["theta"]], ty, mat(case["T"]), True),
            W.enc_equation(ty, a, b, mat(case["gamma"]), tgt, False), W.enc_crs(ocrs, True)]


def test_cpp_host_layer_builds():
    """CPU: the header compiles warning-free and the program links against the C ABI."""
    assert os.path.exists(build_program())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bls12_381", "bn254"])
def test_cpp_prover_tests(name, tmp_path):
    exe = build_program()
    c = curve(name)
    for case in c.golden["cases"][:8]:  # the reference's 2x1 cases and the dense 2x2 ones, all four types
        p = str(tmp_path / (case["name"] + ".bin"))
        write_case(c, case, p)
        r = subprocess.run([exe, p], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.startswith("OK"), (case["name"], r.stdout, r.stderr)
