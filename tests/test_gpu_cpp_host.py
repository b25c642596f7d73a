"""C++ host layer (include/gs_amd.hpp -- the compiled-language mirror of the reference's
Provable / Verifiable / commit API) driven by tests/cpp/test_prover.cpp, the reference's
tests/prover.rs restated in C++.  The golden cases are serialised to a small blob, the
program is built with g++ against the in-tree libgs_amd.so and run on the GPU."""
import os
import struct
import subprocess

import numpy as np
import pytest

from gsutil import HERE, REPO, curve

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
    with open(path, "wb") as f:
        f.write(struct.pack("<4I", c.curve_id, ty, case["m"], case["n"]))
        for s in secs:
            f.write(struct.pack("<Q", len(s)))
            f.write(s)


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
