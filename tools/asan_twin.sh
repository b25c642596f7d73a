#!/bin/bash
# CPU-only sanitizer run of the device arithmetic (GPU ASan is not available on the pool):
# builds the CPU twin with -fsanitize=address,undefined and the limb-contract checks, then drives
# scalar multiplication, Straus MSM and a full pairing for both curves against the golden fixtures.
set -e
cd "$(dirname "$0")/.."
CLANG=/opt/rocm/lib/llvm/bin/clang++
$CLANG -O1 -g -std=c++17 -Wno-psabi -DGS_FQ28_CHECK -fsanitize=address,undefined -fno-sanitize-recover=undefined \
  -shared -fPIC tests/twin/host_twin.cpp -o /tmp/libhost_twin_asan.so
LD_PRELOAD=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 python3 - <<'PY'
import ctypes, sys
sys.path.insert(0, "tests")
import numpy as np
from gsutil import curve, ptr
twin = ctypes.CDLL("/tmp/libhost_twin_asan.so")
for cname in ("bls12_381", "bn254"):
    c = curve(cname); g = c.golden
    g1 = c.g1(g["g1_smul"][0]["out"]); g2 = c.g2(g["g2_smul"][0]["out"])
    for e in g["g1_smul"][:5]:
        out = np.zeros(2 * c.nq, dtype=np.uint64)
        getattr(twin, "twin_g1_smul_" + cname)(ptr(g1), ptr(c.fr_hex(e["k"])), ptr(out)); assert c.g1_dec(out) == e["out"]
    for e in g["g2_smul"][:5]:
        out = np.zeros(4 * c.nq, dtype=np.uint64)
        getattr(twin, "twin_g2_smul_" + cname)(ptr(g2), ptr(c.fr_hex(e["k"])), ptr(out)); assert c.g2_dec(out) == e["out"]
    pe = g["pairing"][1]
    out = np.zeros(12 * c.nq, dtype=np.uint64)
    getattr(twin, "twin_multi_pairing_" + cname)(1, ptr(c.g1(pe["p"])), ptr(c.g2(pe["q"])), ptr(out), 1)
    assert c.f12_dec(out) == pe["out"]
    lm = g["left_mul"]; ks = np.concatenate([c.fr_hex(s) for s in lm["lhs"][0]])
    P2 = np.concatenate([c.g2(v[1]) for v in lm["com2"]]); out2 = np.zeros(4 * c.nq, dtype=np.uint64)
    getattr(twin, "twin_g2_msm_" + cname)(3, ptr(P2), ptr(ks), ptr(out2)); assert c.g2_dec(out2) == lm["out2"][0][1]
    # 3-lane cooperative final exponentiation (three host threads), line tables, wire codecs
    miller = np.zeros(12 * c.nq, dtype=np.uint64)
    getattr(twin, "twin_multi_pairing_" + cname)(1, ptr(c.g1(pe["p"])), ptr(c.g2(pe["q"])), ptr(miller), 0)
    out3 = np.zeros(3 * 12 * c.nq, dtype=np.uint64)
    getattr(twin, "twin_coop_" + cname)(1, ptr(miller), ptr(out3)); assert c.f12_dec(out3[:12 * c.nq]) == pe["out"]
    ps = g["pairing_sum"]; n = len(ps["x"])
    P = np.concatenate([c.g1(x[1]) for x in ps["x"]]); Q = np.concatenate([c.g2(y[1]) for y in ps["y"]])
    for tw in (0, 1):
        o = np.zeros(2 * 12 * c.nq, dtype=np.uint64)
        getattr(twin, "twin_multi_pairing_fixed_" + cname)(n, ptr(P), ptr(Q), (1 << n) - 1, ptr(o), tw)
        assert c.f12_dec(o[:12 * c.nq]) == ps["out"][3]
    for grp, pt, nb in ((1, g1, 2), (2, g2, 4)):
        for comp in (1, 0):
            enc = np.zeros(nb * c.nq * 8, dtype=np.uint8)
            getattr(twin, "twin_wire_enc_" + cname)(grp, comp, ptr(pt), ptr(enc))
            back = np.zeros(nb * c.nq, dtype=np.uint64)
            sz = (1 if comp else 2) * (nb // 2) * c.nq * 8
            assert getattr(twin, "twin_wire_dec_" + cname)(grp, comp, 1, ptr(enc[:sz].copy()), ptr(back)) == 1
            assert (back == pt).all()
    print(cname, "asan/ubsan clean")
PY
