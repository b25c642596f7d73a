"""Fq multiplications per device primitive, counted by the CPU twin (the device headers compiled for the host with
-DGS_FQ28_CHECK, whose mul() increments a counter).  Output: profiles/r4/fq_mul_counts.json (latest round), read by bench.py to turn
the per-kernel work items reported by gs_prof_get_work into "useful Fq multiplications per step" (ALU roofline).
A second counter gives the multiply-add INSTRUCTIONS the device kernels execute for the same primitive (static counts
of the generated multiplier kernels: a squaring is 301 mads, not the 392 of the product it is credited as): the
"executed_mads" table, bench.py's roofline.alu.executed_mad_frac.
Run from the repo root:  python tools/count_fq_muls.py"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gsutil import curve, ptr  # noqa: E402

SO = os.path.join(ROOT, "tests", "twin", "libhost_twin.so")
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-O2", "-std=c++17", "-Wno-psabi", "-DGS_FQ28_CHECK", "-shared",
                       "-fPIC", os.path.join(ROOT, "tests", "twin", "host_twin.cpp"), "-o", SO])
twin = ctypes.CDLL(SO)
out = {}
for name in ("bls12_381", "bn254"):
    c = curve(name)
    f0 = getattr(twin, "twin_opcount_" + name)
    f0.restype = ctypes.c_long
    lm = getattr(twin, "twin_last_mads_" + name)
    lm.restype = ctypes.c_long
    MADS = [False]

    def f(*a):  # Fq multiplications, or (MADS[0]) the executed multiply-adds of the same primitive
        v = f0(*a)
        return lm() if MADS[0] else v
    g = c.golden
    # eight distinct points of each group: (i + 2) * generator
    g1, g2 = c.g1(g["g1_smul"][0]["out"]), c.g2(g["g2_smul"][0]["out"])
    P, Q = np.zeros((8, 2 * c.nq), dtype=np.uint64), np.zeros((8, 4 * c.nq), dtype=np.uint64)
    for i in range(8):
        getattr(twin, "twin_g1_smul_" + name)(ptr(g1), ptr(c.fr(i + 2)), ptr(P[i]))
        getattr(twin, "twin_g2_smul_" + name)(ptr(g2), ptr(c.fr(i + 2)), ptr(Q[i]))
    P, Q = P.reshape(-1), Q.reshape(-1)
    rng = np.random.default_rng(20241220)

    def scal(n):
        return np.concatenate([c.fr(int.from_bytes(rng.bytes(40), "little") % c.r) for _ in range(n)])

    def avg(op, nt, reps=8):
        return sum(f(op, nt, ptr(P), ptr(Q), ptr(scal(8))) for _ in range(reps)) / reps

    def table():
      m = [avg(10, k, 1) for k in (1, 2, 3)]
      t = [avg(14, k, 1) for k in (1, 2, 3)]
      mf = [avg(15, k, 1) for k in (1, 3)]
      tf = [avg(16, k, 1) for k in (1, 3)]
      return {
          "g1_smul": avg(0, 1), "g2_smul": avg(1, 1),
          "g1_straus4_per_term": avg(2, 4) / 4, "g1_straus8_per_term": avg(2, 8) / 8,
          "g2_straus4_per_term": avg(3, 4) / 4, "g2_straus8_per_term": avg(3, 8) / 8,
          # lanes that serve several outputs from one table build, 4- or 5-bit windows: per term and output
          **{"%s_straus%d%s%s_per_term" % (g, cap, "w5" if w == 5 else "", "x%d" % mo if mo > 1 else ""):
             avg(op, cap | (w << 8) | (mo << 12)) / (cap * mo)
             for g, op in (("g1", 17), ("g2", 18)) for cap in (4, 8) for w in (4, 5) for mo in (1, 2, 4)
             if not (w == 4 and mo == 1)},
          "g1_madd": avg(4, 2, 1), "g2_madd": avg(5, 2, 1), "g1_add": avg(6, 2, 1), "g2_add": avg(7, 2, 1),
          "g1_red_tail": avg(8, 2, 1), "g2_red_tail": avg(9, 2, 1),
          "miller_per_lane": m[0] - (m[2] - m[0]) / 2, "miller_per_pair": (m[2] - m[0]) / 2,
          "miller2_per_lane": t[0] - (t[2] - t[0]) / 2, "miller2_per_triple": (t[2] - t[0]) / 2,
          "miller_per_fixed_pair": (mf[1] - mf[0]) / 2, "miller2_per_fixed_triple": (tf[1] - tf[0]) / 2,
          "f12_mul": avg(11, 1, 1), "final_exp": avg(12, 1, 1), "final_exp_coop_lane": avg(13, 1, 1) / 3,
      }

    state = rng.bit_generator.state
    out[name] = table()
    L, N32 = (14, 12) if name == "bls12_381" else (10, 8)
    out[name]["mads_per_fq_mul"] = 2 * L * L
    out[name]["min_mads_per_fq_mul"] = 2 * N32 * N32  # a saturated 32-bit-limb Montgomery product
    rng.bit_generator.state = state  # the same scalars for the second pass
    MADS[0] = True
    out[name]["executed_mads"] = table()
path = os.path.join(ROOT, "profiles", "r4", "fq_mul_counts.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
