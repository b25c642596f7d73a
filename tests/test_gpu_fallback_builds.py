"""The FALLBACK builds the Makefile can produce are compared with the oracle too.

csrc/Makefile ships two code-generation workarounds: the Miller / final-exponentiation bodies are taken inline only
when the compiler accepts -mllvm -amdgpu-long-branch-factor=0 (otherwise the out-of-line form is built: INLINE=), and
the multiplier is reached from fixed-register asm statements unless -DGS_NO_ASM_CALL selects the C++-call form.  A box
with another ROCm would ship one of those variants, so each is built next to the product library
(tools/build_variant.sh -> groth_sahai_rs_amd/lib/var/<name>.so) and must pass a forced-shape oracle batch -- every
equation bit-exact, verdicts exact -- on both curves.  The variant is loaded in a child interpreter through GS_AMD_LIB
(capi.py honours it; it is still the HIP library, never a fallback to the CPU)."""
import os
import subprocess
import sys

import pytest

from gsutil import REPO

pytestmark = pytest.mark.gpu

VARIANTS = {
    "outofline": 'GS_MAKE_ARGS="INLINE=" tools/build_variant.sh outofline',
    "noasmcall": "tools/build_variant.sh noasmcall -DGS_NO_ASM_CALL",
}
CHILD = r"""
import sys
sys.path.insert(0, %(tests)r)
from gpubatch import run_batch
from test_gpu_variants import SHAPES, expected_kernels
for cid, cname in ((0, "bls12_381"), (1, "bn254")):
    for ty, shape in ((0, "twin6_straus8x2w5_lane"), (2, "single9_straus4w5_coop"), (1, "pair12_straus8x2w5_lane")):
        o = SHAPES[shape]
        run_batch(cid, cname, ty, 66, 4, 4, range(66), opts=o, expect=expected_kernels(ty, 4, 4, o), seed=9700 + ty)
print("variant ok")
"""


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_fallback_build_matches_oracle(name):
    lib = os.path.join(REPO, "groth_sahai_rs_amd", "lib", "var", name + ".so")
    if not os.path.exists(lib):
        pytest.skip("variant library not built: run `%s` (5 minutes of hipcc)" % VARIANTS[name])
    env = dict(os.environ, GS_AMD_LIB=lib)
    r = subprocess.run([sys.executable, "-c", CHILD % dict(tests=os.path.join(REPO, "tests"))], env=env, cwd=REPO,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "variant ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
