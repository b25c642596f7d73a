#!/usr/bin/env python3
"""Where the cycles of the G1 Straus lanes go (diagnosis build: tools/build_variant.sh stamps -DGS_DEBUG_STAMPS).

    GS_AMD_LIB=groth_sahai_rs_amd/lib/var/stamps.so python3 tools/straus_phases.py [log2n]

Runs the verifier of a 2^log2n PPE batch (its Gamma^T c is G1 Straus lanes only: k_var_multi4w5x4.vg1) and prints the
per-wave shader cycles of the phases the lanes stamped."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd.workload import Workload

    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    eng = gs.Engine(0, 0)
    wl = Workload(eng, ty=0, N=1 << log2n, m=4, n=4, seed=11, corrupt_every=0)
    wl.prove()
    eng.sync()
    buf = (ctypes.c_ulonglong * 16)()
    eng.lib.gs_debug_stamps(eng.ctx, buf)  # clear what the prover's G1 lanes stamped
    wl.verify()
    eng.sync()
    eng.lib.gs_debug_stamps(eng.ctx, buf)
    v = list(buf)
    waves = max(v[4], 1)
    names = ["digits + top", "main loop", "  doubling calls", "  addition steps", "waves", "table build"]
    for i, nm in enumerate(names):
        print("%-18s %14d  per wave %12.0f" % (nm, v[i], v[i] / waves))
    print("(verify of 2^%d PPE 4x4: every G1 Straus lane of it; one output per run: divide by the outputs per lane)" % log2n)


if __name__ == "__main__":
    main()
