#!/usr/bin/env python3
"""Where the cycles of k_final go (diagnosis build: tools/build_variant.sh stamps -DGS_DEBUG_STAMPS).

    GS_AMD_LIB=groth_sahai_rs_amd/lib/var/stamps.so python3 tools/final_phases.py [log2n]

Runs the verifier of a 2^log2n PPE batch and prints, per wave of k_final (one lane = one cell's final exponentiation),
the shader cycles lane 0 stamped around the phases."""
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
    wl.verify()
    eng.sync()
    eng.lib.gs_debug_stamps(eng.ctx, buf)  # warm (and clear)
    wl.verify()
    eng.sync()
    eng.lib.gs_debug_stamps(eng.ctx, buf)
    v = list(buf)
    waves = max(v[9], 1)
    rows = [(6, "cell product"), (7, "final exponentiation"), (13, "  easy part"), (12, "  x-powers (5)"),
            (10, "    compressed squarings"), (11, "    prefix + shared inversion"), (8, "compare + store")]
    print("waves %d" % waves)
    for i, nm in rows:
        print("%-32s %16d  per wave %12.0f" % (nm, v[i], v[i] / waves))
    tot = v[6] + v[7] + v[8]
    print("%-32s %16d  per wave %12.0f" % ("lane total", tot, tot / waves))


if __name__ == "__main__":
    main()
