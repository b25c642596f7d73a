"""Reduce rocprofv3 counter CSVs (separate --pmc passes, as MI355X_MICROARCH.md prescribes)
to per-kernel, per-launch figures:  python tools/summarize_pmc.py <fetch_dir> <write_dir> <sq_dir> > traffic.json
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so it is
doubled (guide, HBM / rocprofv3 section).  Kernel names are cut at the template argument list's end."""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0]
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    fetch, write, sq = (load(d) for d in sys.argv[1:4])
    out = {}
    for k in fetch:
        if not k.startswith("void gs::"):
            continue
        mean = lambda a, c: (sum(a[k][c]) / len(a[k][c])) if a[k].get(c) else None
        fk, wk = mean(fetch, "FETCH_SIZE"), mean(write, "WRITE_SIZE")
        e = {"launches": len(fetch[k]["FETCH_SIZE"]), "fetch_kib_per_launch": fk, "write_kib_per_launch": wk,
             "hbm_bytes_per_launch_corrected": (2 * fk + (wk or 0)) * 1024}
        busy, wait, valu, waves = (mean(sq, c) for c in ("SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_WAVES"))
        wc = mean(sq, "SQ_WAVE_CYCLES")
        if wait is not None and wc:
            e["wait_any_frac"] = wait / wc
        if valu is not None and waves:
            e["valu_insts_per_wave"] = valu / waves
        out[k] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
