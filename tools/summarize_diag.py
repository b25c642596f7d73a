"""Reduce the counter CSVs of tools/pmc_diag.sh to per-kernel means:  python tools/summarize_diag.py <dir> > diag.json
Every counter is averaged over the launches of a kernel; durations come from the kernel trace of the same pass (the
GRBM pass gives clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / duration, MI355X_MICROARCH.md "DVFS give-back")."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0]
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and row.get("Start_Timestamp"):
                    acc[name]["_dur_ns_grbm"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    # durations of the GRBM pass from its kernel trace when the counter rows carry no timestamps
    for f in glob.glob(root + "/grbm/**/*kernel_trace.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0]
                if "_dur_ns_grbm" not in acc[name] or len(acc[name]["_dur_ns_trace"]) < 10 ** 9:
                    acc[name]["_dur_ns_trace"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    out = {}
    for k, cs in acc.items():
        if not k.startswith("void gs::"):
            continue
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["launches"] = max(len(v) for v in cs.values())
        wc = e.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA"):
                if c in e:
                    e[c + "_frac"] = e[c] / wc
        dur = e.get("_dur_ns_grbm") or e.get("_dur_ns_trace")
        if e.get("GRBM_GUI_ACTIVE") and dur:
            e["clock_ghz"] = e["GRBM_GUI_ACTIVE"] / 8.0 / dur
            e["dur_ms_grbm_pass"] = dur / 1e6
        if e.get("SQC_ICACHE_REQ"):
            e["icache_miss_frac"] = e.get("SQC_ICACHE_MISSES", 0.0) / e["SQC_ICACHE_REQ"]
        out[k] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
