"""Kernels and copies of the LAST host-pointer prove + verify in a rocprofv3 --kernel-trace --memory-copy-trace
directory, on one time axis (ms from the first event of the prove call)."""
import csv
import glob
import sys

d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "k_seg<" in name:
            body = name.split("k_seg<gs::")[1]
            name = body.split("<")[0]
            if name in ("k_var_multi", "k_var_tab", "k_miller", "k_miller_pair"):  # the instance matters: keep its arguments
                name = body.replace("gs::", "").replace("Bls12_381", "B").replace("Bn254", "N").split(", unsigned long")[0][:48]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + name[:40]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %.1f MB" % (r.get("Direction", "?").replace("MEMORY_COPY_", ""), int(r.get("Bytes", r.get("Size", 0)) or 0) / 1e6)))
ev.sort()
# the last k_prep_prove starts the last call pair
starts = [i for i, e in enumerate(ev) if "k_prep_prove" in e[2]]
# optional second argument: which k_prep_prove launch opens the listing (-1 = the last one, the default; -2 = the one
# before it: under bench.py the last step is the per-kernel profiling step, which does not merge a mixed call's parts)
i0 = starts[int(sys.argv[2]) if len(sys.argv) > 2 else -1]
# copies that precede it belong to the call too: walk back over copies within 20 ms
while i0 > 0 and ev[i0 - 1][2].startswith("C") and ev[i0][0] - ev[i0 - 1][0] < 20e6:
    i0 -= 1
t0 = ev[i0][0]
for s, e, n in ev[i0:]:
    print("%9.2f %9.2f  %8.2f  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
