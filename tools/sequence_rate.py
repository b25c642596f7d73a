"""A SEQUENCE of bench configurations in ONE process (the Bench.run of bench.py): what profiles/r3/scratch_pool.txt
was measured with -- a large batch must run at its stand-alone time whatever ran before it in the process.
    python tools/sequence_rate.py m12 p12 m16 p16     (p = PPE, m = mixed 50/25/25; 12 / 16 = log2 equations;
                                                       a trailing r = with the per-kernel profile pass)"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import bench
class A: pass
b = bench.Bench(A(), None, 0, 1, 0, "cuda:0", "cuda:0")
seq = sys.argv[1:]
for s in seq:
    kw = dict(p10=dict(log2n=10), m10=dict(log2n=10, mixed=True), p8=dict(log2n=8), p14=dict(log2n=14), p12=dict(log2n=12), m12=dict(log2n=12, mixed=True), p16=dict(log2n=16), m16=dict(log2n=16, mixed=True), p12r=dict(log2n=12, roofline=True), m12r=dict(log2n=12, mixed=True, roofline=True))[s]
    r = b.run(steps=3, warmup=1, seed_off=3, **kw)
    print(s, round(r["value"]), round(r["ms_per_step"], 1), flush=True)
