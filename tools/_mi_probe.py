import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import groth_sahai_rs_amd as gs
from groth_sahai_rs_amd.workload import Workload
os.makedirs("gpurun_out/probe", exist_ok=True)
log = open("gpurun_out/probe/probe.txt", "a")
def say(*a):
    print(*a, file=log, flush=True); print(*a, flush=True)
t0 = time.time()
eng = gs.Engine(0, 0)
say("engine", time.time() - t0)
wl = Workload(eng, ty=0, N=int(sys.argv[1]), m=4, n=4, seed=5, corrupt_every=0)
say("workload", time.time() - t0)
eng.prof_enable(True); eng.prof_reset()
wl.prove(); eng.sync(); say("prove", time.time() - t0)
wl.verify(); eng.sync(); say("verify", time.time() - t0)
say(wl.ok.cpu().numpy().all(), {p[0]: round(p[1], 2) for p in eng.prof_get()})
