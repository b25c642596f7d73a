#!/usr/bin/env python3
"""Headline benchmark: Groth-Sahai proofs+verifies per second on a BLS12-381
pairing-product-equation batch (BASELINE.json metric; configs[1] = 2^12
independent PPEs with 4 G1 + 4 G2 variables each, one MI355X).

  python bench.py --gpus N --steps K --warmup W          (N = 1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = commit_and_prove + verify of every equation of the per-GPU batch
(inputs resident in HBM).  N > 1: equations are independent, each rank owns its
own 2^12 batch (weak scaling, no data-path collective in exact mode); ranks
barrier, time the same K steps, rank 0 reports MAX-over-ranks time.

Extra objects on the JSON line:
  roofline      dominant kernel, HIP-event timed through the library's own hook
  cpu_baseline  oracle/gs_ref.c (CPU restatement of the reference path, NOT
                arkworks) timed on a bounded sample on the host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def host_cores():
    """CPUs this process may actually use: hardware threads, capped by the affinity mask and by the cgroup CPU
    quota (the one-GPU box shows 256 hardware threads but grants 16 CPUs of time)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n, os.cpu_count() or 1


def cpu_baseline(sample_units, threads):
    """Time the C restatement of the reference path on `sample_units` PPE 4x4 units (all granted cores), and on two
    units with one thread (the reference's own single-equation bench shape, BASELINE configs[0])."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import gs_ref_py
    except Exception as ex:  # oracle not built
        return {"value": None, "unit": "proofs+verifies/s", "cores": 0, "kind": "port", "sample": "unavailable: %s" % ex}
    t1, u1, ok1 = gs_ref_py.bench_ppe(2, 4, 4, 1)
    t, units, ok = gs_ref_py.bench_ppe(sample_units, 4, 4, threads)
    return {
        "value": units / t,
        "unit": "proofs+verifies/s",
        "cores": threads,
        "kind": "port",
        "single_thread": u1 / t1,
        "hardware_threads_visible": os.cpu_count(),
        "sample": "%d PPE 4x4 commit_and_prove+verify units, reference evaluation order (5 pairing_sums, "
        "per-op normalisation), %d threads over equations (= the CPUs granted to this process), all verified=%s, "
        "%.1f s; single thread: %.2f units/s" % (units, threads, ok and ok1, t, u1 / t1),
    }


# measured v_mad_u64_u32 issue peak of the chip (tools/ubench.hip, profiles/r1/ubench_valu.txt: 4.34 cycles per
# wave-instruction per SIMD at the 2.4 GHz the runtime reports, 1024 SIMDs x 64 lanes) in G lane-mads/s
VALU_MAD_PEAK_G = 1024 * 64 * 2.4e9 / 4.34 / 1e9


def alu_roofline(work, ms, curve_id, dominant):
    """Useful Fq multiplications of one profiled step (per-kernel work items from gs_prof_get_work x the
    per-primitive counts of profiles/r1/fq_mul_counts.json, tools/count_fq_muls.py) x 2 L^2 multiply-adds each,
    against the measured v_mad_u64_u32 peak: SURVEY.md 8(d)'s ALU roofline.  Kernels without an entry (scalar
    preparation, GT products, boundary conversions) count as zero, so the figure is a lower bound."""
    try:
        cnt = json.load(open(os.path.join(ROOT, "profiles", "r1", "fq_mul_counts.json")))[
            "bls12_381" if curve_id == 0 else "bn254"]
    except Exception as ex:  # the counts are a committed measurement artefact; without them report nothing
        return {"error": "fq_mul_counts.json unavailable: %s" % ex}
    muls = {}
    partials = cells = 0
    for name, (lanes, items) in work.items():
        g = "g2" if name.endswith("g2") else "g1"
        k = name.split(".")[0]
        if k == "k_fix":
            m = items * 16 * (65535.0 / 65536.0) * cnt[g + "_madd"]  # 16-bit windows
        elif k == "k_var":
            m = lanes * cnt[g + "_smul"]
        elif k in ("k_var_multi4", "k_var_multi8"):
            m = items * cnt["%s_straus%s_per_term" % (g, k[-1])]
        elif k == "k_red":
            m = max(items - 2 * lanes, 0) * cnt[g + "_add"] + lanes * cnt[g + "_red_tail"]
        elif name == "k_miller.twin":
            m = lanes * cnt["miller2_per_lane"] + items * cnt["miller2_per_triple"]
            partials += 2 * lanes
        elif k == "k_miller":
            m = lanes * cnt["miller_per_lane"] + items * cnt["miller_per_pair"]
            partials += lanes
        elif name == "k_final":
            m = lanes * cnt["final_exp"]
            cells += lanes
        elif name == "k_final.coop":
            m = lanes * cnt["final_exp_coop_lane"]
            cells += lanes // 3
        else:
            continue
        muls[name] = m
    for name in ("k_final", "k_final.coop"):  # products of the Miller partials of each cell
        if name in muls and partials > cells:
            muls[name] += (partials - cells) * cnt["f12_mul"] * (3 if name.endswith("coop") else 1)
    total = sum(muls.values())
    step_s = sum(ms.values()) / 1e3
    mads = cnt["mads_per_fq_mul"]
    dom = {}
    if dominant in muls and ms.get(dominant):
        a = muls[dominant] * mads / (ms[dominant] / 1e3) / 1e9
        dom = {"kernel": dominant, "achieved": a, "frac": a / VALU_MAD_PEAK_G}
    a = total * mads / step_s / 1e9
    return {"bound": "valu (v_mad_u64_u32 issue)", "achieved": a, "peak": VALU_MAD_PEAK_G, "unit": "G mad/s",
            "frac": a / VALU_MAD_PEAK_G, "fq_muls_per_step": total, "mads_per_fq_mul": mads, "dominant": dom,
            "fq_muls_by_kernel": {k: round(v) for k, v in muls.items()}}



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2n", type=int, default=12, help="equations per GPU (2^k); configs[1] = 12")
    ap.add_argument("--m", type=int, default=4)
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--curve", type=int, default=0)
    ap.add_argument("--type", type=int, default=0)
    ap.add_argument("--mode", choices=["exact", "rlc"], default="exact",
                    help="verifier: exact = reference semantics (bool per equation); rlc = batched pairing-product check")
    ap.add_argument("--mixed", action="store_true", help="configs[2]: 50%% PPE, 25%% MSMEG1, 25%% MSMEG2")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="CPU-baseline sample units (0 = auto, ~20 s)")
    args = ap.parse_args()

    import torch
    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd.workload import Workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # GS_BENCH_BACKEND=gloo is a REHEARSAL switch for a one-GPU box (ranks share cuda:0, collectives on the
        # CPU): it exercises the launch contract and the rank logic, its numbers mean nothing
        backend = os.environ.get("GS_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    coll_dev = dev if (dist is None or dist.get_backend() == "nccl") else "cpu"
    N = 1 << args.log2n

    eng = gs.Engine(args.curve, local)
    if args.mixed:
        # one CRS, three sub-batches (the engine runs one type/shape per call)
        wls = [Workload(eng, ty=0, N=N // 2, m=args.m, n=args.n, seed=20241220 + 2 + rank, device=dev)]
        crs = wls[0].crs
        for ty in (1, 2):
            wls.append(Workload(eng, ty=ty, N=N // 4, m=args.m, n=args.n, seed=20241220 + 2 + rank, device=dev))
            assert (wls[-1].crs == crs).all()
    else:
        wls = [Workload(eng, ty=args.type, N=N, m=args.m, n=args.n, seed=20241220 + 1 + rank, device=dev)]
    wl = wls[0]
    from groth_sahai_rs_amd.dist import allgather_accumulators

    def one_step(collective=True):
        for w in wls:
            w.prove()
        if args.mode == "exact":
            for w in wls:
                w.verify()
            return None
        accs = [w.verify_rlc() for w in wls]
        allacc = []
        for a in accs:  # cross-GPU product of GT accumulators: all-gather (RCCL) + fixed-order local product
            allacc += allgather_accumulators(a.to(coll_dev)) if collective else [a]
        pairs = torch.cat(allacc).cpu().numpy()
        return eng.gt_finalize(pairs)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        verdict = one_step()
        if args.mode == "rlc":
            assert verdict == 1, "batched verifier rejected a valid batch"
    eng.sync()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- correctness of what was timed: all valid proofs accepted, corrupted ones rejected
    for w in wls:
        w.prove()
        bad = set(w.corrupt())
        w.verify()
        eng.sync()
        ok = w.ok.cpu().numpy()
        expect = [0 if i in bad else 1 for i in range(w.N)]
        assert ok.tolist() == expect, "verification verdicts wrong on the benchmark batch"
        if args.mode == "rlc" and bad:
            assert eng.gt_finalize(w.verify_rlc().cpu().numpy()) == 0, "batched verifier accepted a corrupted batch"

    # ---- roofline leg: per-kernel HIP-event timing of one more step (rank 0)
    roof = None
    if rank == 0:
        eng.prof_enable(True)
        eng.prof_reset()
        one_step(collective=False)  # rank 0 alone: no collective inside the profiled step
        eng.sync()
        prof = eng.prof_get()
        eng.prof_enable(False)
        tot = sum(p[1] for p in prof) or 1.0
        name, ms, launches = max(prof, key=lambda p: p[1])
        avg_s = ms / max(launches, 1) / 1e3
        bpu = sum(w.bytes_per_unit() * w.N for w in wls) / N
        achieved = N * bpu / avg_s / 1e9
        traffic = None
        try:  # HBM bytes per launch from the committed PMC collection (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
            # FETCH_SIZE doubled per the gfx950 correction), valid for the default 2^12 PPE workload only
            if args.log2n == 12 and not args.mixed and args.curve == 0 and args.type == 0 and args.mode == "exact":
                tj = json.load(open(os.path.join(ROOT, "profiles", "r1", "traffic_2p12.json")))
                for k, v in tj.items():
                    if name.split(".")[0] in k and ("true" in k) == name.endswith(".twin"):
                        traffic = v["hbm_bytes_per_launch_corrected"]
        except Exception:
            traffic = None
        alu = alu_roofline(eng.prof_get_work(), {p[0]: p[1] for p in prof}, args.curve, name)
        if "fq_muls_per_step" in alu:
            alu["fq_muls_per_unit"] = alu["fq_muls_per_step"] / N
        roof = {
            "bound": "hbm",
            "kernel": name,
            "achieved": achieved,
            "peak": 8000.0,
            "unit": "GB/s",
            "frac": achieved / 8000.0,
            "traffic": traffic,
            "avg_kernel_ms": ms / max(launches, 1),
            "bytes_per_unit": bpu,
            "kernel_share_of_step": ms / tot,
            "kernels_ms": {p[0]: round(p[1], 3) for p in prof},
            "note": "integer-ALU bound path (SURVEY.md 8d): HBM fraction is ~1e-4 by construction; see alu",
            "alu": alu,
        }

    total_units = N * world * args.steps
    res = {
        "metric": "GS proofs+verifies/sec (BLS12-381 PPE batch)" if args.curve == 0 and args.type == 0 else
        "GS proofs+verifies/sec (curve %d type %d)" % (args.curve, args.type),
        "value": total_units / dt,
        "unit": "proofs+verifies/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 limbs (381-bit Montgomery)" if args.curve == 0 else "u32 limbs (254-bit Montgomery)",
        "data": "synthetic",
        "config": {"workload": "2^%d independent %s equations per GPU, m=%d n=%d, commit_and_prove+verify(%s)"
                   % (args.log2n, "mixed 50%% PPE/25%% MSMEG1/25%% MSMEG2" if args.mixed else
                      ["PPE", "MSMEG1", "MSMEG2", "QuadEqu"][args.type], args.m, args.n, args.mode),
                   "curve": "BLS12-381" if args.curve == 0 else "BN254", "equations_per_gpu": N,
                   "parallelism": "equation-sharded x%d" % world},
    }
    if rank == 0:
        res["roofline"] = roof
        if not args.no_cpu and world == 1:  # the CPU baseline is a rank-0, N = 1 measurement
            threads, _ = host_cores()
            sample = args.cpu_sample or max(128 * threads, 256)   # ~10-30 s of CPU work on the granted host cores
            res["cpu_baseline"] = cpu_baseline(sample, threads)
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
