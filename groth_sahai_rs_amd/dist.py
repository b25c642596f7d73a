"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Equations are independent given the shared CRS, so the
batch is partitioned by equation index into contiguous blocks and the prover
needs NO collective.  Verifier: exact mode combines per-rank verdicts with one
tiny all-reduce (MIN of "all ok" / SUM of failure counts); batched (RLC) mode
all-gathers the per-rank GT accumulators (2 x GT bytes each) and every rank
multiplies them in rank order before the single final exponentiation -- a
product in Fp12 is not an RCCL reduction operator, hence gather + local product.
"""


def shard_range(n_total, rank, world):
    """Contiguous block [lo, hi) of equation indices owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def allreduce_failures(n_failed_local, device=None):
    """Total number of rejected proofs over all ranks (0 => everything verified): the rank-combined verdict of the
    exact verifier.  `n_failed_local` is an int or a 0-d tensor (e.g. ``(ok == 0).sum()`` still on the GPU: no host
    round trip before the collective); `device` is where the collective runs (the GPU for RCCL, "cpu" for gloo)."""
    import torch
    import torch.distributed as dist

    if torch.is_tensor(n_failed_local):
        t = n_failed_local.to(dtype=torch.int64, device=device).reshape(1)
    else:
        t = torch.tensor([int(n_failed_local)], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def allgather_accumulators(acc_bytes):
    """acc_bytes: uint8 tensor (the rank's GT accumulator pair).  Returns the list
    of all ranks' accumulators in RANK ORDER (deterministic product order)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [acc_bytes]
    outs = [torch.empty_like(acc_bytes) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, acc_bytes)
    return outs
