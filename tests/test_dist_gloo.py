"""N > 1 path on CPU: world_size-2 gloo run of the sharding / verdict / accumulator
plumbing in groth_sahai_rs_amd/dist.py (the GPU run uses the same code over RCCL)."""
import os
import socket
import subprocess
import sys
import textwrap

from gsutil import REPO


def test_shard_range_partitions():
    from groth_sahai_rs_amd.dist import shard_range

    for n in (0, 1, 7, 4096, 262144):
        for w in (1, 2, 3, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent(
    """
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from groth_sahai_rs_amd.dist import shard_range, allreduce_failures, allgather_accumulators
    dist.init_process_group("gloo")
    r, w = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(1000, r, w)
    # each rank "verifies" its block; equation 777 is bad
    failed = sum(1 for i in range(lo, hi) if i == 777)
    total = allreduce_failures(failed)
    assert total == 1, total
    # the bench passes the count as a tensor (no host round trip before the collective)
    assert allreduce_failures(torch.tensor(failed), device="cpu") == 1
    acc = torch.full((1152,), r + 1, dtype=torch.uint8)   # stand-in for the 2 x GT accumulator bytes
    outs = allgather_accumulators(acc)
    assert [int(o[0]) for o in outs] == list(range(1, w + 1))   # rank order on every rank
    dist.barrier()
    dist.destroy_process_group()
    print("rank", r, "ok")
    """
)


def test_gloo_world_size_2(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % REPO)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2
