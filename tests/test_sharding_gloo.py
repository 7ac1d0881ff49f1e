"""The N>1 path on CPU: world_size-2 gloo run of the scatter -> per-rank compute -> gather
driver (babyjubjub-rs_amd/shard.py) and of the pre-sharded partition bench.py uses.  The
per-rank compute is the ORACLE here (tests may use it); on GPUs it is libbjj_hip.so."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, n, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from conftest import Oracle
        from babyjubjub_rs_amd import shard, workload
        orc = Oracle()
        orc.threads = 2
        dev = torch.device("cpu")
        full = torch.from_numpy(workload.scalars_254(n).reshape(-1)) if rank == 0 else None

        def compute(sc):
            return torch.from_numpy(orc.mul_fixed_base(sc.numpy()).reshape(-1))

        res = shard.scatter_compute_gather([full], [32], n, compute, 64, dev)
        # pre-sharded mode: each rank generates and processes its own block
        lo, hi = workload.shard_bounds(n, ws, rank)
        mine = orc.mul_fixed_base(workload.scalars_254(hi - lo, offset=lo))
        if rank == 0:
            want = orc.mul_fixed_base(workload.scalars_254(n))
            q.put(("gather_ok", bool((res.numpy().reshape(-1, 64) == want).all())))
            q.put(("shard0_ok", bool((mine == want[lo:hi]).all())))
        else:
            want = orc.mul_fixed_base(workload.scalars_254(n))
            q.put(("shard%d_ok" % rank, bool((mine == want[lo:hi]).all())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_world(ws, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, n, q)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    return dict(q.get(timeout=5) for _ in range(ws + 1))


def test_scatter_compute_gather_world2():
    # ragged: 37 is not a multiple of the world size -- exact block sizes travel, nothing is padded
    assert _run_world(2, 37) == {"gather_ok": True, "shard0_ok": True, "shard1_ok": True}


def test_scatter_compute_gather_world3_with_an_empty_rank():
    # n = 2 over 3 ranks: blocks [0,1) [1,2) and an EMPTY block for rank 2 (no message is posted for it)
    assert _run_world(3, 2) == {"gather_ok": True, "shard0_ok": True, "shard1_ok": True, "shard2_ok": True}
    assert _run_world(3, 36) == {"gather_ok": True, "shard0_ok": True, "shard1_ok": True, "shard2_ok": True}
