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
        # the pipelined schedule: peer blocks in pieces (here of >= 64 items so that small batches are really cut up)
        calls = []

        def compute_piece(arrays, count, out_view, k):
            calls.append((k, count))
            out_view.copy_(torch.from_numpy(orc.mul_fixed_base(arrays[0].numpy()).reshape(-1)))

        res_p = shard.scatter_compute_gather_pipelined([full], [32], n, compute_piece, 64, dev, pieces=4, min_piece=64)
        lo_, hi_ = workload.shard_bounds(n, ws, rank)
        want_calls = [(0, hi_ - lo_)] if rank == 0 else [(k, b - a) for k, (a, b) in enumerate(workload.piece_bounds(hi_ - lo_, 4, 64))]
        assert calls == [c for c in want_calls if c[1] > 0], (rank, calls, want_calls)
        # pre-sharded mode: each rank generates and processes its own block
        lo, hi = workload.shard_bounds(n, ws, rank)
        mine = orc.mul_fixed_base(workload.scalars_254(hi - lo, offset=lo))
        if rank == 0:
            want = orc.mul_fixed_base(workload.scalars_254(n))
            q.put(("gather_ok", bool((res.numpy().reshape(-1, 64) == want).all()) and bool((res_p.numpy().reshape(-1, 64) == want).all())))
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


def test_pipelined_schedule_cuts_peer_blocks_into_pieces():
    # 1 000 items over 2 ranks: the peer's 500-item block travels as 128 + 128 + 128 + 116; 3 ranks, ragged: 334 / 333 / 333
    assert _run_world(2, 1000) == {"gather_ok": True, "shard0_ok": True, "shard1_ok": True}
    assert _run_world(3, 1000) == {"gather_ok": True, "shard0_ok": True, "shard1_ok": True, "shard2_ok": True}


def test_piece_geometry_matches_the_library():
    from babyjubjub_rs_amd import workload
    assert workload.piece_bounds(500, 4, 64) == [(0, 128), (128, 256), (256, 384), (384, 500)]
    assert workload.piece_bounds(1 << 21, 4) == [(i << 19, (i + 1) << 19) for i in range(4)]        # cfg 5: 2^21 per device
    assert workload.piece_bounds(40000, 4) == [(0, 40000)] and workload.piece_bounds(0) == []
    assert workload.piece_bounds(70000, 16) == [(0, 35008), (35008, 70000)]
    assert workload.piece_bounds(643, 4, 64) == [(0, 192), (192, 384), (384, 576), (576, 643)]


def test_scatter_compute_gather_world8_the_node_size_of_the_scaling_bench():
    """G = 8 (what `bench.py --gpus 8` and BASELINE configs[4] use): ragged blocks (1003 = 7 x 126 + 121), pieces per peer block,
    and n < G (5 items: three ranks own nothing and neither send nor receive)"""
    ok = {"gather_ok": True}
    ok.update({"shard%d_ok" % r: True for r in range(8)})
    assert _run_world(8, 1003) == ok
    assert _run_world(8, 5) == ok
