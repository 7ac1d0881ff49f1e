"""Multi-GPU driver for the batch path: one process per GPU (torch.distributed; backend
"nccl" is RCCL on ROCm, "gloo" on CPU for tests).

The path shards embarrassingly: every item reads only its own record, so the
data path has NO collective.  Two modes:
  * pre-sharded (what bench.py times): each rank owns a contiguous block of the batch
  * scatter/gather (BASELINE.json cfg 5): rank 0 holds the whole batch; inputs are scattered,
    each rank runs the kernel on its shard, results are gathered back to rank 0

`compute(shard_arrays...) -> result array` is supplied by the caller (the GPU library in
production; tests pass the oracle so the plumbing is checked on CPU).
"""
import torch
import torch.distributed as dist

from .workload import shard_bounds


def scatter_rows(full, n, row_bytes, device, src=0):
    """Rank `src` holds `full` (uint8 tensor of n*row_bytes); returns this rank's shard.
    Shards are padded to equal length for the collective and trimmed afterwards."""
    ws, rank = dist.get_world_size(), dist.get_rank()
    per = (n + ws - 1) // ws
    out = torch.empty(per * row_bytes, dtype=torch.uint8, device=device)
    chunks = None
    if rank == src:
        padded = torch.zeros(ws * per * row_bytes, dtype=torch.uint8, device=device)
        padded[: n * row_bytes] = full.reshape(-1)[: n * row_bytes]
        chunks = list(padded.chunk(ws))
    dist.scatter(out, chunks, src=src)
    lo, hi = shard_bounds(n, ws, rank)
    return out[: (hi - lo) * row_bytes]


def gather_rows(shard, n, row_bytes, device, dst=0):
    """Inverse of scatter_rows: returns the n*row_bytes result on rank `dst` (None elsewhere)."""
    ws, rank = dist.get_world_size(), dist.get_rank()
    per = (n + ws - 1) // ws
    padded = torch.zeros(per * row_bytes, dtype=torch.uint8, device=device)
    padded[: shard.numel()] = shard.reshape(-1)
    bufs = [torch.empty_like(padded) for _ in range(ws)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat(bufs)[: n * row_bytes]


def scatter_compute_gather(inputs, row_bytes_in, n, compute, row_bytes_out, device, src=0):
    """inputs: list of full uint8 tensors on rank `src` (ignored elsewhere)."""
    shards = [scatter_rows(inputs[i] if dist.get_rank() == src else None, n, rb, device, src)
              for i, rb in enumerate(row_bytes_in)]
    res = compute(*shards)
    return gather_rows(res, n, row_bytes_out, device, src)
