"""Multi-process sharding of the batch path: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm,
"gloo" on CPU for tests).  The single-process, all-devices form is native: bjj_multi_* in libbjj_hip.so.

The path shards embarrassingly: every item reads only its own record, so the data path has NO
collective.  Two modes:
  * pre-sharded (what bench.py's headline times): each rank owns a contiguous block of the batch
  * scatter/gather (BASELINE.json cfg 5): rank 0 holds the whole batch; the input blocks are sent out, each rank runs
    the kernels on its block, the results come back to rank 0

Scatter / gather move EXACT block sizes (contiguous ceil(n/G) blocks, `workload.shard_bounds`) as point-to-point
operations posted in ONE batch (`dist.batch_isend_irecv` = one ncclGroupStart/End on RCCL) for all arrays and all
peers: no padded copy of the batch, no per-array collective, and the root's own block is a view of its input.

`compute(shard_arrays...) -> result array` is supplied by the caller (the GPU library in production; tests pass the
oracle so the plumbing is checked on CPU).
"""
import torch
import torch.distributed as dist

from .workload import shard_bounds


def _post(ops_spec):
    """ops_spec: list of (isend | irecv, tensor, peer).  Posts them as ONE batch and waits.  The gloo backend (CPU tests,
    the single-GPU developer mode of bench.py) cannot move device tensors: they are staged through host copies there."""
    if not ops_spec:
        return
    stage = dist.get_backend() == "gloo"
    ops, copies = [], []
    for fn, t, peer in ops_spec:
        if stage and t.is_cuda:
            h = t.cpu() if fn is dist.isend else torch.empty(t.shape, dtype=t.dtype)
            if fn is dist.irecv:
                copies.append((t, h))
            t = h
        ops.append(dist.P2POp(fn, t, peer))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for dst, h in copies:
        dst.copy_(h)


def scatter_arrays(fulls, n, row_bytes, device, src=0, recv=None):
    """fulls: list of uint8 tensors of n*row_bytes[i] bytes on rank `src` (ignored elsewhere).  Returns this rank's
    blocks, one tensor per array (views of `fulls` on the root).  recv: optional preallocated per-array receive buffers
    (>= block bytes) so that a timed loop does not allocate."""
    ws, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(n, ws, rank)
    outs, ops = [], []
    for i, rb in enumerate(row_bytes):
        if rank == src:
            flat = fulls[i].reshape(-1)
            outs.append(flat[lo * rb:hi * rb])
            for peer in range(ws):
                plo, phi = shard_bounds(n, ws, peer)
                if peer != src and phi > plo:
                    ops.append((dist.isend, flat[plo * rb:phi * rb], peer))
        else:
            buf = recv[i][:(hi - lo) * rb] if recv is not None else torch.empty((hi - lo) * rb, dtype=torch.uint8, device=device)
            outs.append(buf)
            if hi > lo:
                ops.append((dist.irecv, buf, src))
    _post(ops)
    return outs


def gather_array(shard, n, row_bytes, device, dst=0, out=None):
    """Inverse of scatter_arrays for one result array: the n*row_bytes result on rank `dst` (None elsewhere).
    out: optional preallocated result tensor on the root."""
    ws, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(n, ws, rank)
    ops = []
    if rank == dst:
        if out is None:
            out = torch.empty(n * row_bytes, dtype=torch.uint8, device=device)
        mine = shard.reshape(-1)[:(hi - lo) * row_bytes]
        if mine.data_ptr() != out[lo * row_bytes:].data_ptr():     # the root may have computed in place
            out[lo * row_bytes:hi * row_bytes] = mine
        for peer in range(ws):
            plo, phi = shard_bounds(n, ws, peer)
            if peer != dst and phi > plo:
                ops.append((dist.irecv, out[plo * row_bytes:phi * row_bytes], peer))
    elif hi > lo:
        ops.append((dist.isend, shard.reshape(-1)[:(hi - lo) * row_bytes], dst))
    _post(ops)
    return out if rank == dst else None


def scatter_compute_gather(inputs, row_bytes_in, n, compute, row_bytes_out, device, src=0):
    """inputs: list of full uint8 tensors on rank `src` (ignored elsewhere)."""
    shards = scatter_arrays(inputs, n, row_bytes_in, device, src)
    res = compute(*shards)
    return gather_array(res, n, row_bytes_out, device, src)
