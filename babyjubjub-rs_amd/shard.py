"""Multi-process sharding of the batch path: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm,
"gloo" on CPU for tests).  The single-process, all-devices form is native: bjj_multi_* in libbjj_hip.so.

The path shards embarrassingly: every item reads only its own record, so the data path has NO
collective.  Two modes:
  * pre-sharded (what bench.py's headline times): each rank owns a contiguous block of the batch
  * scatter/gather (BASELINE.json cfg 5): rank 0 holds the whole batch; the input blocks are sent out, each rank runs
    the kernels on its block, the results come back to rank 0 -- serially (scatter_compute_gather) or PIPELINED
    (scatter_compute_gather_pipelined: peer blocks travel in pieces, a peer computes piece c while piece c+1 arrives,
    the root computes its own block from t = 0 while its sends are in flight; the schedule of bjj_multi_* in the library)

Scatter / gather move EXACT block sizes (contiguous ceil(n/G) blocks, `workload.shard_bounds`) as point-to-point
operations posted in ONE batch (`dist.batch_isend_irecv` = one ncclGroupStart/End on RCCL) for all arrays and all
peers: no padded copy of the batch, no per-array collective, and the root's own block is a view of its input.

`compute(shard_arrays...) -> result array` is supplied by the caller (the GPU library in production; tests pass the
oracle so the plumbing is checked on CPU).
"""
import torch
import torch.distributed as dist

from .workload import piece_bounds, shard_bounds


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


def _post_async(ops_spec):
    """Posts ops_spec as ONE batch and returns wait(): on RCCL the wait is a dependency of the CURRENT torch stream on the
    transfers (the host does not block); on gloo it blocks the host.  gloo + device tensors (the single-GPU developer mode of
    bench.py) cannot be asynchronous: they are staged through host copies at once and wait() is a no-op."""
    if not ops_spec:
        return lambda: None
    if dist.get_backend() == "gloo" and any(t.is_cuda for _, t, _ in ops_spec):
        _post(ops_spec)
        return lambda: None
    reqs = dist.batch_isend_irecv([dist.P2POp(fn, t, peer) for fn, t, peer in ops_spec])

    def wait():
        for r in reqs:
            r.wait()
    return wait


def scatter_compute_gather_pipelined(fulls, row_bytes_in, n, compute, row_bytes_out, device, src=0, pieces=4,
                                     min_piece=1 << 15, recv=None, res=None, out=None, streams=None):
    """BASELINE cfg 5 with the transfers hidden behind the kernels.  fulls: full uint8 input tensors on rank `src` (ignored
    elsewhere).  compute(arrays, count, out_view, k) must ENQUEUE the kernels for `count` items of the given input views and
    write count*row_bytes_out bytes into out_view; k numbers the calls of this rank (alternate two streams over it: the
    library then overlaps the tail of one piece with the head of the next).  streams: optional list of torch streams; piece k
    is issued under streams[k % len(streams)], so that the wait for its arrival and its kernels are ordered on that stream.
      root:  post the sends of ALL pieces (one batch per piece, all arrays and peers), compute its own block in place at once,
             post the receives of the result pieces, wait.
      peer:  post the receives of all pieces; per piece: wait for it, compute it, post the send of its results.
    recv / res: optional preallocated per-array input blocks / result block of this rank; out: optional result tensor on
    the root.  Returns the n*row_bytes_out result on the root, None elsewhere."""
    import contextlib
    ws, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(n, ws, rank)
    cnt = hi - lo
    on = (lambda k: torch.cuda.stream(streams[k % len(streams)])) if streams else (lambda k: contextlib.nullcontext())
    waits = []
    if rank == src:
        flats = [f.reshape(-1) for f in fulls]
        if out is None:
            out = torch.empty(n * row_bytes_out, dtype=torch.uint8, device=device)
        peers = [(p,) + shard_bounds(n, ws, p) for p in range(ws) if p != src]
        rounds = max([len(piece_bounds(phi - plo, pieces, min_piece)) for _, plo, phi in peers] + [0])
        for c in range(rounds):            # inputs out, piece by piece -- all posted before the root's own kernels
            ops = []
            for p, plo, phi in peers:
                pb = piece_bounds(phi - plo, pieces, min_piece)
                if c < len(pb):
                    a, b = plo + pb[c][0], plo + pb[c][1]
                    ops += [(dist.isend, flats[i][a * rb:b * rb], p) for i, rb in enumerate(row_bytes_in)]
            waits.append(_post_async(ops))
        if cnt:
            with on(0):
                compute([flats[i][lo * rb:hi * rb] for i, rb in enumerate(row_bytes_in)], cnt, out[lo * row_bytes_out:hi * row_bytes_out], 0)
        for c in range(rounds):            # results back, piece by piece
            ops = []
            for p, plo, phi in peers:
                pb = piece_bounds(phi - plo, pieces, min_piece)
                if c < len(pb):
                    a, b = plo + pb[c][0], plo + pb[c][1]
                    ops.append((dist.irecv, out[a * row_bytes_out:b * row_bytes_out], p))
            waits.append(_post_async(ops))
        for w in waits:
            w()
        return out
    pb = piece_bounds(cnt, pieces, min_piece)
    bufs = [(recv[i][:cnt * rb] if recv is not None else torch.empty(cnt * rb, dtype=torch.uint8, device=device))
            for i, rb in enumerate(row_bytes_in)]
    result = res[:cnt * row_bytes_out] if res is not None else torch.empty(cnt * row_bytes_out, dtype=torch.uint8, device=device)
    arrive = [_post_async([(dist.irecv, bufs[i][a * rb:b * rb], src) for i, rb in enumerate(row_bytes_in)]) for a, b in pb]
    for k, (a, b) in enumerate(pb):
        with on(k):
            arrive[k]()                    # this stream (on gloo: the host) waits for piece k only
            compute([bufs[i][a * rb:b * rb] for i, rb in enumerate(row_bytes_in)], b - a, result[a * row_bytes_out:b * row_bytes_out], k)
            waits.append(_post_async([(dist.isend, result[a * row_bytes_out:b * row_bytes_out], src)]))
    for w in waits:
        w()
    return None

