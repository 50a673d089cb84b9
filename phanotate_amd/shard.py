"""Per-contig sharding across the GPUs of one node.

Contigs are independent units (phanotate.py:40,56), so the multi-GPU path is: every rank takes a
subset of contigs, runs the whole path on its own GPU, and rank 0 re-assembles the per-contig
results in input order.  No collective touches the data path.  torch.distributed carries two things: the barrier
and the scalar reductions of a timed region (RCCL on the GPUs: `init_group` registers "cuda:nccl"), and the gather of
the per-contig results to rank 0 — flat host arrays, so they travel as CPU tensors over the "cpu:gloo" half of the same
group: fixed-dtype buffers, point to point, straight into their place in rank 0's arrays; nothing is pickled and nothing
is staged through HBM.  The CPU tests and `--smoke-single-device` (N ranks on one GPU) run exactly this gather.
"""
import heapq


def init_group(rank, world, device=None, single_device=False):
    """torch.distributed for a sharded run, one process per GPU.  CUDA tensors (barrier, all_reduce of the timed regions)
    go over RCCL, CPU tensors (the result gather) over gloo — one group, two backends.  `single_device`: every rank shares
    one GPU (RCCL refuses two ranks on one device), so the group is gloo only; the gather is the same code either way.
    Returns (torch.distributed, device the reductions use)."""
    import os

    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if single_device or device is None:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        return dist, "cpu"
    torch.cuda.set_device(device)
    dist.init_process_group(backend="cpu:gloo,cuda:nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    return dist, "cuda"


def partition(lengths, world):
    """Greedy longest-first assignment of contigs to `world` ranks, balanced by total bases.

    Deterministic: ties broken by contig index, then by rank.  Returns a list of index lists."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    heap = [(0, r) for r in range(world)]
    heapq.heapify(heap)
    out = [[] for _ in range(world)]
    for i in order:
        load, r = heapq.heappop(heap)
        out[r].append(i)
        heapq.heappush(heap, (load + int(lengths[i]), r))
    for r in range(world):
        out[r].sort()
    return out


def merge_flat(parts, n_total):
    """parts: per rank (indices, status, offsets, genes) as Annotator.annotate_flat returns them for the contigs `indices`
    (ascending).  -> (status[n_total], offsets[n_total+1], genes) in input order."""
    import numpy as np

    status = np.zeros(n_total, np.int32)
    count = np.zeros(n_total, np.int64)
    for idx, st, offs, _ in parts:
        idx = np.asarray(idx, np.int64)
        status[idx] = st
        count[idx] = np.diff(offs)
    offsets = np.zeros(n_total + 1, np.int64)
    np.cumsum(count, out=offsets[1:])
    genes = None
    for idx, st, offs, g in parts:
        if genes is None:
            genes = np.zeros(int(offsets[-1]), g.dtype)
        idx = np.asarray(idx, np.int64)
        if len(g):  # scatter this rank's runs of genes to their places in input order
            dst0 = np.repeat(offsets[idx] - offs[:-1], np.diff(offs))
            genes[dst0 + np.arange(len(g))] = g
    return status, offsets, genes


def gather_flat(dist, rank, world, idx, st, offs, genes):
    """Bring every rank's (indices, status, offsets, genes) to rank 0 as fixed-dtype CPU tensors: one int64 header exchange
    (gather of [contigs, genes] per rank), then per rank two point-to-point messages — int64[3 n] = indices | gene counts |
    statuses, and the gene records as bytes.  Rank 0 returns the list `merge_flat` takes; the others None."""
    import numpy as np
    import torch

    idx = np.ascontiguousarray(idx, np.int64)
    st = np.ascontiguousarray(st, np.int32)
    offs = np.ascontiguousarray(offs, np.int64)
    genes = np.ascontiguousarray(genes)
    n, g = len(idx), len(genes)
    assert len(st) == n and len(offs) == n + 1 and int(offs[-1]) == g
    head = torch.tensor([n, g, genes.dtype.itemsize], dtype=torch.int64)
    heads = [torch.zeros(3, dtype=torch.int64) for _ in range(world)] if rank == 0 else None
    dist.gather(head, heads, dst=0)
    if rank != 0:
        if n:
            dist.send(torch.from_numpy(np.concatenate([idx, np.diff(offs), st.astype(np.int64)])), dst=0)
        if g:
            dist.send(torch.from_numpy(genes.view(np.uint8).reshape(-1)), dst=0)
        return None
    parts = [(idx, st, offs, genes)]
    for r in range(1, world):
        nr, gr, isz = (int(x) for x in heads[r])
        assert isz == genes.dtype.itemsize
        hdr = np.zeros(3 * nr, np.int64)
        gbuf = np.zeros(gr, genes.dtype)
        if nr:
            dist.recv(torch.from_numpy(hdr), src=r)
        if gr:
            dist.recv(torch.from_numpy(gbuf.view(np.uint8).reshape(-1)), src=r)
        o = np.zeros(nr + 1, np.int64)
        np.cumsum(hdr[nr : 2 * nr], out=o[1:])
        assert int(o[-1]) == gr
        parts.append((hdr[:nr], hdr[2 * nr :].astype(np.int32), o, gbuf))
    return parts


def run_sharded_flat(seqs, annotate_flat, rank=0, world=1, dist=None, mine=None):
    """annotate_flat(list of sequences) -> (status, offsets, genes) flat arrays.  Every rank runs its shard; rank 0 returns the
    three arrays for the whole input in input order, the other ranks None (fixed-dtype arrays cross the wire, see gather_flat).
    `seqs` is the whole input, or with `mine` (this rank's ascending indices, n_total) only this rank's contigs."""
    if mine is None:
        n_total = len(seqs)
        idx = list(range(n_total)) if world == 1 else partition([len(s) for s in seqs], world)[rank]
        local = seqs if world == 1 else [seqs[i] for i in idx]
    else:
        idx, n_total = mine
        local = seqs
    st, offs, genes = annotate_flat(local)
    if world == 1:
        return st, offs, genes  # one rank holds every contig, already in input order
    gathered = gather_flat(dist, rank, world, idx, st, offs, genes)
    if rank != 0:
        return None
    return merge_flat(gathered, n_total)
