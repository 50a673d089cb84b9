"""Per-contig sharding across the GPUs of one node.

Contigs are independent units (phanotate.py:40,56), so the multi-GPU path is: every rank takes a
subset of contigs, runs the whole path on its own GPU, and rank 0 re-assembles the per-contig
results in input order.  No collective touches the data path; torch.distributed (RCCL on GPUs, gloo
in the CPU tests) is only used to bring the small result lists back.
"""
import heapq


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


def run_sharded(seqs, annotate, rank=0, world=1, dist=None):
    """annotate(list of sequences) -> list of per-contig results.  Returns the full list in input
    order on rank 0 (None elsewhere).  `dist` is torch.distributed (initialised) when world > 1."""
    if world == 1:
        return annotate(seqs)
    parts = partition([len(s) for s in seqs], world)
    mine = parts[rank]
    local = annotate([seqs[i] for i in mine])
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(list(zip(mine, local)), gathered, dst=0)
    if rank != 0:
        return None
    out = [None] * len(seqs)
    for part in gathered:
        for i, r in part:
            out[i] = r
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


def run_sharded_flat(seqs, annotate_flat, rank=0, world=1, dist=None, mine=None):
    """Like run_sharded with flat results (three arrays per rank cross the wire instead of one object per contig).
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
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((idx, st, offs, genes), gathered, dst=0)
    if rank != 0:
        return None
    return merge_flat(gathered, n_total)
